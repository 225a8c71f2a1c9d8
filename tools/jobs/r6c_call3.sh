#!/bin/bash
# plain-conv weight gradients with chunks requested two ahead: bitwise tests, then A/B against the build before
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_properties.py -m gpu -x -q > $O/r6c_c3_tests.log 2>&1; tail -3 $O/r6c_c3_tests.log
bash tools/ab_libs.sh r6c_c3 "wgrad|wnorm" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
