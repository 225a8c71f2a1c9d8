#!/bin/bash
# cross entropy through LDS, deeper embedding-gradient loads, branch-free input activation of the plain chains
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_nets.py tests/test_gpu_ops.py -m gpu -x -q > $O/r6c_c9_tests.log 2>&1; tail -3 $O/r6c_c9_tests.log
timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "channel_split_stack_kernels_equal" > $O/r6c_c9_bitwise.log 2>&1; tail -2 $O/r6c_c9_bitwise.log
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "golden" > $O/r6c_c9_step.log 2>&1; tail -2 $O/r6c_c9_step.log
bash tools/ab_libs.sh r6c_c9 "ce_partial|embed_bwd|pstack2_kernel|stack2_fwd" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
