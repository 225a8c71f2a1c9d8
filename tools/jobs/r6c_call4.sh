#!/bin/bash
# phases of the plain-conv weight gradient; small-net weight-norm backward / weight preparation A/B
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 200 python tools/ps2_phase_cycles.py 2>&1 | grep -v -i warn | tee $O/r6c_c4_phases.txt
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_properties.py tests/test_gpu_step.py -m gpu -x -q > $O/r6c_c4_tests.log 2>&1; tail -3 $O/r6c_c4_tests.log
bash tools/ab_libs.sh r6c_c4 "wgrad|wnorm|weight_prep" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
