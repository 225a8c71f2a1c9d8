#!/bin/bash
# gated blocks' weight gradient (k = 3): chunks requested two ahead through branch-free buffer loads
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "channel_split_stack_kernels_equal or full_size" > $O/r6c_c22_bitwise.log 2>&1; tail -2 $O/r6c_c22_bitwise.log
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "golden" > $O/r6c_c22_step.log 2>&1; tail -2 $O/r6c_c22_step.log
bash tools/ab_libs.sh r6c_c22 "stack_wgrad" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
