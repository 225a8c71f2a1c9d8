#!/bin/bash
# plain convs' weight gradient: the dY fragment of a k step read once per wave where all its tiles share the cout band
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "channel_split_stack_kernels_equal or plain_stack" > $O/r6c_c15_bitwise.log 2>&1; tail -2 $O/r6c_c15_bitwise.log
timeout 900 python -m pytest tests/test_gpu_nets.py -m gpu -x -q > $O/r6c_c15_nets.log 2>&1; tail -2 $O/r6c_c15_nets.log
bash tools/ab_libs.sh r6c_c15 "pstack_wgrad" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
