#!/bin/bash
# plain chains, data gradient: the next layer's activation-derivative pieces requested in front of this layer's plane stores
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "channel_split_stack_kernels_equal or plain_stack" > $O/r6c_c19_bitwise.log 2>&1; tail -2 $O/r6c_c19_bitwise.log
bash tools/ab_libs.sh r6c_c19 "pstack2_kernel" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
