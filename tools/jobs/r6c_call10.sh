#!/bin/bash
# last tile of frame half 1 left out from block l_drop on: with / without the priority flip, against the kernel before
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "channel_split_stack_kernels_equal" > $O/r6c_c10_bitwise.log 2>&1; tail -2 $O/r6c_c10_bitwise.log
bash tools/ab_libs.sh r6c_c10 "stack2_fwd|ce_partial|embed_bwd|pstack_wgrad" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip_noflip.so $PWD/crank_amd/libcrank_hip.so
