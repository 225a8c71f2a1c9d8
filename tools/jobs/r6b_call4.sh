#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 600 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "channel_split_stack_kernels_equal" > $O/r6b_c4_bitwise.log 2>&1; tail -2 $O/r6b_c4_bitwise.log
CRANK_AMD_LIB=$PWD/crank_amd/libcrank_hip_prof_f.so timeout 200 python tools/store_cost_phases.py fwd 2>&1 | grep -v -i warn | grep -v "^      wave" | tee $O/r6b_c4_store_phases.txt
bash tools/fwd_times.sh $O/r6b_c4_fwd_times.txt $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
