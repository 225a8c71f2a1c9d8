#!/bin/bash
# Round 6, second session, call 2: the forward's new prologue (batched first conv, biases through the scalar cache, head prefetch)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 600 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "channel_split_stack_kernels_equal" > $O/r6b_c2_bitwise.log 2>&1; tail -3 $O/r6b_c2_bitwise.log
timeout 600 python -m pytest tests/test_gpu_nets.py -m gpu -x -q > $O/r6b_c2_nets.log 2>&1; tail -3 $O/r6b_c2_nets.log
CRANK_AMD_LIB=$PWD/crank_amd/libcrank_hip_prof_f.so timeout 200 python tools/store_cost_phases.py fwd 2>&1 | grep -v -i warn | tee $O/r6b_c2_store_phases.txt
bash tools/fwd_times.sh $O/r6b_c2_fwd_times.txt $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
