#!/bin/bash
# plain chains' prologue: the window's rows requested by a branch-free loop (VEC as a template parameter), layer-0 fragments in two parts
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "channel_split_stack_kernels_equal or plain_stack" > $O/r6c_c20_bitwise.log 2>&1; tail -2 $O/r6c_c20_bitwise.log
timeout 900 python -m pytest tests/test_gpu_nets.py -m gpu -x -q > $O/r6c_c20_nets.log 2>&1; tail -2 $O/r6c_c20_nets.log
timeout 200 python tools/ps2_phase_cycles.py 2>&1 | grep -v -i warn | tee $O/r6c_c20_phases.txt
bash tools/ab_libs.sh r6c_c20 "pstack2_kernel" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
