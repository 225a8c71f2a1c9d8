#!/bin/bash
# cross entropy of 12 / 14 classes with the row in registers
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $O/r6c_c16_ops.log 2>&1; tail -2 $O/r6c_c16_ops.log
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "golden" > $O/r6c_c16_step.log 2>&1; tail -2 $O/r6c_c16_step.log
bash tools/ab_libs.sh r6c_c16 "ce_partial" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
