#!/bin/bash
# EMA statistics: the reduce over the per-chunk tables with 32 tables in flight
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "ema or quantizer or vq" > $O/r6c_c18_ops.log 2>&1; tail -2 $O/r6c_c18_ops.log
bash tools/ab_libs.sh r6c_c18 "vq_ema" $PWD/crank_amd/libcrank_hip_base.so $PWD/crank_amd/libcrank_hip.so
