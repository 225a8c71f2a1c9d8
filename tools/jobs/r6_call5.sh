#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 600 python tools/s2p_phase_cycles.py 2>&1 | grep -v Warn | tee $O/r6_s2p_phase_cycles.txt
timeout 900 python -m pytest tests/test_gpu_dp.py -m gpu -x -q -s -k "n_ranks and cyclegan and 8" > $O/r6_c5_dp.log 2>&1; grep -E "ranks vs one|grad/|passed|failed" $O/r6_c5_dp.log | tail
