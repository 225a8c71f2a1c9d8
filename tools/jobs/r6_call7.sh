#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python tools/diag_dp8.py cyclegan 2>&1 | grep -v Warn | tail -8 | tee $O/r6_diag_dp8.txt
timeout 900 python tools/diag_dp8.py lsgan 2>&1 | grep -v Warn | tail -5 | tee -a $O/r6_diag_dp8.txt
