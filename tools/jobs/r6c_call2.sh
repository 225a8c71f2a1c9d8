#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 200 python tools/ps2_phase_cycles.py 2>&1 | grep -v -i warn | tee $O/r6c_ps2_phase2.txt
