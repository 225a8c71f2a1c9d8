#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python tools/s2_phase_cycles.py > $OUT/r3_s22_s2_phase.txt 2>&1; grep -E "^[a-z0-9 -]+:|all   :|residency" $OUT/r3_s22_s2_phase.txt | cut -c1-420
