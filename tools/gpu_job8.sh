#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python tools/diag_bwd.py > $OUT/r3_s8_diag.txt 2>&1; grep -c "differing        0" $OUT/r3_s8_diag.txt; grep -v "differing        0" $OUT/r3_s8_diag.txt | head
python tools/s2b_phase_cycles.py 2>&1 | grep "all :" 
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/sa2 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sa2 -- python $GRAFT_REPO_ROOT/tools/prof_stacks_alone.py 8 > /tmp/sa.log 2>&1; python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/sa2 stack2_bwd )
