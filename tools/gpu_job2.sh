#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
tools/probe/issue > $OUT/r3_s2_issue.txt 2>&1; cat $OUT/r3_s2_issue.txt
timeout 300 python tools/s2_phase_cycles.py > $OUT/r3_s2_s2_phase.txt 2>&1; grep -E "all   :|residency|saving|no-grad" $OUT/r3_s2_s2_phase.txt
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r3_s2_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r3_s2_pytest.log
tail -40 $OUT/r3_s2_pytest.log
