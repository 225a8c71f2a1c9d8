#!/bin/bash
# round 3, GPU session 1: the whole -m gpu suite on the new tree, then baselines for the kernel work
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/r3_s1_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r3_s1_pytest.log
tail -15 $OUT/r3_s1_pytest.log
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/sa && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sa -- python $GRAFT_REPO_ROOT/tools/prof_stacks_alone.py 8 > /tmp/sa.log 2>&1; python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/sa > $OUT/r3_s1_stacks_alone_kstats.txt 2>&1 )
head -20 $OUT/r3_s1_stacks_alone_kstats.txt
timeout 300 python tools/s2_phase_cycles.py > $OUT/r3_s1_s2_phase.txt 2>&1; tail -12 $OUT/r3_s1_s2_phase.txt
timeout 300 python tools/skb_phase_cycles.py > $OUT/r3_s1_skb_phase.txt 2>&1; tail -8 $OUT/r3_s1_skb_phase.txt
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/r3_s1_bench.json 2> $OUT/r3_s1_bench.err; tail -c 1500 $OUT/r3_s1_bench.json
