#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 6 --precision bf16x3f --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s33_bench_x3f.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s33_x3f_kernel_stats.csv; head -22 $OUT/r3_s33_x3f_kernel_stats.csv | cut -c1-140; cat $OUT/r3_s33_bench_x3f.json | cut -c1-200; tail -2 /tmp/bk.err )
