#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r3_s19_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r3_s19_pytest.log
tail -12 $OUT/r3_s19_pytest.log
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s19_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s19_kernel_stats.csv; grep -E "weight_prep|wnorm|recon" $OUT/r3_s19_kernel_stats.csv | cut -c1-130; cat $OUT/r3_s19_bench_under_rocprof.json | cut -c1-300 )
