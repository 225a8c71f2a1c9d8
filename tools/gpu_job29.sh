#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_properties.py tests/test_gpu_nets.py -m gpu -q -p no:cacheprovider -x -k "plain or channel_split or grouped or standalone or classifier or spkradv or cross_entropy" > $OUT/r3_s29_tests.log 2>&1; tail -4 $OUT/r3_s29_tests.log
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s29_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s29_kernel_stats.csv; grep -E "pstack_wgrad|stack2_fwd_kernel<3" $OUT/r3_s29_kernel_stats.csv | cut -c1-130; cat $OUT/r3_s29_bench_under_rocprof.json | cut -c1-200 )
