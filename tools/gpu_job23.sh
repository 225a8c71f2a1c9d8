#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -m gpu -q -p no:cacheprovider -x -k "vq or quantizer or ema or goldens_bf16x3 or replays or oracle_fwd or emulating" > $OUT/r3_s23_tests.log 2>&1; tail -6 $OUT/r3_s23_tests.log
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s23_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s23_kernel_stats.csv; grep -E "vq_|masked_loss" $OUT/r3_s23_kernel_stats.csv | cut -c1-130; cat $OUT/r3_s23_bench_under_rocprof.json | cut -c1-200 )
