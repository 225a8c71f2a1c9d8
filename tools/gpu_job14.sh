#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -q -p no:cacheprovider -k "channel_split or plain_stack" > $OUT/r3_s14_bitwise.log 2>&1; tail -15 $OUT/r3_s14_bitwise.log
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider > $OUT/r3_s14_ops.log 2>&1; tail -15 $OUT/r3_s14_ops.log
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -q -p no:cacheprovider -s -k "forward_only or replays or test_step_matches_reference_goldens_bf16x3" > $OUT/r3_s14_step.log 2>&1; grep -E "bf16x3f|passed|failed|Error" $OUT/r3_s14_step.log | tail -12
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s14_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s14_kernel_stats.csv; head -40 $OUT/r3_s14_kernel_stats.csv | cut -c1-130; cat $OUT/r3_s14_bench_under_rocprof.json | cut -c1-300 )
