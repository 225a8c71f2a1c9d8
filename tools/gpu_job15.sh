#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python tools/ps2_phase_cycles.py > $OUT/r3_s15_ps2_phase.txt 2>&1; tail -12 $OUT/r3_s15_ps2_phase.txt
timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -q -p no:cacheprovider -k "channel_split" > $OUT/r3_s15_bitwise.log 2>&1; tail -5 $OUT/r3_s15_bitwise.log
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s15_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s15_kernel_stats.csv; grep -E "pstack|recon|wnorm|weight_prep" $OUT/r3_s15_kernel_stats.csv | cut -c1-130; cat $OUT/r3_s15_bench_under_rocprof.json | cut -c1-300 )
