#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r3_s32_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r3_s32_pytest.log
tail -6 $OUT/r3_s32_pytest.log
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s32_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s32_kernel_stats.csv )
( cd /tmp && rm -rf /tmp/bl && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bl -- python $GRAFT_REPO_ROOT/bench.py --trainer lsgan --steps 30 --warmup 6 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s32_lsgan_bench_under_rocprof.json 2>/tmp/bl.err; f=$(find /tmp/bl -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s32_lsgan_kernel_stats.csv )
timeout 900 bash tools/pmc_traffic.sh > $OUT/r3_s32_pmc.log 2>&1; cp $OUT/pmc_traffic.csv $OUT/r3_s32_pmc_traffic.csv
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/r3_s32_bench.json 2> $OUT/r3_s32_bench.err; python -c "
import json
d=json.load(open('$OUT/r3_s32_bench.json'))
print({k:d[k] for k in ['value','ms_per_step','eager_ms_per_step','step_mfma_frac']}); print(d['stacks_alone']['ms'], d['stacks_alone']['frac_of_mfma_peak']); print(d['other_configs']['lsgan']['ms_per_step']); print(d['parity_mode']['ms_per_step'], d['parity_mode_both_directions']['ms_per_step']); print(d['cpu_baseline']['value']); r=d['roofline']; print({k:r[k] for k in ('bound','achieved','frac','traffic','avg_launch_us')})"
