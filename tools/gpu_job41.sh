#!/bin/bash
# final validation of round 3: bench line, kernel stats of the same command under rocprofv3, full GPU suite (4 workers)
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 240 python bench.py --steps 100 --warmup 10 > $OUT/r3_s41_bench.json 2> $OUT/r3_s41_bench.err; python -c "
import json
d=json.load(open('$OUT/r3_s41_bench.json'))
print({k:d[k] for k in ['value','ms_per_step','eager_ms_per_step','step_mfma_frac']}); print(d['stacks_alone']['ms'], d['stacks_alone']['frac_of_mfma_peak']); print(d['other_configs']['lsgan']['ms_per_step']); print(d['parity_mode']['ms_per_step'], d['parity_mode_both_directions']['ms_per_step']); print(d['cpu_baseline']['value']); r=d['roofline']; print({k:r[k] for k in ('bound','achieved','frac','traffic','avg_launch_us')})"
( cd /tmp && rm -rf /tmp/bk && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s41_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s41_kernel_stats.csv; head -4 $OUT/r3_s41_kernel_stats.csv | cut -c1-150 )
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -n 4 > $OUT/r3_s41_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r3_s41_pytest.log
tail -8 $OUT/r3_s41_pytest.log
