#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s27_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s27_kernel_stats.csv; head -30 $OUT/r3_s27_kernel_stats.csv | cut -c1-130 )
( cd /tmp && rm -rf /tmp/bl && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bl -- python $GRAFT_REPO_ROOT/bench.py --trainer lsgan --steps 30 --warmup 6 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s27_lsgan_bench_under_rocprof.json 2>/tmp/bl.err; f=$(find /tmp/bl -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s27_lsgan_kernel_stats.csv; tail -2 /tmp/bl.err )
timeout 900 bash tools/pmc_traffic.sh > $OUT/r3_s27_pmc.log 2>&1; cp $OUT/pmc_traffic.csv $OUT/r3_s27_pmc_traffic.csv; head -8 $OUT/r3_s27_pmc_traffic.csv
timeout 600 python tools/ps2_phase_cycles.py > $OUT/r3_s27_ps2_phase.txt 2>&1; tail -8 $OUT/r3_s27_ps2_phase.txt
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/r3_s27_bench.json 2> $OUT/r3_s27_bench.err; python -c "
import json
d=json.load(open('$OUT/r3_s27_bench.json'))
print({k:d[k] for k in ['value','ms_per_step','eager_ms_per_step','step_mfma_frac']}); print(d['stacks_alone']); print(d['other_configs']); print(d['parity_mode']['ms_per_step'], d['parity_mode_both_directions']); print(d['cpu_baseline']); r=d['roofline']; print({k:r[k] for k in r if k!='classes'}); print(r.get('classes'))"
