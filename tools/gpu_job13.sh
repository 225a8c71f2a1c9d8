#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r3_s13_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r3_s13_pytest.log
tail -8 $OUT/r3_s13_pytest.log
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s13_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s13_kernel_stats.csv; head -45 $OUT/r3_s13_kernel_stats.csv | cut -c1-150 )
timeout 900 bash tools/pmc_traffic.sh > $OUT/r3_s13_pmc.log 2>&1; cp $OUT/pmc_traffic.csv $OUT/r3_s13_pmc_traffic.csv; head -12 $OUT/r3_s13_pmc_traffic.csv
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/r3_s13_bench.json 2> $OUT/r3_s13_bench.err; python -c "
import json
d=json.load(open('$OUT/r3_s13_bench.json'))
print({k:d[k] for k in ['value','ms_per_step','eager_ms_per_step','step_mfma_frac']}); print(d['stacks_alone']); print(d['other_configs']); print(d['parity_mode']['ms_per_step']); r=d['roofline']; print({k:r[k] for k in r if k!='classes'})"
