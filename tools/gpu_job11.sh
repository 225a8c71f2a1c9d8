#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r3_s11_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r3_s11_pytest.log
tail -15 $OUT/r3_s11_pytest.log
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/r3_s11_bench.json 2> $OUT/r3_s11_bench.err; python -c "
import json
d=json.load(open('$OUT/r3_s11_bench.json'))
print({k:d[k] for k in ['value','ms_per_step','eager_ms_per_step','step_mfma_frac']}); print(d['stacks_alone']); print(d['other_configs'])"
