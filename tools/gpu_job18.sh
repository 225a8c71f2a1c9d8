#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -q -p no:cacheprovider -s -k "forward_only" > $OUT/r3_s18_step.log 2>&1; grep -E "bf16x3f|passed|failed|Error|assert" $OUT/r3_s18_step.log | tail -20
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/r3_s18_bench.json 2> $OUT/r3_s18_bench.err; python -c "
import json
d=json.load(open('$OUT/r3_s18_bench.json'))
print({k:d[k] for k in ['value','ms_per_step','eager_ms_per_step','step_mfma_frac']}); print(d['stacks_alone']); print(d['other_configs']); print(d['parity_mode']); print(d.get('parity_mode_both_directions')); r=d['roofline']; print({k:r[k] for k in r if k!='classes'})"
