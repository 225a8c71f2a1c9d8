#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/s2_phase_cycles.py 2>&1 | grep -E "residency|all   :" > gpurun_out/r3_s39_s2_phase.txt; cat gpurun_out/r3_s39_s2_phase.txt
timeout 200 python bench.py --steps 200 --warmup 20 2>&1 | tail -1 > gpurun_out/r3_s39_bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_s39_bench.json'))
print(d['ms_per_step'], d.get('stacks_alone'))
PY
timeout 600 python -m pytest tests/test_gpu_nets.py tests/test_gpu_step.py -x -q -m gpu 2>&1 | tail -4
