#!/bin/bash
# the EMA blend that leaves the search images current: its tests, the graph / step tests, the bench line and launch count
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "vq or codebook or quant or ema" > gpurun_out/bi_ops.log 2>&1; tail -2 gpurun_out/bi_ops.log
timeout 280 python -m pytest tests/test_gpu_step.py -m gpu -x -q --durations=5 > gpurun_out/bi_step.log 2>&1; tail -9 gpurun_out/bi_step.log
timeout 120 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bi_b.json
python -c "import json;d=json.load(open('gpurun_out/bi_b.json'));print('ms/step',round(d['ms_per_step'],4), d['launch'])"
