#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "vq or codebook or quantizer or ema" > $O/c17_vq_tests.log 2>&1; tail -4 $O/c17_vq_tests.log
timeout 200 python tools/vq_phase_cycles.py 2>&1 | grep -v amdgpu.ids | tee $O/c17_vq_phases.txt
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2> $O/c17_b.err | grep '^{' | tail -1 > $O/c17_b.json
python -c "import json;d=json.load(open('$O/c17_b.json'));print('ms/step',d['ms_per_step']); c=d['roofline']['classes']; print([ (k[:24], round(v['avg_us'],1)) for k,v in c.items()])"
