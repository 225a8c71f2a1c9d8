#!/bin/bash
# the VQ search with packed candidate keys: parity (every VQ test), phase cycles, the bench line's vq class
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "vq or codebook or quant" > gpurun_out/vqkey_pytest.log 2>&1; tail -3 gpurun_out/vqkey_pytest.log
timeout 300 python tools/vq_phase_cycles.py > gpurun_out/vqkey_phase.txt 2>&1; tail -12 gpurun_out/vqkey_phase.txt
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > gpurun_out/vqkey_b.json
python -c "import json;d=json.load(open('gpurun_out/vqkey_b.json'));c=d['roofline']['classes'];print('ms/step',round(d['ms_per_step'],4),[(k[:16],round(v['avg_us'],1)) for k,v in c.items()])"
