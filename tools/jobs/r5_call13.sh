#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_properties.py -m gpu -x -q -k "lsgan" > $O/c13_tests.log 2>&1; tail -3 $O/c13_tests.log
rm -f $O/c13_ab_reuse_enc.txt
for rep in 1 2; do for v in 1 0; do
  CRANK_AMD_REUSE_ENC=$v timeout 300 python bench.py --trainer lsgan --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2> $O/c13_b.err | grep '^{' | tail -1 > $O/c13_b.json
  python -c "import json;d=json.load(open('$O/c13_b.json'));print('lsgan CRANK_AMD_REUSE_ENC=$v ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'])" | tee -a $O/c13_ab_reuse_enc.txt
done; done
