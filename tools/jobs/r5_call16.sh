#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_properties.py -m gpu -x -q -k "cyclegan or stargan or lsgan" > $O/c16_tests.log 2>&1; tail -3 $O/c16_tests.log
rm -f $O/c16_ab.txt
for t in cyclegan stargan; do for v in 1 0 1 0; do
  CRANK_AMD_REUSE_ENC=$v timeout 300 python bench.py --trainer $t --batch 32 --steps 40 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2> $O/c16_b.err | grep '^{' | tail -1 > $O/c16_b.json
  python -c "import json;d=json.load(open('$O/c16_b.json'));print('$t B=32 CRANK_AMD_REUSE_ENC=$v ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'])" | tee -a $O/c16_ab.txt
done; done
