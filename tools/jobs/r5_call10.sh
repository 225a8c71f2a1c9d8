#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "lsgan or cyclegan or stargan or golden" > $O/c10_step.log 2>&1; tail -3 $O/c10_step.log
rm -f $O/c10_ab_d_batch.txt
for rep in 1 2; do for v in 1 0; do
  CRANK_AMD_D_BATCH=$v timeout 300 python bench.py --trainer lsgan --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2> $O/c10_b.err | grep '^{' | tail -1 > $O/c10_b.json
  python -c "import json;d=json.load(open('$O/c10_b.json'));print('lsgan CRANK_AMD_D_BATCH=$v ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'])" | tee -a $O/c10_ab_d_batch.txt
done; done
for t in cyclegan stargan; do for v in 1 0; do
  CRANK_AMD_D_BATCH=$v timeout 300 python bench.py --trainer $t --batch 32 --steps 30 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2> $O/c10_b.err | grep '^{' | tail -1 > $O/c10_b.json
  python -c "import json;d=json.load(open('$O/c10_b.json'));print('$t B=32 CRANK_AMD_D_BATCH=$v ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'])" | tee -a $O/c10_ab_d_batch.txt
done; done
rm -rf $O/c10_p; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c10_p -- python bench.py --trainer lsgan --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > $O/c10_prof.log 2>&1
python tools/kstats.py $O/c10_p > $O/c10_lsgan_kstats.txt; head -16 $O/c10_lsgan_kstats.txt; rm -rf $O/c10_p
