#!/bin/bash
# Round 5, fourth GPU session: the codebook image of the VQ search
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "vq or codebook or quantizer or ema" > $O/c4_vq_tests.log 2>&1; tail -5 $O/c4_vq_tests.log
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q > $O/c4_step_tests.log 2>&1; tail -3 $O/c4_step_tests.log
for v in 1 0 1 0; do
  CRANK_AMD_VQ_IMAGE=$v timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-roofline 2> $O/c4_b$v.err | tail -1 > $O/c4_bench_img$v.json
  python -c "import json;d=json.load(open('$O/c4_bench_img$v.json'));print('CRANK_AMD_VQ_IMAGE=$v ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'])" | tee -a $O/c4_ab_vq_image.txt
done
rm -rf $O/c4_p; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_p -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $O/c4_prof.log 2>&1
python tools/kstats.py $O/c4_p > $O/c4_kstats.txt; grep -E "vq_|total" $O/c4_kstats.txt | head; rm -rf $O/c4_p
