#!/bin/bash
# Round 5, third GPU session: bf16x3f with the plain chains channel-split as well (pstack2x_kernels.hip)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_nets.py -m gpu -x -q -s -k "split_forward_plain_backward" > $O/c3_x3f_nets.log 2>&1; grep -E "bf16x3f|passed|failed|Error|error" $O/c3_x3f_nets.log | head -30
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "bf16x3" > $O/c3_x3f_step.log 2>&1; tail -5 $O/c3_x3f_step.log
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2> $O/c3_b.err | tail -1 > $O/c3_bench_bf16.json
python -c "import json;d=json.load(open('$O/c3_bench_bf16.json'));print('bf16 ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'])"
for v in 1 0; do
  CRK_S2X=$v timeout 300 python bench.py --precision bf16x3f --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2> $O/c3_b$v.err | tail -1 > $O/c3_bench_x3f_s2x$v.json
  python -c "import json;d=json.load(open('$O/c3_bench_x3f_s2x$v.json'));print('CRK_S2X=$v bf16x3f ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'])"
done
for oc in 0 1; do
rm -rf $O/c3_p; CRANK_AMD_OVERLAP_C=$oc timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_p -- python bench.py --precision bf16x3f --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > $O/c3_prof.log 2>&1
python tools/kstats.py $O/c3_p > $O/c3_bf16x3f_kstats_overlap$oc.txt; head -14 $O/c3_bf16x3f_kstats_overlap$oc.txt; rm -rf $O/c3_p
done
bash tools/pmc_stacks_traffic.sh $PWD/$O/c3_pmc_stacks_traffic.txt > /dev/null 2>&1; tail -8 $O/c3_pmc_stacks_traffic.txt
