#!/bin/bash
# Round 5, second GPU session: the channel-split split-operand forward (stack2x_kernels.hip) - parity, then its price.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_nets.py -m gpu -x -q -s -k "split_forward_plain_backward" > $O/c2_x3f_nets.log 2>&1; grep -E "bf16x3f|passed|failed|Error|error" $O/c2_x3f_nets.log | head -30
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "bf16x3" > $O/c2_x3f_step.log 2>&1; tail -5 $O/c2_x3f_step.log
for v in 1 0; do
  CRK_S2X=$v timeout 300 python bench.py --precision bf16x3f --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2> $O/c2_b$v.err | tail -1 > $O/c2_bench_x3f_s2x$v.json
  python -c "import json;d=json.load(open('$O/c2_bench_x3f_s2x$v.json'));print('CRK_S2X=$v bf16x3f ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'])"
done
rm -rf $O/c2_p; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_p -- python bench.py --precision bf16x3f --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > $O/c2_prof.log 2>&1
python tools/kstats.py $O/c2_p > $O/c2_bf16x3f_kstats.txt; head -24 $O/c2_bf16x3f_kstats.txt; rm -rf $O/c2_p
