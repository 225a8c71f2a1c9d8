#!/bin/bash
# weight-gradient group count of the gated stacks (CRK_WG_GROUPS, default 32): step time and the kernel classes it moves
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/wg_groups.txt; : > $O
for g in 32 16 8 32 16 64; do
  CRK_WG_GROUPS=$g timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > gpurun_out/wg_b.json
  python -c "import json;d=json.load(open('gpurun_out/wg_b.json'));c=d['roofline']['classes'];print('groups $g ms/step',round(d['ms_per_step'],4),'stacks_alone',round(d['stacks_alone']['ms'],4),[(k[:16],round(v['avg_us'],1)) for k,v in c.items()])" | tee -a $O
done
