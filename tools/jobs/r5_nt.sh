#!/bin/bash
# (record of an experiment: the variant libraries it loops over were built from source edits that were NOT kept - see the
# profiles/round5_* file of the same experiment for what each variant was)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/nt_stores.txt; : > $O
for rep in 1 2; do for lib in libcrank_hip.so libcrank_hip_nt2.so; do
  CRANK_AMD_LIB=$PWD/crank_amd/$lib timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > gpurun_out/nt_b.json
  python -c "import json;d=json.load(open('gpurun_out/nt_b.json'));c=d['roofline']['classes'];print('$lib ms/step',round(d['ms_per_step'],4),'stacks_alone',round(d['stacks_alone']['ms'],4),[(k[:16],round(v['avg_us'],1)) for k,v in c.items() if k.startswith('stack')])" | tee -a $O
done; done
