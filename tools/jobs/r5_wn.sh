#!/bin/bash
# (record of an experiment: the variant libraries it loops over were built from source edits that were NOT kept - see the
# profiles/round5_* file of the same experiment for what each variant was)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/wn_var.txt; : > $O
for lib in libcrank_hip.so libcrank_hip_wn_a.so libcrank_hip_wn_b.so libcrank_hip_wn_c.so libcrank_hip_wn_d.so libcrank_hip.so; do
  rm -rf /tmp/wnp; CRANK_AMD_LIB=$PWD/crank_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wnp -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > /tmp/wn_b.log 2>&1
  ks=$(find /tmp/wnp -name '*kernel_stats.csv' | head -1)
  echo "== $lib $(grep '^{' /tmp/wn_b.log | tail -1 | python -c 'import json,sys;print(round(json.loads(sys.stdin.read())["ms_per_step"],4))')" >> $O
  grep -E "wnorm_bwd_multi|weight_prep_multi|adam_kernel" $ks | cut -d, -f1-4 >> $O
  kt=$(find /tmp/wnp -name '*kernel_trace.csv' | head -1)
  python - "$kt" >> $O <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Kernel_Name'].startswith('wnorm_bwd_multi'):
        d[r.get('Grid_Size') or r.get('Grid_Size_X')].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for g,v in sorted(d.items()): print('   wnorm_bwd_multi grid',g,'n',len(v),'avg us %.1f'%(sum(v)/len(v)))
PY
done
cat $O
