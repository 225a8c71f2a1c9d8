#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in default wn_4_16 wn_8_8 wn_4_8; do
  if [ $v = default ]; then L=$GRAFT_REPO_ROOT/crank_amd/libcrank_hip.so; else L=$GRAFT_REPO_ROOT/crank_amd/libcrank_hip_$v.so; fi
  ( cd /tmp && rm -rf /tmp/bk && CRANK_AMD_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-extras --no-roofline > /tmp/b.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); echo "$v $(grep wnorm_bwd_multi $f | cut -d, -f2-7) $(python -c "import json;print(json.load(open('/tmp/b.json'))['ms_per_step'])")" )
done
