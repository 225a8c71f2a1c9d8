#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for i in 1 2; do
for g in 1 0; do CRANK_AMD_GROUP_MAINT=$g timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grouped=$g', d['ms_per_step'])"; done
done
