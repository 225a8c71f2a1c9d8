#!/bin/bash
# ms/step of the headline step over the weight-gradient group counts (CRK_WG_GROUPS: utterance groups of the stack
# weight gradients, CRK_WG_CPG: 64-frame chunks per group of the plain-conv ones), one session.
# usage (on the GPU box): bash tools/sweep_wgrad_groups.sh -> gpurun_out/sweep_wgrad_groups.txt
mkdir -p gpurun_out
out=gpurun_out/sweep_wgrad_groups.txt
: > $out
for cfg in "32 8" "16 8" "64 8" "32 16" "32 4" "16 16" "32 8"; do
  set -- $cfg
  ms=$(CRK_WG_GROUPS=$1 CRK_WG_CPG=$2 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "groups=$1 cpg=$2 ms_per_step=$ms" >> $out
done
cat $out
