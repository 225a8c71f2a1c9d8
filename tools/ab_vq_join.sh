#!/bin/bash
# A/B of the gradient join in the quantizer backward (CRANK_AMD_VQ_JOIN), both variants in one session (pool boxes differ).
# usage (on the GPU box): bash tools/ab_vq_join.sh  -> gpurun_out/ab_vq_join.txt
mkdir -p gpurun_out
out=gpurun_out/ab_vq_join.txt
: > $out
for rep in 1 2; do
  for j in 0 1; do
    for tr in vqvae lsgan; do
      line=$(CRANK_AMD_VQ_JOIN=$j timeout 300 python bench.py --trainer $tr --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1)
      echo "join=$j $tr rep=$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("launches_per_step"))')" >> $out
    done
  done
done
cat $out
