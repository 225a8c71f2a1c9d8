#!/bin/bash
# A/B of one environment switch on the headline step and the lsgan step, both values in one session, two repetitions.
# usage (on the GPU box): bash tools/ab_env.sh VAR "trainers"  -> gpurun_out/ab_VAR.txt
var=$1; trainers=${2:-"vqvae lsgan"}
mkdir -p gpurun_out
out=gpurun_out/ab_$var.txt
: > $out
for rep in 1 2 3; do
  for v in ${VALUES:-0 1}; do
    for tr in $trainers; do
      ms=$(env $var=$v timeout 300 python bench.py --trainer $tr --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --no-extras 2>gpurun_out/ab_err.log | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
      echo "$var=$v $tr rep=$rep ms_per_step=$ms" >> $out
    done
  done
done
cat $out; tail -3 gpurun_out/ab_err.log
