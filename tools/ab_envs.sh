#!/bin/bash
# A/B of environment settings on the replayed vqvae step, all in one session, alternating, REPS rounds:
#   tools/ab_envs.sh <tag> "A=1" "B=0 C=2" ...   ("-" = no setting)  -> gpurun_out/<tag>_envs.txt
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
TAG=$1; shift
: > $O/${TAG}_envs.txt
for rep in $(seq 1 ${REPS:-2}); do for e in "$@"; do
  [ "$e" = "-" ] && ee="" || ee="$e"
  ms=$(env $ee timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-roofline 2>/tmp/abe.err | grep '^{' | tail -1 | python -c 'import json,sys;d=json.loads(sys.stdin.read());print(round(d["ms_per_step"],4), "eager", round(d.get("eager_ms_per_step") or 0,4))')
  echo "[$e] rep=$rep ms_per_step=$ms" | tee -a $O/${TAG}_envs.txt
done; done
