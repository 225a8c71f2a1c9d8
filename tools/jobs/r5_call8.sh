#!/bin/bash
# Round 5: the data-parallel path after the message merge (C3 rides in C2, SPKRADV + C one exchange, G's all-reduce asynchronous in replay)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-roofline 2> $O/c8_b.err | grep '^{' | tail -1 > $O/c8_bench.json
python -c "import json;d=json.load(open('$O/c8_bench.json'));print('single ms/step',d['ms_per_step'])" | tee $O/c8_dp_ab.txt
for rep in 1 2; do for cfg in "1 1" "0 0" "1 0" "0 1"; do set -- $cfg
  CRANK_AMD_DP_RIDE=$1 CRANK_AMD_DP_JOIN_SC=$2 timeout 300 python bench.py --gpus 1 --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2> $O/c8_fd.err | grep '^{' | tail -1 > $O/c8_fd.json
  python -c "import json;d=json.load(open('$O/c8_fd.json'));print('force-dist RIDE=$1 JOIN_SC=$2 ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'],d['launch'])" | tee -a $O/c8_dp_ab.txt
done; done
