#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out; rm -f $O/c18_dp_graph.txt
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-roofline 2> $O/c18_b.err | grep '^{' | tail -1 > $O/c18_b.json
python -c "import json;d=json.load(open('$O/c18_b.json'));print('single ms/step',d['ms_per_step'])" | tee -a $O/c18_dp_graph.txt
for rep in 1 2; do for v in 0 1; do
  CRANK_AMD_DP_GRAPH_COLLECTIVES=$v timeout 300 python bench.py --gpus 1 --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2> $O/c18_fd$v.err | grep '^{' | tail -1 > $O/c18_fd.json
  python -c "import json;d=json.load(open('$O/c18_fd.json'));print('force-dist GRAPH_COLLECTIVES=$v ms/step',d['ms_per_step'],'eager',d['eager_ms_per_step'],d['launch'])" 2>&1 | tail -1 | tee -a $O/c18_dp_graph.txt
done; done
tail -5 $O/c18_fd1.err | cut -c1-300
CRANK_AMD_DP_GRAPH_COLLECTIVES=1 timeout 600 python -m pytest tests/test_gpu_dp.py -m gpu -x -q -k "rccl" 2>&1 | tail -3
