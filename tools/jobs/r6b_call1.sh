#!/bin/bash
# Round 6, second session, call 1: where the plane stores' time lands (phase cycles with / without real stores), and the
# step's kernel sequence on HEAD
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
CRANK_AMD_LIB=$PWD/crank_amd/libcrank_hip_prof_f.so timeout 200 python tools/store_cost_phases.py fwd 2>&1 | grep -v -i warn | tee $O/r6b_store_phases.txt
for v in prof_b prof_bns; do
  CRANK_AMD_LIB=$PWD/crank_amd/libcrank_hip_$v.so timeout 200 python tools/store_cost_phases.py bwd 2>&1 | grep -v -i warn | tee -a $O/r6b_store_phases.txt
done
rm -rf /tmp/bip; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bip -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > /tmp/bi_b.log 2>&1
kt=$(find /tmp/bip -name '*kernel_trace.csv' | head -1); ks=$(find /tmp/bip -name '*kernel_stats.csv' | head -1)
[ -n "$kt" ] && python tools/step_sequence.py "$kt" $O/r6b_step_sequence.txt && head -1 $O/r6b_step_sequence.txt
[ -n "$ks" ] && cp "$ks" $O/r6b_kernel_stats.csv
grep '^{' /tmp/bi_b.log | tail -1 | python -c 'import json,sys;print("ms/step under rocprof", round(json.loads(sys.stdin.read())["ms_per_step"],4))'
