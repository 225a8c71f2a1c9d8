#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "channel_split_stack_kernels_equal" > $O/r6_c6_bitwise.log 2>&1; tail -3 $O/r6_c6_bitwise.log
timeout 600 python tools/s2p_phase_cycles.py 2>&1 | grep -v Warn | tee $O/r6_s2p_phase_cycles.txt
cd /tmp
for pipe in 1 0; do for mode in nograd fwd; do
  rm -rf /tmp/pf_$pipe$mode
  CRK_S2_PIPE=$pipe timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_$pipe$mode -- python $GRAFT_REPO_ROOT/tools/prof_fwd.py $mode 8 > /tmp/pf.log 2>&1 || tail -3 /tmp/pf.log
  f=$(find /tmp/pf_$pipe$mode -name "*kernel_trace.csv" | head -1)
  python - "$f" "pipe=$pipe $mode" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "stack2" in n:
        acc[n.split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[2], "  ".join(f"{k[5:]}: {sum(sorted(v)[:-1])/(len(v)-1):6.1f}" for k, v in sorted(acc.items())))
PY
done; done 2>&1 | tee $O/r6_c6_fwd_times.txt
cd $GRAFT_REPO_ROOT
for pipe in 1 0; do
  CRK_S2_PIPE=$pipe timeout 600 python bench.py --no-cpu-baseline --no-extras 2> $O/r6_c6_b.err | tail -1 > $O/r6_c6_b$pipe.json
  python -c "import json;d=json.load(open('$O/r6_c6_b$pipe.json'));print('CRK_S2_PIPE=$pipe ms/step',d['ms_per_step'],'roof',d['roofline']['frac'],'stacks_alone',d['stacks_alone'].get('frac_of_mfma_peak'))" | tee -a $O/r6_c6_ab_pipe.txt
done
