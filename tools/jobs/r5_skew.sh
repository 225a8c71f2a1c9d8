#!/bin/bash
# Experiment: two independent 4-wave forward workgroups per CU (128-row windows, CRK_S2_CFG=41) with the second one started late
# (CRK_S2_SKEW, units of ~512 cycles) against the shipped 8-wave shapes; saving (fwd) and no-grad G forwards, kernel trace.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/skew.txt; : > $O
run() {  # label, env...
  label=$1; shift
  for mode in fwd nograd; do
    rm -rf /tmp/skw; env "$@" timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/skw -- python tools/prof_fwd.py $mode 8 > /tmp/skw.log 2>&1 || tail -3 /tmp/skw.log
    echo "== $label $mode" >> $O; python tools/kstats.py /tmp/skw stack2_fwd >> $O
  done
}
run "shipped shapes" CRK_S2_CFG=0
for sk in 0 4 8 12 16 24; do run "cfg 41 skew $sk" CRK_S2_CFG=41 CRK_S2_SKEW=$sk; done
cat $O
