#!/bin/bash
# stack2_fwd kernel times of a generator forward (tools/prof_fwd.py) without / with saved planes, for each library given:
#   tools/fwd_times.sh <out file> <lib.so> [<lib.so> ...]
cd /tmp; export TMPDIR=/tmp
OUT=$1; shift
for lib in "$@"; do for mode in nograd fwd; do
  rm -rf /tmp/pf_x
  CRANK_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_x -- python $GRAFT_REPO_ROOT/tools/prof_fwd.py $mode 8 > /tmp/pf.log 2>&1 || tail -3 /tmp/pf.log
  f=$(find /tmp/pf_x -name "*kernel_trace.csv" | head -1)
  python - "$f" "$(basename $lib) $mode" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "stack2" in n:
        acc[n.split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{sys.argv[2]:36s}", "  ".join(f"{k[5:]}: {sum(sorted(v)[:-1])/(len(v)-1):6.1f}" for k, v in sorted(acc.items())))
PY
done; done 2>&1 | tee $OUT
