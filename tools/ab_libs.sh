#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh <tag> <kernel-name regex> <lib.so> [<lib.so> ...]
#   for every library, alternating, three rounds: the replayed vqvae step (bench.py, no extras) -> ms per step;
#   then once per library the same command under rocprofv3 --kernel-trace -> average duration per kernel matching the regex
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
TAG=$1; RE=$2; shift 2
: > $O/${TAG}_ab.txt
for rep in 1 2 3; do for lib in "$@"; do
  CRANK_AMD_LIB=$lib timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-roofline > /tmp/ab.log 2>/tmp/ab.err || tail -3 /tmp/ab.err
  echo "$(basename $lib) rep=$rep ms_per_step=$(grep '^{' /tmp/ab.log | tail -1 | python -c 'import json,sys;print(round(json.loads(sys.stdin.read())["ms_per_step"],4))')" | tee -a $O/${TAG}_ab.txt
done; done
for lib in "$@"; do
  rm -rf /tmp/abp; CRANK_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/abp -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > /tmp/abp.log 2>&1 || tail -3 /tmp/abp.log
  f=$(find /tmp/abp -name "*kernel_trace.csv" | head -1)
  python - "$f" "$(basename $lib)" "$RE" <<'PY' | tee -a $O/${TAG}_ab.txt
import csv, sys, collections, re
acc = collections.defaultdict(list)
rx = re.compile(sys.argv[3])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if rx.search(n):
        acc[n.split("(")[0][:60] + " grid " + r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[2])
for k, v in sorted(acc.items()):
    print(f"   {k:84s} n={len(v):5d} avg {sum(v)/len(v):7.1f} us")
PY
done
