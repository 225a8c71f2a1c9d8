#!/bin/bash
# usage: ablate.sh <ENVVAR> <kernel substring> <mode fwd|bwd> values...
cd /tmp && export TMPDIR=/tmp
V=$1; K=$2; M=$3; shift 3
for c in "$@"; do
  rm -rf /tmp/p_$c
  env $V=$c rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -- python /root/repo/tools/prof_fwd.py $M 4 > /tmp/l_$c.log 2>&1
  f=$(find /tmp/p_$c -name "*kernel_stats.csv" | head -1)
  python - "$f" "$K" "$V=$c" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(f"{sys.argv[3]:16s} {r['Name'][:48]:48s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
