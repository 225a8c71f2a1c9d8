#!/bin/bash
# HBM traffic per kernel launch of the bench workload (separate PMC passes, no tracing besides
# the kernel trace the counter collection needs).  Writes gpurun_out/pmc_traffic.csv
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python /root/repo/bench.py --no-cpu-baseline --no-roofline --no-extras --no-graph --steps 4 --warmup 2 > /tmp/pmc_$c.log 2>&1 || tail -3 /tmp/pmc_$c.log
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
with open("/root/repo/gpurun_out/pmc_traffic.csv", "w") as fo:
    fo.write("kernel,launches,FETCH_SIZE_avg_raw,WRITE_SIZE_avg_raw\n")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["FETCH_SIZE"][1]):
        n = max(v["FETCH_SIZE"][0], 1)
        fo.write(f"\"{k}\",{n},{v['FETCH_SIZE'][1] / n:.1f},{v['WRITE_SIZE'][1] / max(v['WRITE_SIZE'][0], 1):.1f}\n")
print(open("/root/repo/gpurun_out/pmc_traffic.csv").read()[:2500])
PY
