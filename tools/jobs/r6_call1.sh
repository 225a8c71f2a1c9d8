#!/bin/bash
# round 6, call 1: baseline of HEAD (bench line), the Infinity-Cache A/B of the stack kernels, counter list
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
(cd /tmp && timeout 120 rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_a-z]*\|MALL[A-Z0-9_a-z]*\|HBM[A-Z0-9_a-z]*" | sort -u > $O/r6_counters_tcc.txt); wc -l $O/r6_counters_tcc.txt
timeout 900 python bench.py 2> $O/r6_c1_bench.err | tail -1 > $O/r6_c1_bench_line.json; head -c 600 $O/r6_c1_bench_line.json; echo
timeout 900 bash tools/mall_ab.sh $O/r6_mall_ab.txt > $O/r6_mall_ab.log 2>&1; tail -40 $O/r6_mall_ab.txt
# DRAM-destined requests of one stacks_alone pass next to all fabric requests
cd /tmp
for c in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum; do
  rm -rf /tmp/dr_$c; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/dr_$c -- python $GRAFT_REPO_ROOT/tools/prof_stacks_alone.py 3 > /tmp/dr_$c.log 2>&1 || tail -3 /tmp/dr_$c.log
done
python - > $O/r6_dram_counters.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_DRAM_sum"):
    for f in glob.glob(f"/tmp/dr_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][c] += float(r["Counter_Value"])
print("requests summed over 5 passes: RDREQ, RDREQ_DRAM, WRREQ, WRREQ_DRAM")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["TCC_EA0_RDREQ_sum"])[:14]:
    print(f"{k:60s} {v['TCC_EA0_RDREQ_sum']:14.0f} {v['TCC_EA0_RDREQ_DRAM_sum']:14.0f} {v['TCC_EA0_WRREQ_sum']:14.0f} {v['TCC_EA0_WRREQ_DRAM_sum']:14.0f}")
PY
cat $O/r6_dram_counters.txt
