#!/bin/bash
# HBM traffic (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, separate PMC passes) of ONE stacks_alone pass: the four
# generator stacks forward + data gradient + weight gradient + weight-norm backward in isolation (tools/prof_stacks_alone.py),
# per kernel and summed per class - the measured side of DESIGN section 4's byte budget.  Run on the GPU box:
#   bash tools/pmc_stacks_traffic.sh OUT.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/pmc_stacks_traffic.txt}
IT=4
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pst_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pst_$c -- python $ROOT/tools/prof_stacks_alone.py $IT > /tmp/pst_$c.log 2>&1 || tail -3 /tmp/pst_$c.log
done
python - $IT > $OUT <<'PY'
import csv, glob, collections, sys
passes = int(sys.argv[1]) + 2
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pst_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"].split("(")[0]][c]
            a[0] += 1; a[1] += float(r["Counter_Value"])
cls = collections.defaultdict(float)
print(f"per stacks_alone pass ({passes} passes averaged); MB = (FETCH_SIZE x 2 + WRITE_SIZE) KB x 1024 / 1e6")
print(f"{'kernel':64s} launches/pass   read MB/launch  written MB/launch   MB/pass")
for k, v in sorted(acc.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'][1] + kv[1]['WRITE_SIZE'][1])):
    n = max(v["FETCH_SIZE"][0], 1)
    rd = 2 * v["FETCH_SIZE"][1] / n * 1024 / 1e6
    wr = v["WRITE_SIZE"][1] / max(v["WRITE_SIZE"][0], 1) * 1024 / 1e6
    tot = (rd + wr) * n / passes
    c = ("forward" if "stack2_fwd" in k or "stack_fwd" in k or "stack2x" in k else "data gradient" if "stack2_bwd" in k or "stack_bwd" in k else
         "weight gradient" if "wgrad" in k else "weight norm" if "wnorm" in k else "other")
    cls[c] += tot
    if tot > 0.5:
        print(f"{k[:64]:64s} {n / passes:8.2f} {rd:16.1f} {wr:16.1f} {tot:12.1f}")
print()
for c, v in sorted(cls.items(), key=lambda kv: -kv[1]):
    print(f"class {c:18s} {v:10.1f} MB per pass")
print(f"sum {sum(cls.values()):10.1f} MB per pass -> {sum(cls.values()) / 6.3e6 * 1e6:.0f} us at 6.3 TB/s")
PY
cat $OUT
