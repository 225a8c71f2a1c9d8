#!/bin/bash
# How much of the stack kernels' plane traffic reaches HBM?  FETCH_SIZE / WRITE_SIZE count the L2's fabric requests, and the
# 256 MiB Infinity Cache sits behind the fabric: a kernel that re-reads what its producer wrote a few tens of microseconds
# earlier may never touch DRAM.  A/B: one stacks_alone pass (tools/prof_stacks_alone.py) as it is, and with a read-modify-write
# pass over 768 MiB in front of EVERY stack kernel (crk_debug_flush_before), per-kernel durations from rocprofv3's kernel trace.
# What a kernel loses with the pass in place is what the cache was giving it.  Calibrated with tools/mall_calib.py.
#   bash tools/mall_ab.sh OUT.txt      (on the GPU box)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/mall_ab.txt}
: > $OUT
for f in 0 1; do
  rm -rf /tmp/mc_$f
  rocprofv3 --kernel-trace --output-format csv -d /tmp/mc_$f -- python $ROOT/tools/mall_calib.py $f > /tmp/mc_$f.log 2>&1 || tail -3 /tmp/mc_$f.log
done
for mb in 0 768; do
  rm -rf /tmp/ma_$mb
  CRK_FLUSH_MB=$mb rocprofv3 --kernel-trace --output-format csv -d /tmp/ma_$mb -- python $ROOT/tools/prof_stacks_alone.py 6 > /tmp/ma_$mb.log 2>&1 || tail -3 /tmp/ma_$mb.log
done
python - >> $OUT <<'PY'
import csv, glob, collections, statistics
def trace(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows
print("calibration: in-place x += 1 over S MB (reads S, writes S), median us per launch, launches 3.. of 12")
print(f"{'S MB':>6s} {'back to back':>14s} {'TB/s':>6s} {'768 MiB pass between':>22s} {'TB/s':>6s}")
cal = {}
for f in (0, 1):
    by = collections.defaultdict(list)
    for r in trace(f"/tmp/mc_{f}"):
        if "elementwise" not in r["Kernel_Name"]: continue
        g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
        by[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cal[f] = by
sizes = (8, 32, 64, 128, 192, 512)
g0 = sorted(cal[0])
g1 = [g for g in sorted(cal[1])]
for i, mb in enumerate(sizes):
    a = statistics.median(cal[0][g0[i]][2:]) if i < len(g0) else float("nan")
    # with the pass in between the 768 MiB kernel has the largest grid; the i-th smallest grid is size i
    b = statistics.median(cal[1][g1[i]][2:]) if i < len(g1) else float("nan")
    print(f"{mb:6d} {a:14.1f} {2 * mb * 1.048576 / a:6.2f} {b:22.1f} {2 * mb * 1.048576 / b:6.2f}")
print()
res = {}
for mb in (0, 768):
    by = collections.defaultdict(list)
    for r in trace(f"/tmp/ma_{mb}"):
        k = r["Kernel_Name"].split("(")[0]
        by[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res[mb] = by
print("stacks_alone pass, per kernel: launches, mean us as the pass runs | with a 768 MiB pass in front of every stack kernel")
tot = [0.0, 0.0]
for k in sorted(res[0], key=lambda k: -sum(res[0][k])):
    if "flush" in k or k not in res[768]: continue
    a, b = res[0][k], res[768][k]
    if sum(a) < 50: continue
    ma, mb_ = statistics.mean(a[len(a) // 4:]), statistics.mean(b[len(b) // 4:])
    print(f"{k[:70]:70s} n {len(a):4d} {ma:8.1f} | {mb_:8.1f}  ({(mb_ / ma - 1) * 100:+.0f} %)")
PY
cat $OUT
