#!/bin/bash
# Kernel-trace timings of the gated-stack kernels at the benchmark shape: saving / non-saving forwards and the
# backward, for the frame-split (CRK_SK_V=1) and the channel-split (default) kernels.  Run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out; mkdir -p $OUT
for v in ${VARIANTS:-1 2}; do
  for mode in ${MODES:-nograd fwd bwd}; do
    rm -rf "/tmp/kt_${v}_$mode"
    CRK_SK_V=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_${v}_$mode -- python /root/repo/tools/prof_fwd.py $mode 6 > /tmp/kt.log 2>&1 || tail -3 /tmp/kt.log
    f=$(find /tmp/kt_${v}_$mode -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python - "$f" "v$v $mode" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "stack" in n or "vq_" in n or "wnorm" in n:
        acc[(n.split("(")[0][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("VGPR_Count", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v2 = sorted(v)[: max(1, len(v) - len(v) // 6)]  # drop the slowest sixth (first-touch launches)
    print(f"{sys.argv[2]:10s} {k[0]:62s} grid {k[1]:>8s} vgpr {k[2]:>4s} n {len(v):3d} avg {sum(v2)/len(v2):7.1f} min {min(v):7.1f} us")
PY
  done
done
