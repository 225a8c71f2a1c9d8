"""The kernel launches of ONE replayed training step in execution order, from a rocprofv3 --kernel-trace CSV:
python tools/step_sequence.py KERNEL_TRACE.csv OUT.txt   (durations, the idle gap in front of each launch, grid / LDS)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchors = [i for i, r in enumerate(rows) if "recon_fwd_kernel" in r["Kernel_Name"]]
a, b = anchors[-2], anchors[-1]
prev_end, tot, lines = None, 0.0, []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    tot += (e - s) / 1e3
    lines.append("%8.1f us  gap %6.1f  grid %8s wg %4s lds %7s  %s" % ((e - s) / 1e3, gap, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"),
                                                                     r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:100]))
    prev_end = e
span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
with open(sys.argv[2], "w") as f:
    f.write("%d launches, %.1f us of kernel time, %.1f us from anchor to anchor\n" % (b - a, tot, span))
    f.write("\n".join(lines) + "\n")
print(b - a, "launches in the step,", round(tot, 1), "us of kernel time,", round(span, 1), "us span")
