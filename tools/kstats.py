"""Print per-kernel average durations from a rocprofv3 --kernel-trace CSV directory: python tools/kstats.py DIR [filter]"""
import collections
import csv
import glob
import sys

BY_GRID = "--by-grid" in sys.argv
if BY_GRID:
    sys.argv.remove("--by-grid")
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:70]
        if BY_GRID:
            name = name[:48] + f' g{r.get("Grid_Size_X", r.get("Grid_Size", "?"))}x{r.get("Grid_Size_Y", "")} lds{r.get("LDS_Block_Size", "?")}'
        acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tot = sum(sum(v) for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if flt in k:
        print(f"{k:72s} n {len(v):5d} avg {sum(v)/len(v):8.1f} min {min(v):8.1f} us  total {sum(v)/1e3:8.2f} ms ({100*sum(v)/tot:4.1f} %)")
