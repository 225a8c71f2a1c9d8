"""Per-basic-block instruction mix of one kernel in an AMDGPU assembly dump (hipcc -S --cuda-device-only):
which loop a block sits in and how many MFMA / LDS / VALU / move / wait instructions it holds.  Used to spot
register-copy storms and exposed waits in the inner loops without a GPU.
usage: isa_blocks.py file.s kernel-name-substring [min-instructions]"""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and name in l)
end = next(j for j in range(start, len(lines)) if "s_endpgm" in lines[j])


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "mov"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "dsr"
    if op.startswith("ds_"): return "dsw"
    if op.startswith(("buffer_load", "global_load")): return "gld"
    if op.startswith(("buffer_store", "global_store")): return "gst"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


blocks, cur = [], ["entry", "", collections.Counter()]
for l in lines[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
    if m:
        blocks.append(cur)
        d = re.search(r"Depth=(\d+)", l)
        cur = [m.group(1), ("depth " + d.group(1)) if d else "", collections.Counter()]
        continue
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")):
        continue
    cur[2][cls(t[0])] += 1
blocks.append(cur)
tot = collections.Counter()
for n_, d, c in blocks:
    tot.update(c)
    if sum(c.values()) >= minn:
        print(f"{n_:12s} {d:8s} n={sum(c.values()):4d} ", " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("total", dict(tot))
