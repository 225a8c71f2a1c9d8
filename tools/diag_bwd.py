"""A/B of the two data-gradient chains (CRK_SKB_V=1 frame-split / 2 channel-split) on the bitwise-test shapes: per tensor the
number of differing entries and the largest difference.  python tools/diag_bwd.py"""
import os, subprocess, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests.test_gpu_properties import _V_SCRIPT
outs = {}
with tempfile.TemporaryDirectory() as d:
    for tag, env in (("old", {"CRK_SKB_V": "1"}), ("new", {"CRK_SKB_V": "2"}), ("new2", {"CRK_SKB_V": "2"})):
        f = os.path.join(d, tag + ".npz")
        r = subprocess.run([sys.executable, "-c", _V_SCRIPT % REPO, f], env=dict(os.environ, **env), capture_output=True, text=True)
        if r.returncode:
            print(tag, "FAILED", r.stderr[-2000:]); sys.exit(1)
        outs[tag] = dict(np.load(f))
for k in sorted(outs["old"], key=lambda s: (s[-1], s)):
    a, b = outs["new"][k], outs["new2"][k]
    if int((a != b).sum()):
        print(k, "new vs new2 differing", int((a != b).sum()))
    a, b = outs["old"][k], outs["new"][k]
    diff = np.abs(a - b)
    n = int((a != b).sum())
    print(f"{k:5s} shape {str(a.shape):18s} differing {n:8d} / {a.size:8d}  max diff {diff.max():.3e}  scale {np.abs(a).max():.3e}  nan {int(np.isnan(b).sum())}")
    if n and a.ndim == 3:
        idx = np.argwhere(a != b)
        print("      first differing (b,t,c):", idx[:4].tolist(), " t range", idx[:, 1].min(), idx[:, 1].max(), " distinct t", sorted(set(idx[:, 1].tolist())), "distinct b", sorted(set(idx[:, 0].tolist())))
