"""VQ search kernels at the benchmark shape: indices of the three kernels against each other and their times.
    python tools/time_vq.py        (spawns itself once per CRK_VQ_LC value)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if len(sys.argv) > 1:
    import numpy as np
    import torch
    from crank_amd import ops

    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 500, 64, generator=g).cuda()
    w = (torch.randn(512, 64, generator=g) * 0.7).cuda()
    w[100] = w[7]
    x[0, :50] = w[7] + 1e-3 * torch.randn(50, 64, generator=g).cuda()
    for _ in range(3):
        e, qx, idx = ops.vq_apply(x, w)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50):
        ops.vq_apply(x, w)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"CRK_VQ_LC={os.environ.get('CRK_VQ_LC')}: {ev[0].elapsed_time(ev[1]) / 50 * 1e3:.1f} us per call (events, incl. launch)")
    np.savez(sys.argv[1], idx=idx.cpu().numpy(), e=e.cpu().numpy(), qx=qx.cpu().numpy())
else:
    import numpy as np

    outs = {}
    for v in ("0", "1", "2"):
        f = f"/tmp/vq_{v}.npz"
        subprocess.run([sys.executable, os.path.abspath(__file__), f], env=dict(os.environ, CRK_VQ_LC=v), check=True)
        outs[v] = np.load(f)
    for v in ("0", "1"):
        same = all(np.array_equal(outs["2"][k], outs[v][k]) for k in ("idx", "e", "qx"))
        print(f"MFMA kernel vs CRK_VQ_LC={v}: indices, gathered vectors, straight-through values bit-identical: {same}",
              int((outs["2"]["idx"] != outs[v]["idx"]).sum()), "index differences")
