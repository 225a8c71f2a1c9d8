"""Ad-hoc GPU diagnostic: run-to-run determinism and error localisation of an 8-layer
plain stack (not collected by pytest)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from crank_amd import ops  # noqa: E402
from crank_amd.net.module.pwg import ParallelWaveGANDiscriminator  # noqa: E402
from oracle import pwg  # noqa: E402
from tests.test_gpu_nets import _load_same  # noqa: E402

ops.set_precision("bf16x3")
cfg = dict(in_channels=80, out_channels=14, kernel_size=5, layers=8, conv_channels=64)
prod, orac = ParallelWaveGANDiscriminator(**cfg), pwg.ParallelWaveGANDiscriminator(**cfg)
_load_same(prod, orac)
for (B, T) in [(3, 150), (3, 150), (1, 450), (2, 150), (3, 130), (4, 100), (3, 150)]:
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.standard_normal((B, 80, T)).astype(np.float32))
    xo = x.clone().requires_grad_(True)
    orac.zero_grad()
    yo = orac(xo)
    dy = torch.from_numpy(rs.standard_normal(tuple(yo.shape)).astype(np.float32))
    (yo * dy).sum().backward()
    runs = []
    for r in range(3):
        xp = x.cuda().requires_grad_(True)
        prod.zero_grad()
        yp = prod(xp)
        (yp * dy.cuda()).sum().backward()
        torch.cuda.synchronize()
        runs.append((xp.grad.clone(), prod.grad_flat.clone()))
    same = [torch.equal(runs[0][0], runs[i][0]) and torch.equal(runs[0][1], runs[i][1]) for i in (1, 2)]
    d = (runs[0][0].cpu() - xo.grad).abs().amax(dim=1)
    scale = xo.grad.abs().max()
    bad = [(b, torch.nonzero(d[b] > 1e-3 * scale).flatten().tolist()) for b in range(B)]
    bad = [(b, (p[0], p[-1], len(p))) for b, p in bad if p]
    print(f"B={B} T={T}: deterministic {same} dx err {(d.max() / scale).item():.1e} wrong (b,(first,last,count)) {bad}")
