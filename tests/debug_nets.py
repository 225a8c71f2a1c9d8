"""Ad-hoc GPU diagnostics (not collected by pytest): error tables for conv stacks."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from crank_amd import ops  # noqa: E402
from crank_amd.net.module.pwg import ParallelWaveGANDiscriminator, ResidualParallelWaveGANDiscriminator  # noqa: E402
from oracle import pwg  # noqa: E402
from tests.test_gpu_nets import _load_same, _rel  # noqa: E402


def run(kind, cfg, B, T, precision="bf16x3", verbose=False):
    ops.set_precision(precision)
    if kind == 2:
        prod, orac = ParallelWaveGANDiscriminator(**cfg), pwg.ParallelWaveGANDiscriminator(**cfg)
    else:
        prod, orac = ResidualParallelWaveGANDiscriminator(**cfg), pwg.ResidualParallelWaveGANDiscriminator(**cfg)
    _load_same(prod, orac)
    rs = np.random.RandomState(3)
    cin = cfg["in_channels"]
    x = torch.from_numpy(rs.standard_normal((B, cin, T)).astype(np.float32))
    xo = x.clone().requires_grad_(True)
    yo = orac(xo)
    dy = torch.from_numpy(rs.standard_normal(tuple(yo.shape)).astype(np.float32))
    (yo * dy).sum().backward()
    xp = x.cuda().requires_grad_(True)
    prod.zero_grad()
    yp = prod(xp)
    (yp * dy.cuda()).sum().backward()
    torch.cuda.synchronize()
    errs = {"y": _rel(yp, yo), "dx": _rel(xp.grad, xo.grad)}
    for k, p in orac.named_parameters():
        if p.grad is not None:
            errs["d" + k] = _rel(prod.grad_view(k), p.grad)
    bad = {k: f"{v:.1e}" for k, v in errs.items() if v > 1e-3}
    print(f"kind{kind} {cfg} B={B} T={T} {precision}: y {errs['y']:.1e} dx {errs['dx']:.1e} n_bad {len(bad)}")
    if bad:
        print("   BAD:", bad)
    # where is dx wrong (frame positions)?
    if errs["dx"] > 1e-3:
        d = (xp.grad.cpu() - xo.grad).abs().amax(dim=1)  # (B,T)
        scale = xo.grad.abs().max()
        for b in range(B):
            pos = torch.nonzero(d[b] > 1e-3 * scale).flatten().tolist()
            print(f"   dx wrong frames b={b}: {pos[:40]}{'...' if len(pos) > 40 else ''} ({len(pos)} of {T})")


if __name__ == "__main__":
    for k in [3, 5]:
        for layers in [1, 2, 3]:
            for T in [17, 64, 65, 128, 150]:
                run(2, dict(in_channels=64, out_channels=64, kernel_size=k, layers=layers, conv_channels=64), 2, T)
    run(2, dict(in_channels=80, out_channels=14, kernel_size=5, layers=4, conv_channels=64), 2, 128)
    run(2, dict(in_channels=80, out_channels=14, kernel_size=5, layers=8, conv_channels=64), 3, 150)
    for T in [17, 150]:
        run(1, dict(in_channels=67, out_channels=15, kernel_size=3, layers=2, stacks=1), 2, T)
        run(1, dict(in_channels=113, out_channels=1, kernel_size=5, layers=8, stacks=4), 2, T)
