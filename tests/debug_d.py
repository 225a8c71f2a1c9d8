"""Ad-hoc: residual discriminator in bf16 mode, y / dx errors vs oracle (race hunting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crank_amd import ops
from crank_amd.net.module.pwg import ResidualParallelWaveGANDiscriminator
from oracle import pwg
from tests.test_gpu_nets import _load_same
ops.set_precision("bf16")
cfg = dict(in_channels=113, out_channels=1, kernel_size=5, layers=8, stacks=4)
prod, orac = ResidualParallelWaveGANDiscriminator(**cfg, dropout=0.0), pwg.ResidualParallelWaveGANDiscriminator(**cfg, dropout=0.0)
_load_same(prod, orac)
B, T = 2, 200
rs = np.random.RandomState(100)
x = torch.from_numpy(rs.standard_normal((B, 113, T)).astype(np.float32))
xo = x.clone().requires_grad_(True); yo = orac(xo)
dy = torch.from_numpy(rs.standard_normal(tuple(yo.shape)).astype(np.float32))
(yo * dy).sum().backward()
outs = []
for rep in range(3):
    xp = x.cuda().requires_grad_(True); prod.zero_grad(); yp = prod(xp); (yp * dy.cuda()).sum().backward()
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.cpu() - b).abs().max() / b.abs().max()).item()
    print(f"rep {rep}: y {rel(yp.detach(), yo.detach()):.3e} dx {rel(xp.grad, xo.grad):.3e}")
    outs.append((yp.detach().cpu(), xp.grad.cpu()))
print("y deterministic", all(torch.equal(outs[0][0], o[0]) for o in outs), "dx deterministic", all(torch.equal(outs[0][1], o[1]) for o in outs))
d = (outs[0][1] - xo.grad).abs().amax(dim=1)  # (B,T) per-frame dx error
print("frames with large dx error:", torch.nonzero(d > 0.05 * xo.grad.abs().max()).tolist()[:40])
# determinism / coverage of the saved planes
L = 8; P = B * T * 64
def planes():
    y = prod(x.cuda().requires_grad_(True)); torch.cuda.synchronize()
    fn = y.grad_fn
    while fn is not None and not hasattr(fn, "saved_ws"):
        fn = fn.next_functions[0][0]
    return fn.saved_ws.clone()
a, b2 = planes(), planes()
f32 = (4 * L + 2) * P
for name, lo, hi in (("X0", 0, P), ("TA", L * P, 2 * L * P), ("SB", 2 * L * P, 3 * L * P), ("SKIP", 4 * L * P, (4 * L + 1) * P)):
    print(name, "equal across runs", torch.equal(a[lo:hi], b2[lo:hi]), "checksum %.6f" % a[lo:hi].double().abs().sum().item(),
          "nan", torch.isnan(a[lo:hi]).any().item())
h = a[f32:].view(torch.int16)
h2 = b2[f32:].view(torch.int16)
for name, k in (("Xb_hi", 0), ("Zb_hi", 2)):
    s0 = h[k * L * P:(k + 1) * L * P]; s1 = h2[k * L * P:(k + 1) * L * P]
    print(name, "equal across runs", torch.equal(s0, s1), "checksum %.6f" % s0.view(torch.bfloat16).double().abs().sum().item())
sb_a = a[2 * L * P:3 * L * P].view(L, B, T, 64); sb_b = b2[2 * L * P:3 * L * P].view(L, B, T, 64)
ta_a = a[L * P:2 * L * P].view(L, B, T, 64)
zb = h[2 * L * P:3 * L * P].view(torch.bfloat16).float().view(L, B, T, 64)
# which run is wrong? z = ta*sb must hold
for nm, sbx in (("run a", sb_a), ("run b", sb_b)):
    bad = torch.nonzero(((ta_a * sbx) - zb).abs() > 0.02)
    print(nm, "SB entries inconsistent with z = ta*sb:", len(bad), bad[:12].tolist())
    if len(bad):
        print("   layers", sorted(set(bad[:, 0].tolist())), "frames", sorted(set(bad[:, 2].tolist()))[:40], "channels", sorted(set(bad[:, 3].tolist()))[:64])
