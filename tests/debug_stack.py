"""Ad-hoc GPU diagnostic for the fused stack forward: compare the saved planes of a small
gated-residual discriminator against the oracle's intermediates (not collected by pytest)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from crank_amd import ops
from crank_amd.net.module.pwg import ResidualParallelWaveGANDiscriminator
from oracle import pwg
from tests.test_gpu_nets import _load_same

ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = dict(in_channels=20, out_channels=3, kernel_size=3, layers=L, stacks=1)
prod, orac = ResidualParallelWaveGANDiscriminator(**cfg), pwg.ResidualParallelWaveGANDiscriminator(**cfg)
_load_same(prod, orac)
B, T = 2, 100
x = torch.from_numpy(np.random.RandomState(0).standard_normal((B, 20, T)).astype(np.float32))
xp = x.cuda().requires_grad_(True)
y = prod(xp)
torch.cuda.synchronize()
fn = y.grad_fn
while fn is not None and not hasattr(fn, "saved_ws"):
    fn = fn.next_functions[0][0]
ws = fn.saved_ws.cpu()
N = B * T
P = N * 64
planes = lambda k: ws[k * P:(k + 1) * P].view(B, T, 64)
# oracle intermediates
with torch.no_grad():
    h = orac.first_conv(x)
    ref = {"X0": h.transpose(1, 2)}
    skips = 0
    for l, blk in enumerate(orac.conv_layers):
        g = blk.conv(h)
        xa, xb = g.split(64, dim=1)
        ta, sb = torch.tanh(xa), torch.sigmoid(xb)
        z = ta * sb
        s = blk.conv1x1_skip(z)
        h = (blk.conv1x1_out(z) + h) * math.sqrt(0.5)
        skips = skips + s
        ref[f"TA{l}"], ref[f"SB{l}"], ref[f"Z{l}"] = ta.transpose(1, 2), sb.transpose(1, 2), z.transpose(1, 2)
        if l + 1 < L:
            ref[f"X{l + 1}"] = h.transpose(1, 2)
    ref["SKIP"] = skips.transpose(1, 2)
got = {"X0": planes(0), "SKIP": planes(4 * L)}
for l in range(L):
    got[f"TA{l}"], got[f"SB{l}"], got[f"Z{l}"] = planes(L + l), planes(2 * L + l), planes(3 * L + l)
    if l + 1 < L:
        got[f"X{l + 1}"] = planes(l + 1)
for k in ref:
    d = (got[k] - ref[k]).abs()
    e = d.max().item() / (ref[k].abs().max().item() + 1e-12)
    bad = torch.nonzero(d > 1e-3 * ref[k].abs().max())
    print(f"{k}: rel err {e:.2e}; wrong elements {len(bad)} of {d.numel()}", bad[:6].tolist() if len(bad) else "")
print("y err", ((y.detach().cpu() - orac(x)).abs().max() / orac(x).abs().max()).item())
# is the result a permutation of the reference? (frame 5 of utterance 0)
g, r = got["TA0"][0, 5], ref["TA0"][0, 5]
print("got ", [round(v, 3) for v in g[:16].tolist()])
print("ref ", [round(v, 3) for v in r[:16].tolist()])
for c in range(8):
    j = int((r - g[c]).abs().argmin())
    print(f"got[ch {c}] = {g[c]:.4f} closest ref channel {j} ({r[j]:.4f})")
# maybe frames are permuted: compare got[0, f, 0] with ref[0, :, 0]
for f in range(6):
    j = int((ref["TA0"][0, :, 0] - got["TA0"][0, f, 0]).abs().argmin())
    print(f"got[frame {f}, ch 0] = {got['TA0'][0, f, 0]:.4f} closest ref frame {j} ({ref['TA0'][0, j, 0]:.4f})")
