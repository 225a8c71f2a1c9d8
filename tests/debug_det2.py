import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_properties import _full_G
from crank_amd import ops
ops.set_precision("bf16")
model, batch, dec_h, h = _full_G()
G = model["G"].train(); x = batch["in_feats"]
w = torch.randn(64, 500, 80, device="cuda", generator=torch.Generator("cuda").manual_seed(0))
def run():
    G.zero_grad(); xi = x.clone().requires_grad_(True)
    o = G(xi, None, dec_h, spkrvec=h, use_ema=False); (o["decoded"] * w).sum().backward(); torch.cuda.synchronize()
    return G.grad_flat.clone()
a, b = run(), run()
for k, off, shp in G._entries:
    n = 1
    for s_ in shp: n *= s_
    d = (a[off:off + n] - b[off:off + n]).abs().max().item()
    if d > 0: print(k, shp, d, a[off:off + n].abs().max().item())
