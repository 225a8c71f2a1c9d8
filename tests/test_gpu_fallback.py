"""The per-layer fallback kernels (conv_tile_kernel, the table weight-gradient kernel: what runs for shapes the
fused stack / chain kernels refuse, and under CRK_NO_FUSE=1) against the fused kernels on the same weights and
inputs, in bf16x3 arithmetic where both are ~fp32 accurate: generator with all four gated stacks, the 1x1 chains
and the conditioning input; the gated discriminator; a plain conv stack."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import REPO

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import sys, torch, numpy as np
sys.path.insert(0, %r)
from crank_amd import ops
from crank_amd.bin.train import get_model
from crank_amd.net.module.pwg import ParallelWaveGANDiscriminator, ResidualParallelWaveGANDiscriminator
from crank_amd.utils import load_yaml
from tests.helpers import fill_models, make_batch
ops.set_precision("bf16x3")
out = {}
conf = load_yaml(None, batch_size=2, batch_len=96)
G = get_model(conf, 3, "cuda")["G"].train()
fill_models({"G": G})
b = make_batch(2, 96, 3, device="cuda", seed=5)
dec_h = torch.cat([b["lcf0"], b["uv"]], -1)
h = b["org_h"].clone(); h[:, :] = h[:, 0:1]
x = b["in_feats"].clone().requires_grad_(True)
o = G(x, None, dec_h, spkrvec=h, use_ema=False)
gen = torch.Generator().manual_seed(2)
(o["decoded"] * torch.randn(o["decoded"].shape, generator=gen).cuda()).sum().backward()
torch.cuda.synchronize()
out["G_decoded"], out["G_dx"], out["G_gp"] = o["decoded"].detach().cpu().numpy(), x.grad.cpu().numpy(), G.grad_flat.cpu().numpy()
out["G_qidx"] = torch.stack(o["qidx"]).cpu().numpy()
for name, net, cin in [("D", ResidualParallelWaveGANDiscriminator(in_channels=37, out_channels=1, kernel_size=5, layers=4, stacks=2, dropout=0.0), 37),
                       ("C", ParallelWaveGANDiscriminator(in_channels=20, out_channels=6, kernel_size=3, layers=4), 20)]:
    fill_models({name: net})
    xx = torch.randn(2, cin, 90, generator=gen).cuda().requires_grad_(True)
    y = net(xx)
    (y * torch.randn(y.shape, generator=gen).cuda()).sum().backward()
    torch.cuda.synchronize()
    out[name + "_y"], out[name + "_dx"], out[name + "_gp"] = y.detach().cpu().numpy(), xx.grad.cpu().numpy(), net.grad_flat.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_per_layer_fallback_kernels_agree_with_the_fused_ones(tmp_path):
    outs = {}
    for mode in ("0", "1"):
        f = tmp_path / f"nofuse{mode}.npz"
        r = subprocess.run([sys.executable, "-c", _SCRIPT % REPO, str(f)], env=dict(os.environ, CRK_NO_FUSE=mode),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(f)
    assert (outs["0"]["G_qidx"] == outs["1"]["G_qidx"]).mean() > 0.999
    for k in outs["0"].files:
        if k == "G_qidx":
            continue
        a, b = outs["0"][k], outs["1"][k]
        assert np.isfinite(b).all(), k
        err = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-30))
        print(k, err)
        assert err < 1e-3, (k, err)
