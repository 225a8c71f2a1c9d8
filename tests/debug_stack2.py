"""Ad-hoc: fused-stack debug dumps (CRK_SK_DBG=1: TA<-res, SB<-xs readback; =2: TA<-acc, SB<-bias)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from crank_amd import ops
from crank_amd.net.module.pwg import ResidualParallelWaveGANDiscriminator
from oracle import pwg
from tests.test_gpu_nets import _load_same

ops.set_precision("bf16x3")
mode = int(os.environ.get("CRK_SK_DBG", "0"))
cfg = dict(in_channels=20, out_channels=3, kernel_size=3, layers=1, stacks=1)
prod, orac = ResidualParallelWaveGANDiscriminator(**cfg), pwg.ResidualParallelWaveGANDiscriminator(**cfg)
_load_same(prod, orac)
B, T = 2, 100
x = torch.from_numpy(np.random.RandomState(0).standard_normal((B, 20, T)).astype(np.float32))
y = prod(x.cuda().requires_grad_(True))
torch.cuda.synchronize()
fn = y.grad_fn
while fn is not None and not hasattr(fn, "saved_ws"):
    fn = fn.next_functions[0][0]
ws = fn.saved_ws.cpu()
P = B * T * 64
planes = lambda k: ws[k * P:(k + 1) * P].view(B, T, 64)
X0, TA, SB = planes(0), planes(1), planes(2)
with torch.no_grad():
    h = orac.first_conv(x)
    blk = orac.conv_layers[0]
    pre = blk.conv(h).transpose(1, 2)  # (B,T,128) incl. bias
    bias = blk.conv.bias
def rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
if mode == 1:
    print("res vs X0", rel(TA, X0), " xs readback vs X0", rel(SB, X0))
    bad = torch.nonzero((TA - X0).abs() > 1e-3)
    print("bad res", len(bad), bad[:8].tolist())
    bad = torch.nonzero((SB - X0).abs() > 2e-2 * X0.abs().max())
    print("bad xs", len(bad), bad[:8].tolist())
elif mode == 2:
    want = pre[..., :64] - bias[:64]
    print("acc vs conv(no bias)", rel(TA, want))
    print("bias as loaded vs bias", rel(SB, bias[:64].expand_as(SB)))
    bad = torch.nonzero((TA - want).abs() > 1e-3 * want.abs().max())
    print("bad acc", len(bad), bad[:8].tolist())
    print("got", TA[0, 5, :8].tolist()); print("ref", want[0, 5, :8].tolist())
    # is got[frame] the reference of another frame / channel?
    for f in (5, 6):
        d = (want[0] - TA[0, f, 0]).abs()
        print("got[f=%d,c=0] closest ref (frame,ch)" % f, np.unravel_index(int(d.argmin()), d.shape), d.min().item())
