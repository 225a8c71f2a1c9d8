"""The four generator stacks forward + backward in isolation at the benchmark shape (what bench.py's `stacks_alone`
times), enqueued eagerly so that a kernel trace shows every launch:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/prof_stacks_alone.py [iters]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from crank_amd import ops  # noqa: E402
from crank_amd.bin.train import get_model  # noqa: E402
from crank_amd.utils import load_yaml  # noqa: E402

ops.set_precision("bf16")
if int(os.environ.get("CRK_FLUSH_MB", "0")) > 0:  # tools/mall_ab.sh: a > Infinity-Cache read-modify-write pass before every stack kernel
    from crank_amd import _lib
    assert _lib.lib().crk_debug_flush_before(int(os.environ["CRK_FLUSH_MB"]) << 20) == 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, T = 64, 500
torch.manual_seed(0)
G = get_model(load_yaml(None, batch_size=B, batch_len=T), 14, "cuda")["G"]
ins = []
for st in list(G.encoders) + list(G.decoders):
    x = torch.randn(B, T, st.net.in_ch, device="cuda", requires_grad=True)
    c = torch.randn(B, T, st.net.aux_ch, device="cuda") if st.net.aux_ch > 0 else None
    ins.append((st, x, c))
ones = {}
for it in range(iters + 2):
    G.defer_wnorm = True
    for st, x, c in ins:
        y = st(x, c=c) if c is not None else st(x)
        if tuple(y.shape) not in ones:
            ones[tuple(y.shape)] = torch.ones_like(y)
        torch.autograd.grad(y, x, ones[tuple(y.shape)])
    G.defer_wnorm = False
    G.finish_grads()
torch.cuda.synchronize()
print("done")
