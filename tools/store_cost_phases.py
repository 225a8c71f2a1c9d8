"""Where the plane stores' time lands: phase cycles of the LAST stack2_fwd_kernel launch of a generator forward (dec0) without /
with saved planes, and of the LAST stack2_bwd_kernel launch of a stacks-alone pass (enc0), from instrumented builds
(tools/build_variant.sh prof_f stack2_kernels.hip -DS2_PROF; prof_b stack2b_kernels.hip -DS2B_PROF;
prof_bns stack2b_kernels.hip "-DS2B_PROF -DS2B_ABL=32").   CRANK_AMD_LIB=<lib> python tools/store_cost_phases.py fwd|bwd"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from crank_amd import _lib, ops  # noqa: E402
from crank_amd.bin.train import get_model  # noqa: E402
from crank_amd.synthetic import make_batch  # noqa: E402
from crank_amd.utils import load_yaml  # noqa: E402

ops.set_precision("bf16")
L = _lib.lib()
B, T = 64, 500
torch.manual_seed(0)
conf = load_yaml(None, batch_size=B, batch_len=T)
m = get_model(conf, 14, "cuda")
G = m["G"]
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if what == "fwd":
    L.crk_debug_s2_prof.argtypes = [ctypes.c_void_p]
    names = ["taps", "gate", "wait A", "1x1+upd", "operand", "wait B", "prologue barrier", "TOTAL", "pro: first conv / state", "pro: tables", "pro: cond tile", "pro: operand put", "pro: bias req", "pro: other req", "pro: x -> LDS", "pro: x barrier"]
    b = make_batch(B, T, 14, device="cuda")
    dec_h = torch.cat([b["lcf0"], b["uv"]], -1)
    h = b["org_h"].clone(); h[:, :] = h[:, 0:1]
    for mode in ("nograd", "fwd", "nograd", "fwd"):
        for _ in range(4):
            with torch.set_grad_enabled(mode != "nograd"):
                o = G(b["in_feats"], None, dec_h, spkrvec=h)
        torch.cuda.synchronize()
        buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
        assert L.crk_debug_s2_prof(buf.ctypes.data) == 0
        v = buf.reshape(256, 8, 16).astype(np.float64)
        mean = v.mean(axis=(0, 1))
        print(f"dec0 forward, {mode:6s}: " + "  ".join(f"{n} {mean[i]:7.0f}" for i, n in enumerate(names)))
        for w in (0, 2, 4, 6):
            print(f"      wave {w}: " + "  ".join(f"{n} {v[:, w, i].mean():7.0f}" for i, n in enumerate(names[:8])))
else:
    L.crk_debug_s2b_prof.argtypes = [ctypes.c_void_p]
    names = ["prologue", "P1 1x1+gate", "wait A", "taps(rest)", "dX epi", "wait B", "first conv", "TOTAL", "step0", "steps1-8", "steps9-16", "-"]
    ins = []
    for st in list(G.encoders) + list(G.decoders):
        x = torch.randn(B, T, st.net.in_ch, device="cuda", requires_grad=True)
        c = torch.randn(B, T, st.net.aux_ch, device="cuda") if st.net.aux_ch > 0 else None
        ins.append((st, x, c))
    ones = {}
    order = [ins[1], ins[2], ins[3], ins[0]]  # enc0 last: its backward is the last stack2_bwd launch
    for it in range(4):
        G.defer_wnorm = True
        for st, x, c in order:
            y = st(x, c=c) if c is not None else st(x)
            if tuple(y.shape) not in ones:
                ones[tuple(y.shape)] = torch.ones_like(y)
            torch.autograd.grad(y, x, ones[tuple(y.shape)])
        G.defer_wnorm = False
        G.finish_grads()
    torch.cuda.synchronize()
    buf = np.zeros(256 * 4 * 12, dtype=np.uint64)
    assert L.crk_debug_s2b_prof(buf.ctypes.data) == 0
    v = buf.reshape(256, 4, 12).astype(np.float64)
    mean = v.mean(axis=(0, 1))
    layers = 8
    print(f"enc0 backward ({os.path.basename(os.environ.get('CRANK_AMD_LIB', 'product'))}): " + "  ".join(f"{n} {mean[i]:7.0f}" for i, n in enumerate(names[:11])))
    print("   per block: " + " ".join(f"{names[i]} {mean[i] / layers:.0f}" for i in (1, 2, 8, 9, 10, 3, 4, 5)))
