"""Timing of the fused reconstruction-loss launch at the step's shape, by part: python tools/time_recon.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crank_amd import ops

B, T, D = 64, 500, 80
x = torch.randn(B, T, D, device="cuda")
y = x + 0.3 * torch.randn(B, T, D, device="cuda")
m = torch.rand(B, T, device="cuda") > 0.1


def win(res):
    return [torch.hann_window(w, dtype=torch.float32, device="cuda") for _, _, w in res]


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


CASES = (("both", [(64, 64, 16), (128, 128, 32)]), ("res1 only", [(64, 64, 16)]), ("res2 only", [(128, 128, 32)]))
if len(sys.argv) > 1:  # one case per process (under rocprofv3 --kernel-trace --stats: the kernels' own durations)
    CASES = CASES[int(sys.argv[1]): int(sys.argv[1]) + 1]
for tag, res in CASES:
    w = win(res)
    xg = x.clone().requires_grad_(True)

    def fwd_grad():
        return ops.recon_loss(xg, y, m, res, w, 0.0)

    def fwd_nograd():
        with torch.no_grad():
            return ops.recon_loss(x, y, m, res, w, 0.0)

    def fwd_bwd():
        v = ops.recon_loss(xg, y, m, res, w, 0.0)
        (2.0 * v[0] + v[2]).backward()

    print(f"{tag:10s}: forward+unit gradient {timeit(fwd_grad):6.1f} us   forward only {timeit(fwd_nograd):6.1f} us   forward+backward {timeit(fwd_bwd):6.1f} us")
