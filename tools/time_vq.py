"""Average duration of crk_vq_forward at the benchmark shape (N = 32 000 frames, D = 64, K = 512)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crank_amd import ops  # noqa: E402

torch.manual_seed(0)
x = torch.randn(64, 500, 64, device="cuda")
cb = torch.randn(512, 64, device="cuda") * 0.7
for _ in range(20):
    ops.vq_apply(x, cb)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(200):
    e, qx, idx = ops.vq_apply(x, cb)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 200
print(f"vq_forward (+ output allocation): {us:.1f} us per call, {2 * 32000 * 512 * 64 / us / 1e6:.1f} TFLOP/s fp32, "
      f"{32000 * 520 / us / 1e3:.0f} GB/s algorithmic; idx checksum {int(idx.sum())}")
