"""Calibration of the cache-flush A/B (tools/mall_ab.sh): an in-place `x += 1` over S MB reads what the previous launch of
the same kernel wrote.  Back to back, a buffer that fits the 256 MiB Infinity Cache is served from it; with a pass over a
768 MiB buffer in between it comes from HBM.  Run under `rocprofv3 --kernel-trace`; tools/mall_ab.sh reads the trace and
prints the kernel's duration per size with and without the pass in between."""
import sys

import torch

flush = int(sys.argv[1]) if len(sys.argv) > 1 else 0
big = torch.zeros(768 << 18, device="cuda") if flush else None
for mb in (8, 32, 64, 128, 192, 512):
    x = torch.zeros(mb << 18, device="cuda")
    for _ in range(12):
        x.add_(1.0)
        if flush:
            big.add_(1.0)
    torch.cuda.synchronize()
print("done")
