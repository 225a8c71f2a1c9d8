"""Diagnostic: parameter gradients of the speaker-adversarial net's conv chain (128 -> 64 -> 64 -> 14, k = 3) for one batch of
8 x 96 frames against the same utterances in shards of 4 / 2 / 1 (sums of the shards' gradients), split-operand arithmetic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from crank_amd import ops
from crank_amd.net.module.flat import FlatModel
from crank_amd.net.module.pwg import KIND_PLAIN, HipStack

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
ops.set_precision(prec)
torch.manual_seed(3)


class M(FlatModel):
    def __init__(self):
        super().__init__()
        self.stack = HipStack(KIND_PLAIN, 128, 14, 3, 3, stacks=1, aux_channels=0, bias=True, dropout=0.0)
        self._alloc(self.stack.entries("", 0), self.stack.n_params, "cuda")
        self.stack.bind(self, 0)
        self.stack.init_parameters()


m = M()
B, T = 8, 96
g = torch.Generator().manual_seed(1)
x = torch.randn(B, T, 128, generator=g).cuda()
w = torch.randn(B, T, 14, generator=g).cuda()
lens = [96, 60, 96, 75, 50, 96, 88, 64]
for b, n in enumerate(lens):
    w[b, n:] = 0


def grads(shard, want_dx):
    m.zero_grad()
    for s in range(0, B, shard):
        xi = x[s:s + shard].clone().requires_grad_(want_dx)
        (m.stack(xi) * w[s:s + shard]).sum().backward()
    torch.cuda.synchronize()
    return m.grad_flat.clone()


for want_dx in (True, False):
    ref = grads(8, want_dx)
    for shard in (4, 2, 1):
        gsh = grads(shard, want_dx)
        err = float((gsh - ref).abs().max() / ref.abs().max())
        print(f"{prec} want_dx={want_dx}: shards of {shard} vs one batch of 8: relative max error {err:.2e}")
