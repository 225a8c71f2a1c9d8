"""Scratch probe (not a test): forward time of one generator stack over batch sizes.  Run on a GPU box:
    python tools/occupancy_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crank_amd import ops
from crank_amd.net.module.flat import FlatModel
from crank_amd.net.module.pwg import KIND_GENERATOR, HipStack
class M(FlatModel):
    def __init__(self):
        super().__init__()
        self.stack = HipStack(KIND_GENERATOR, 80, 64, 5, 8, stacks=4, aux_channels=0, bias=True)
        self._alloc(self.stack.entries("", 0), self.stack.n_params, "cuda")
        self.stack.bind(self, 0); self.stack.init_parameters()
def main():
    ops.set_precision("bf16")
    m = M()
    for B in (16, 32, 36, 48, 64, 73, 96, 128):
        x = torch.randn(B, 500, 80, device="cuda")
        with torch.no_grad():
            for _ in range(3): m.stack(x)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(20): m.stack(x)
            ev[1].record(); torch.cuda.synchronize()
        print(os.environ.get("CRK_S2_CFG"), "B", B, "us per stack forward (incl. first conv + head)", ev[0].elapsed_time(ev[1]) / 20 * 1e3)


if __name__ == "__main__":
    main()
