"""Scratch measurement (not a test): the replayed vqvae step with the weight gradients of selected generator stacks on a
second stream (crk_net_set_wgrad_stream), so that they run beside the NEXT stack's data-gradient chain - the k = 3 stacks'
chain is 192 workgroups on 256 compute units.  One trainer per variant, variants alternate, three repetitions.
    python tools/wgrad_stream_ab.py        -> gpurun_out/wgrad_stream_ab.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from crank_amd import _lib, ops, parallel
from crank_amd.bin.train import build_trainer
from crank_amd.net.trainer.basetrainer import GraphedStep
from crank_amd.synthetic import make_batch
from crank_amd.utils import load_yaml

VARIANTS = {"none": (), "dec0": ("dec0",), "dec0+dec1": ("dec0", "dec1"), "all": ("dec0", "dec1", "enc1", "enc0")}


def measure(which, steps=300):
    dev = torch.device("cuda", 0)
    ops.set_precision("bf16")
    conf = load_yaml(None, trainer_type="vqvae", batch_size=64, batch_len=500)
    torch.manual_seed(1234)
    np.random.seed(1234)
    parallel.seed_shared_python_rng(1234)
    trainer = build_trainer(conf, 14, "/tmp/crank_amd_wsab", device=dev, grad_reduce_fn=parallel.install())
    trainer.steps = 1
    trainer.check_custom_start()
    G = trainer.model["G"]
    side = torch.cuda.Stream(device=dev)
    stacks = {"enc0": G.encoders[0], "enc1": G.encoders[1], "dec0": G.decoders[0], "dec1": G.decoders[1]}
    for name in which:
        rc = _lib.lib().crk_net_set_wgrad_stream(stacks[name].net.handle, side.cuda_stream)
        assert rc == 0, rc
    batch = make_batch(64, 500, 14, seed=1234, device=dev)
    graphed = GraphedStep(trainer, batch, warmup=3)
    for _ in range(30):
        vals = graphed.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graphed.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    loss = float(vals["G"])
    del graphed, trainer
    return ms, loss


def main():
    os.makedirs("gpurun_out", exist_ok=True)
    lines = []
    for rep in range(3):
        for name, which in VARIANTS.items():
            ms, loss = measure(which)
            lines.append(f"wgrad side stream: {name:10s} rep={rep} ms_per_step={ms:.4f} loss_G={loss:.6f}")
            print(lines[-1], flush=True)
    open("gpurun_out/wgrad_stream_ab.txt", "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
