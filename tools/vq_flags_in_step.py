"""How often the split-f16 VQ search of a real training step needs its exact re-scoring paths (crk_debug_vq_flags):
python tools/vq_flags_in_step.py [steps]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crank_amd import _lib, ops  # noqa: E402
from crank_amd.bin.train import build_trainer  # noqa: E402
from crank_amd.synthetic import make_batch  # noqa: E402
from crank_amd.utils import load_yaml  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ops.set_precision("bf16")
    conf = load_yaml(None, batch_size=64, batch_len=500, trainer_type="vqvae")
    torch.manual_seed(1234)
    trainer = build_trainer(conf, 14, "/tmp/vq_flags")
    trainer.steps = 1
    batch = make_batch(64, 500, 14, seed=1234, device="cuda")
    L = _lib.lib()
    out = (ctypes.c_ulonglong * 3)()
    for s in range(steps):
        L.crk_debug_vq_flags(out, 1)
        trainer.train(batch)
        torch.cuda.synchronize()
        L.crk_debug_vq_flags(out, 0)
        G = trainer.model["G"]
        norms = [q.weight.detach().norm(dim=1) for q in G.quantizers]
        print(f"step {s}: two-candidate {int(out[1])}, full scan {int(out[2])} of {4 * 32000} searched frames; "
              + "; ".join(f"codebook {i}: |w| min {float(n.min()):.3g} median {float(n.median()):.3g} max {float(n.max()):.3g}"
                          for i, n in enumerate(norms)))


if __name__ == "__main__":
    main()
