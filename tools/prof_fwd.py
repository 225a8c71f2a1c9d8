"""Profile driver: a few G forwards at the benchmark shape (used under rocprofv3).
    python tools/prof_fwd.py [fwd|bwd|nograd] [iterations]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crank_amd import ops
from crank_amd.bin.train import get_model
from crank_amd.synthetic import make_batch
from crank_amd.utils import load_yaml

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
ops.set_precision("bf16")
conf = load_yaml(None, batch_size=64, batch_len=500)
m = get_model(conf, 14, "cuda")
b = make_batch(64, 500, 14, device="cuda")
dec_h = torch.cat([b["lcf0"], b["uv"]], -1)
h = b["org_h"].clone(); h[:, :] = h[:, 0:1]
for i in range(3 if len(sys.argv) < 3 else int(sys.argv[2])):
    with torch.set_grad_enabled(mode != "nograd"):
        o = m["G"](b["in_feats"], None, dec_h, spkrvec=h)
    if mode == "bwd":
        o["decoded"].sum().backward()
torch.cuda.synchronize()
print("done")
