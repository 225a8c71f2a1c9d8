"""Where the host spends its time enqueueing a step (cProfile over 200 eager steps at the benchmark shape)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crank_amd.bin.train import build_trainer  # noqa: E402
from crank_amd.synthetic import make_batch  # noqa: E402
from crank_amd.utils import load_yaml  # noqa: E402

conf = load_yaml(None, batch_size=64, batch_len=500)
tr = build_trainer(conf, 14, "/tmp/hp_exp")
b = make_batch(64, 500, 14, device="cuda")
for _ in range(20):
    tr.train(b)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    tr.train(b)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
st.sort_stats("cumtime").print_stats(28)
