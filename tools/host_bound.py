"""Is the step loop host bound?  Time to ENQUEUE k steps (no synchronisation) against the time until the GPU
has finished them, at the benchmark shape."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crank_amd.bin.train import build_trainer  # noqa: E402
from crank_amd.synthetic import make_batch  # noqa: E402
from crank_amd.utils import load_yaml  # noqa: E402

conf = load_yaml(None, batch_size=64, batch_len=500)
tr = build_trainer(conf, 14, "/tmp/hb_exp")
b = make_batch(64, 500, 14, device="cuda")
for _ in range(20):
    tr.train(b)
torch.cuda.synchronize()
k = 100
t0 = time.perf_counter()
for _ in range(k):
    tr.train(b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / k:.3f} ms/step, until drained {1e3 * (t2 - t0) / k:.3f} ms/step, "
      f"GPU backlog at the end of the loop {1e3 * (t2 - t1):.1f} ms")
