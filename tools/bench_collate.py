"""Batch assembly at the benchmark shape (B=64, T=500, 80-dim, 14 speakers) over a synthetic ragged
corpus: host-inclusive time per batch and the collate kernel's own time (HIP events around the
launch loop).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel summary kept in profiles/."""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crank_amd.net.trainer.dataset import BaseDataset  # noqa: E402

rs = np.random.RandomState(7)
S, D, T, B, U = 14, 80, 500, 64, 980  # VCC2020: 14 speakers x 70 utterances
spkrs = [f"spk{i:02d}" for i in range(S)]
lens = rs.randint(250, 1001, size=U)
feats = {f"/c/{spkrs[i % S]}/u{i:04d}.h5": i for i in range(U)}


def reader(h5f, ext="mlfb"):
    n = lens[feats[h5f]]
    r = np.random.RandomState(feats[h5f])
    if ext == "mlfb":
        return (r.standard_normal((n, D)) * 2 + 1).astype(np.float32)
    if ext == "lcf0":
        return (5 + 0.3 * r.standard_normal((n, 1))).astype(np.float32)
    return (r.uniform(size=(n, 1)) < 0.7).astype(np.float32)


sc = {"mlfb": SimpleNamespace(mean_=np.ones(D), scale_=np.full(D, 2.0)), "lcf0": SimpleNamespace(mean_=np.array([5.0]), scale_=np.array([0.3]))}
for i, s in enumerate(spkrs):
    sc[s] = {"lcf0": SimpleNamespace(mean_=np.array([5.0 + 0.01 * i]), var_=np.array([0.09 + 0.001 * i]))}
conf = {"batch_len": T, "input_feat_type": "mlfb", "output_feat_type": "mlfb", "use_raw": False, "ignore_scaler": [], "spec_augment": False}
t0 = time.perf_counter()
dset = BaseDataset(conf, {"train": {"feats": {k: k for k in feats}, "spkrs": spkrs}}, sc, reader=reader)
torch.cuda.synchronize()
print(f"corpus: {U} utterances, {int(lens.sum())} frames, {sum(v.numel() * 4 for v in dset.packed.values()) / 1e6:.0f} MB in HBM, "
      f"packed + normalised in {time.perf_counter() - t0:.2f} s (host-side generation included)")
import random  # noqa: E402

random.seed(0)
order = [rs.randint(0, U, size=B).tolist() for _ in range(220)]
for idx in order[:20]:
    batch = dset.assemble(idx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for idx in order[20:]:
    batch = dset.assemble(idx)
e1.record()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
out_bytes = sum(v.numel() * v.element_size() for v in batch.values() if isinstance(v, torch.Tensor))
in_bytes = int(np.minimum(lens[order[-1]], T).sum()) * (2 * D + 3) * 4
print(f"assemble: {dt * 1e6:.1f} us per batch wall (host draws + launch + 4 mask clones), device span {e0.elapsed_time(e1) * 1e3 / 200:.1f} us per batch; "
      f"{out_bytes / 1e6:.1f} MB written + {in_bytes / 1e6:.1f} MB read per batch; {B * T / dt / 1e6:.1f} M frames/s")
