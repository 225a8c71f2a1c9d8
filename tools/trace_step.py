"""List every GPU launch of ONE training step in issue order, with the crank_amd source line that issued it:
python tools/trace_step.py [trainer] (torch.profiler with stacks; for finding glue launches)."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
from crank_amd import ops  # noqa: E402
from crank_amd.bin.train import build_trainer  # noqa: E402
from crank_amd.utils import load_yaml  # noqa: E402
from tests.helpers import make_batch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "vqvae"
ops.set_precision("bf16")
conf = load_yaml(None, batch_size=64, batch_len=500, trainer_type=name)
trainer = build_trainer(conf, 14, "/tmp/trace_step")
batch = make_batch(64, 500, 14, device="cuda")
for _ in range(4):
    trainer.train(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.train(batch)
    torch.cuda.synchronize()
evs = prof.events()
cpu = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and e.kernels], key=lambda e: e.time_range.start)
print(sum(len(e.kernels) for e in cpu), "launches attributed to torch ops (the library's own launches are not torch ops)")
seen = set()
for e in cpu:
    # innermost op only: skip an op whose kernels all belong to a child listed too
    key = tuple((k.name, k.duration) for k in e.kernels)
    st = [s for s in (e.stack or []) if "crank_amd/" in s or "tests/" in s or "bench.py" in s]
    where = " | ".join(s.split("repo/")[-1].split(", in ")[0].replace("crank_amd/", "") + ":" + s.split(", in ")[-1] if ", in " in s else s.split("repo/")[-1] for s in st[:4])
    ks = ", ".join(f"{k.name[:38]} {k.duration:.1f}us" for k in e.kernels)
    print(f"{e.name[:34]:34s} [{ks[:90]}]  <- {where[:170]}")
