"""Reproduction of the round-4 abort "Fatal Python error: Aborted ... Garbage-collecting" inside a capture:
python tools/gc_capture_repro.py [nofix]   - with `nofix` GraphedStep's collector hold is patched out and the process is
expected to die with SIGABRT when the collector finalizes an earlier trainer's HIP graphs while the next step is captured."""
import gc
import sys

import torch

sys.path.insert(0, ".")
from crank_amd import ops  # noqa: E402
from crank_amd.bin.train import build_trainer  # noqa: E402
from crank_amd.net.trainer import basetrainer  # noqa: E402
from crank_amd.utils import load_yaml  # noqa: E402
from tests.helpers import fill_models, make_batch  # noqa: E402

NOFIX = len(sys.argv) > 1 and sys.argv[1] == "nofix"
GARBAGE = sys.argv[2] if len(sys.argv) > 2 else "vqvae"
if NOFIX:  # nothing is collected before the capture: the old trainer's graphs are finalized inside it
    basetrainer.hold_collector_for_capture = lambda: False
from crank_amd.net.trainer.trainer_vqvae import VQVAETrainer  # noqa: E402

plain = VQVAETrainer._get_loss_dict


def collecting(self, batch=None):  # a collection at the start of the captured step, whatever the collector's state
    if torch.cuda.is_current_stream_capturing():
        print("collected inside the capture:", gc.collect(), flush=True)
    return plain(self, batch)


VQVAETrainer._get_loss_dict = collecting
ops.set_precision("bf16")
conf = load_yaml(None, batch_size=4, batch_len=160, hip_graph=True)


def run(n, ttype="vqvae"):
    torch.manual_seed(1)
    c = conf if ttype == "vqvae" else load_yaml(None, batch_size=4, batch_len=160, hip_graph=True, trainer_type=ttype,
                                                n_steps_gan_start=0, n_steps_cycle_start=0, use_cyclic_training=ttype != "lsgan")
    tr = build_trainer(c, 5, "/tmp/crank_amd_gcgraph")
    tr.steps = 1
    tr.check_custom_start()
    fill_models(tr.model)
    for s in range(n):
        tr.train_graphed(make_batch(4, 160, 5, seed=70 + s, device="cuda"))
    torch.cuda.synchronize()
    assert any(slot[1] is not None for slot in tr._graphs.values())


gc.disable()
run(5, GARBAGE)  # garbage: a trainer <-> GraphedStep cycle owning HIP graphs and their pool
if not NOFIX:
    gc.enable()
    gc.set_threshold(1, 1, 1)
run(6, sys.argv[3] if len(sys.argv) > 3 else "vqvae")
print("CAPTURED-WITH-GARBAGE-OK")
