"""Diagnostic: the speaker-adversarial net's gradient of a cyclegan step, global batch 8 x 96 frames, as one process, as a
forced data-parallel world of one, and as 2 / 4 / 8 ranks sharing the GPU over gloo (tests/dp_gpu_worker.py)."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
worker = os.path.join(REPO, "tests", "dp_gpu_worker.py")
ttype = sys.argv[1] if len(sys.argv) > 1 else "cyclegan"
args = [ttype, "8", "96", "bf16x3", "eager", "0", "1"]
tmp = tempfile.mkdtemp()


def port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(name, nproc, env_over=None):
    out = os.path.join(tmp, name)
    env = dict(os.environ, **(env_over or {}))
    if nproc == 0:
        subprocess.run([sys.executable, worker, out] + args, env=env, check=True, cwd=REPO)
        return np.load(out)
    env.setdefault("CRANK_AMD_DIST_BACKEND", "gloo")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                    "--master-port", str(port()), worker, out] + args, env=env, check=True, cwd=REPO)
    return np.load(f"{out}.rank0.npz")


one = run("single.npz", 0)


def oracle_spkradv_grads():
    """The same scenario on the CPU oracle (fp32): SPKRADV's parameter gradients of step 0, by state-dict key."""
    sys.path.insert(0, REPO)
    import random

    import torch

    from crank_amd.net.trainer import TrainerWrapper
    from crank_amd.utils import load_yaml
    from oracle import modules as om
    from tests.helpers import fill_models, make_batch

    B, T, S = 8, 96, 3
    over = dict(trainer_type=ttype, batch_size=B, batch_len=T)
    if ttype != "vqvae":
        over.update(discriminator_dropout=0.0, n_steps_gan_start=0)
    if ttype in ("cyclegan", "stargan"):
        over.update(use_cyclic_training=True, n_steps_cycle_start=0)
    conf = load_yaml(None, **over)
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    models = om.get_model(conf, S, None)
    fill_models(models)
    for m in models.values():
        m.train()
    opt = om.get_optimizer(conf, models)
    grads = {}
    real = opt["SPKRADV"].step

    def step(*a, **k):
        grads.update({key: p.grad.detach().numpy().copy() for key, p in models["SPKRADV"].named_parameters() if p.grad is not None})
        return real(*a, **k)

    opt["SPKRADV"].step = step
    tr = TrainerWrapper(ttype, model=models, optimizer=opt, criterion=om.get_criterion(conf), dataloader={"spkrs": {f"spk{i}": i for i in range(S)}},
                        writer=None, expdir="/tmp/diag_dp8", conf=conf, feat_conf=conf["feature"], scheduler=None, scaler=None, resume=0,
                        device="cpu", n_jobs=1)
    tr.steps = 1
    tr.check_custom_start()
    random.seed(99); random.random()
    tr.train(make_batch(B, T, S, seed=11))
    return grads


og = oracle_spkradv_grads()


def vs_oracle(r, name):
    w = 0.0
    for key, g in og.items():
        k = f"gradkey/SPKRADV/{key}"
        if k in r.files:
            w = max(w, float(np.abs(r[k] - g).max() / (np.abs(g).max() + 1e-20)))
    print(name, "SPKRADV gradient against the fp32 CPU oracle: relative max error", f"{w:.1e}")


vs_oracle(one, "one process")
for name, nproc, env in (("forced world of one (gloo)", 1, {"CRANK_AMD_FORCE_DIST": "1"}), ("2 ranks", 2, None), ("4 ranks", 4, None), ("8 ranks", 8, None)):
    r = run(name.replace(" ", "_") + ".npz", nproc, env)
    rep = {}
    for k in one.files:
        if k.startswith(("grad/", "loss/")):
            rep[k] = float(np.abs(r[k] - one[k]).max() / (np.abs(one[k]).max() + 1e-20))
    print(name, {k: f"{v:.1e}" for k, v in rep.items() if v > 1e-6})
    vs_oracle(r, name)
