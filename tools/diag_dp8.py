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
for name, nproc, env in (("forced world of one (gloo)", 1, {"CRANK_AMD_FORCE_DIST": "1"}), ("2 ranks", 2, None), ("4 ranks", 4, None), ("8 ranks", 8, None)):
    r = run(name.replace(" ", "_") + ".npz", nproc, env)
    rep = {}
    for k in one.files:
        if k.startswith(("grad/", "loss/")):
            rep[k] = float(np.abs(r[k] - one[k]).max() / (np.abs(one[k]).max() + 1e-20))
    print(name, {k: f"{v:.1e}" for k, v in rep.items() if v > 1e-6})
