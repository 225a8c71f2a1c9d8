"""The product's data-parallel path on the one GPU the test box has: two ranks share cuda:0 and exchange over
gloo (the collectives are the same calls RCCL serves on a multi-GPU node).  2 ranks x B/2 utterances must equal
one process x B utterances: C1 one gradient all-reduce per model inside FlatAdam.step,
C2 one EMA-statistics message per generator forward, C3 one count message per step - all of it the product
code (crank_amd/parallel.py, net/trainer/utils.py, net/module/vqvae2.py), not the oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(args, env=None, timeout=900):
    r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=timeout, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("ttype", ["lsgan", "cyclegan"])
def test_two_ranks_on_one_gpu_equal_one_process(tmp_path, ttype):
    B, T = 4, 120
    worker = os.path.join(REPO, "tests", "dp_gpu_worker.py")
    single = str(tmp_path / "single.npz")
    _run([sys.executable, worker, single, ttype, str(B), str(T)])
    dp = str(tmp_path / "dp.npz")
    env = dict(os.environ, CRANK_AMD_DIST_BACKEND="gloo")
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
          "--master-port", str(_free_port()), worker, dp, ttype, str(B), str(T)], env=env)
    one = np.load(single)
    r0, r1 = np.load(dp + ".rank0.npz"), np.load(dp + ".rank1.npz")
    assert int(r0["world"]) == 2 and int(r1["rank"]) == 1
    # every rank holds the same state, to the bit: gradients are summed by the same all-reduce, the integer EMA
    # statistics are order independent, Adam is deterministic
    for k in r0.files:
        if k.startswith(("grad/", "flat/", "codebook", "ema_")):
            assert np.array_equal(r0[k], r1[k]), k
    # ... and it is the single-process state: reduced gradients of step 1 to 1e-5 of their scale, global loss
    # values to 1e-5 relative
    for k in one.files:
        if k.startswith("grad/"):
            err = np.abs(r0[k] - one[k]).max() / (np.abs(one[k]).max() + 1e-20)
            print(ttype, k, "gradient error vs one process", err)
            assert err < 1e-5, (k, err)
        if k.startswith("loss/"):
            assert np.isclose(float(r0[k]), float(one[k]), rtol=1e-5, atol=1e-7), (k, float(r0[k]), float(one[k]))
    # after 3 steps the two runs have gone through different summation orders three times: parameters agree to
    # fp32 accumulation noise, the EMA cluster sizes (integer statistics of identical code choices) closely
    for k in one.files:
        if k.startswith("flat/") or k.startswith("ema_size"):
            err = np.abs(r0[k] - one[k]).max() / (np.abs(one[k]).max() + 1e-20)
            print(ttype, k, "after 3 steps", err)
            assert err < 1e-3, (k, err)
