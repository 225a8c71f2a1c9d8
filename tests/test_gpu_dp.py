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


def _run(args, env=None, timeout=900, outputs=()):
    """Runs the worker(s); every process must exit with 0.  Each worker appends what it did to `<out>.rank<r>.log` (its own
    stderr included: under torchrun the launcher's stderr only holds the launcher's summary) and ends the file with DONE
    after it has torn the process group down - on a failure the logs say how far every rank got."""
    r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=timeout, cwd=REPO)
    logs = ""
    for f in outputs:
        lf = f.replace(".npz", "") + ".log"
        if os.path.exists(lf):
            logs += f"\n--- {os.path.basename(lf)} ---\n" + open(lf).read()[-3000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:], logs)
    for f in outputs:
        lf = f.replace(".npz", "") + ".log"
        assert os.path.exists(lf) and open(lf).read().rstrip().endswith("DONE"), logs


@pytest.mark.parametrize("ttype", ["lsgan", "cyclegan"])
def test_two_ranks_on_one_gpu_equal_one_process(tmp_path, ttype):
    B, T = 4, 120
    worker = os.path.join(REPO, "tests", "dp_gpu_worker.py")
    single = str(tmp_path / "single.npz")
    _run([sys.executable, worker, single, ttype, str(B), str(T)], outputs=[single])
    dp = str(tmp_path / "dp.npz")
    env = dict(os.environ, CRANK_AMD_DIST_BACKEND="gloo")
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
          "--master-port", str(_free_port()), worker, dp, ttype, str(B), str(T)], env=env,
         outputs=[f"{dp}.rank{r}.npz" for r in range(2)])
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


def _launch(tmp_path, name, nproc, args, backend, extra_env=None):
    worker = os.path.join(REPO, "tests", "dp_gpu_worker.py")
    out = str(tmp_path / name)
    env = dict(os.environ, **(extra_env or {}))
    if nproc == 0:
        _run([sys.executable, worker, out] + args, env=env, outputs=[out])
        return [np.load(out)]
    env["CRANK_AMD_DIST_BACKEND"] = backend
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
          "--master-port", str(_free_port()), worker, out] + args, env=env, outputs=[f"{out}.rank{r}.npz" for r in range(nproc)])
    return [np.load(f"{out}.rank{r}.npz") for r in range(nproc)]


def _close(dp, one, keys, tol):
    for k in one.files:
        if k.startswith(keys):
            err = np.abs(dp[k] - one[k]).max() / (np.abs(one[k]).max() + 1e-20)
            assert err < tol, (k, err)


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("ttype", ["vqvae", "cyclegan"])
def test_n_ranks_of_one_utterance_equal_one_process(tmp_path, ttype, world):
    """The world sizes the driver's scaling run uses beyond 2: N ranks x 1 utterance (sharing cuda:0 over gloo) against one
    process x N utterances, two optimisation steps.  What only shows with more than two ranks: EmaBucket's packing of the
    int32 counts in pairs with the C3 riders in its tail summed over N contributions, `shard_batch` / `make_batch(seed + rank)`
    giving every rank a different utterance, the rank-shared generator of the cyclic trainers' random choices keeping N
    processes on the same sequence of collectives, the gradient mean over N.  Run in the split-operand arithmetic: a rank's
    loss is scaled by count_local / count_global (C3) before its backward, and in plain bf16 the scaled output gradient
    rounds differently from the unscaled one unless the ratio is a power of two (1e-4 of the summed gradient at N = 8 -
    rounding noise, not an exchange error; the 2-rank tests above happen to run at ratios that commute)."""
    args = [ttype, str(world), "96", "bf16x3", "eager", "0", "2"]
    one = _launch(tmp_path, "single.npz", 0, args, None)[0]
    ranks = _launch(tmp_path, "dp.npz", world, args, "gloo")
    assert [int(r["rank"]) for r in ranks] == list(range(world)) and all(int(r["world"]) == world for r in ranks)
    r0 = ranks[0]
    for r in ranks[1:]:  # every rank holds the same state, to the bit
        for k in r0.files:
            if k.startswith(("grad/", "flat/", "codebook", "ema_")):
                assert np.array_equal(r0[k], r[k]), (int(r["rank"]), k)
    report = {k: float(np.abs(r0[k] - one[k]).max() / (np.abs(one[k]).max() + 1e-20)) for k in one.files
              if k.startswith(("grad/", "loss/", "flat/", "ema_size", "codebook"))}
    print(ttype, world, "ranks vs one process, relative max error:", {k: f"{v:.1e}" for k, v in report.items() if v > 0})
    for k, v in report.items():  # where a failing block differs (offsets into the model's flat block)
        if k.startswith("grad/") and v >= 1e-5:
            d = np.abs(r0[k] - one[k]) / (np.abs(one[k]).max() + 1e-20)
            worst = np.argsort(d)[::-1][:12]
            print(k, "size", d.size, "elements above 1e-5:", int((d > 1e-5).sum()), "worst offsets", worst.tolist(),
                  "dp", r0[k][worst[:4]].tolist(), "one", one[k][worst[:4]].tolist())
    for k in one.files:
        if k.startswith("loss/"):
            assert np.isclose(float(r0[k]), float(one[k]), rtol=1e-5, atol=1e-7), (k, float(r0[k]), float(one[k]))
    # The speaker-adversarial net's update is the one part of a step that sits BEHIND an optimizer step and a quantizer: it
    # encodes with G's new parameters, and its input's lower level is enc + dec(q_top) (crank/net/module/vqvae2.py:171-190).
    # The ranks' gradient sum of G differs from the single process's in the last bits (1e-6 above), Adam's first step turns
    # that into last-bit differences of G's parameters, and a frame whose two nearest top-level codes are 1e-7 apart then
    # picks the other one: a handful of frames of SPKRADV's input change by O(1), its loss (a mean over all frames) by 1e-7,
    # its gradient by up to 1e-2 of its largest element (tools/diag_dp8.py: the same gradient sits 3e-2 from the fp32 CPU
    # oracle's for the same reason, for one process and for 8 ranks alike; its conv chain itself is shard-independent to 1e-7,
    # tools/diag_plain_shapes.py).  Everything upstream of that decision is held to 1e-5.
    for k in one.files:
        if k.startswith("grad/"):
            assert report[k] < (5e-2 if k == "grad/SPKRADV" else 1e-5), (k, report[k])
    _close(r0, one, ("flat/", "ema_size", "codebook"), 1e-3)


def test_two_ranks_clip_the_global_gradient(tmp_path):
    """clip_grad_norm != 0 under data parallelism: reduce -> clip -> Adam (the reference clips the gradient of its one
    batch, crank/net/trainer/trainer_vqvae.py:203-206), so 2 ranks x B/2 step like one process x B.  Clipping the local
    gradients first (the round-2 order) scales every rank by a different factor and fails this test."""
    args = ["vqvae", "4", "120", "bf16", "eager", "0.5", "3"]
    one = _launch(tmp_path, "single.npz", 0, args, None)[0]
    r0, r1 = _launch(tmp_path, "dp.npz", 2, args, "gloo")
    for k in r0.files:
        if k.startswith(("flat/", "codebook", "ema_")):
            assert np.array_equal(r0[k], r1[k]), k
    # (grad/ holds the gradient block after the step: reduced AND clipped)
    _close(r0, one, ("grad/",), 1e-5)
    _close(r0, one, ("flat/", "ema_size"), 1e-3)
    # the clip was active: the clipped global norm is the threshold
    g = np.concatenate([one[k].ravel() for k in one.files if k == "grad/G"])
    assert abs(np.linalg.norm(g.astype(np.float64)) - 0.5) < 1e-3, np.linalg.norm(g)


@pytest.mark.parametrize("ttype", ["lsgan", "cyclegan"])
def test_two_ranks_replaying_graph_segments_equal_one_process(tmp_path, ttype):
    """The data-parallel step as a chain of HIP graphs with the host-issued collectives between them (GraphedStep):
    2 ranks on cuda:0 over gloo, 7 steps (3 eager, a capture, replays), against one process stepping eagerly."""
    steps = "7" if ttype == "lsgan" else "12"
    one = _launch(tmp_path, "single.npz", 0, [ttype, "4", "120", "bf16", "eager", "0", steps], None)[0]
    r0, r1 = _launch(tmp_path, "dp.npz", 2, [ttype, "4", "120", "bf16", "graph", "0", steps], "gloo")
    assert int(r0["n_graphs"]) >= 1 and int(r0["n_segments"]) >= 5, (int(r0["n_graphs"]), int(r0["n_segments"]))
    for k in r0.files:
        if k.startswith(("flat/", "codebook", "ema_")):
            assert np.array_equal(r0[k], r1[k]), k
    _close(r0, one, ("grad/",), 1e-5)
    _close(r0, one, ("flat/", "ema_size"), 2e-3)
    # (up to 12 optimisation steps with the two summation orders of one / two ranks: the parameters stay within 2e-3, the
    # commitment losses - means over a few hundred code choices - within 2 %)
    for k in one.files:
        if k.startswith("last_loss/"):
            assert np.isclose(float(r0[k]), float(one[k]), rtol=2e-2, atol=1e-6), (k, float(r0[k]), float(one[k]))


def test_a_rank_that_cannot_capture_takes_every_rank_to_the_eager_path(tmp_path):
    """One rank's capture fails (test hook): a capture exchanges nothing, so the ranks are still paired up; the agreement
    all-reduce right after the attempt (parallel.agree_on_capture) sends BOTH ranks down the eager path - no graphs on
    either, the same state on both, and the state of one process stepping eagerly."""
    one = _launch(tmp_path, "single.npz", 0, ["lsgan", "4", "120", "bf16", "eager", "0", "7"], None)[0]
    r0, r1 = _launch(tmp_path, "dp.npz", 2, ["lsgan", "4", "120", "bf16", "graph", "0", "7"], "gloo",
                     {"CRANK_AMD_TEST_REFUSE_CAPTURE_RANK": "1"})
    assert int(r0["n_graphs"]) == 0 and int(r1["n_graphs"]) == 0
    for k in r0.files:
        if k.startswith(("flat/", "codebook", "ema_")):
            assert np.array_equal(r0[k], r1[k]), k
    _close(r0, one, ("grad/",), 1e-5)
    _close(r0, one, ("flat/", "ema_size"), 2e-3)


@pytest.mark.parametrize("in_graph", ["1", "0"])
def test_captured_step_with_rccl_collectives_in_a_world_of_one(tmp_path, in_graph):
    """The captured step with the collectives served by RCCL (backend "nccl"), the backend of a multi-GPU node, on the one GPU
    of the test box: a process group of one rank with the data-parallel code path forced on (CRANK_AMD_FORCE_DIST) - every
    all-reduce is issued on device tensors.  in_graph "1" (the default under nccl): the collectives are captured with the
    step, which is ONE graph; "0": the chain of graphs with the host-issued collectives between the replays."""
    one = _launch(tmp_path, "single.npz", 0, ["lsgan", "4", "120", "bf16", "eager", "0", "7"], None)[0]
    (r0,) = _launch(tmp_path, "dp.npz", 1, ["lsgan", "4", "120", "bf16", "graph", "0", "7"], "nccl",
                    {"CRANK_AMD_FORCE_DIST": "1", "CRANK_AMD_DP_GRAPH_COLLECTIVES": in_graph})
    assert int(r0["n_graphs"]) >= 1
    assert int(r0["n_segments"]) == 1 if in_graph == "1" else int(r0["n_segments"]) >= 5
    _close(r0, one, ("grad/",), 1e-5)
    _close(r0, one, ("flat/", "ema_size"), 2e-3)
