"""`bench.py` on its N > 1 path on the one GPU of the test box (collected last, tests/conftest.py): two ranks sharing cuda:0
over gloo - the launch contract of the driver (`python -m torch.distributed.run ... bench.py --gpus 2`), the chain of graph
segments, the max-over-ranks timing, ONE JSON line from rank 0 - and the same path in a world of one rank over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.helpers import REPO

pytestmark = pytest.mark.gpu
_SMALL = ["--steps", "3", "--warmup", "1", "--batch", "4", "--no-roofline", "--no-extras", "--no-cpu-baseline"]


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(r):
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu_over_gloo():
    env = dict(os.environ, CRANK_AMD_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_port()), os.path.join(REPO, "bench.py"), "--gpus", "2"] + _SMALL,
                       env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    d = _line(r)
    assert d["world_size_seen"] == 2 and d["n_gpus"] == 2 and d["dist_backend"] == "gloo"
    assert d["config"]["global_batch"] == 8 and d["scaling"] == "weak"
    assert d["launch"].startswith("chain of"), d["launch"]  # (gloo's collectives cannot be captured)
    assert abs(d["value"] - 2 * 4 * 500 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["loss_G"] is not None and d["loss_G"] == d["loss_G"]


def test_bench_eight_ranks_through_its_own_respawn_path():
    """`python bench.py --gpus 8` from a plain shell: bench.py starts its own eight ranks under torch.distributed.run on
    127.0.0.1 (respawn_under_torchrun) - here all eight share cuda:0 and exchange over gloo - and relays rank 0's ONE line:
    the whole-job value is eight ranks' frames over the max-over-ranks time."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CRANK_AMD_DIST_BACKEND"] = "gloo"
    small = ["--steps", "2", "--warmup", "1", "--batch", "1", "--no-roofline", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8"] + small, env=env, capture_output=True, text=True,
                       timeout=1200, cwd=REPO)
    d = _line(r)
    assert d["world_size_seen"] == 8 and d["n_gpus"] == 8 and d["dist_backend"] == "gloo"
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp8" and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 1 * 500 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["loss_G"] is not None and d["loss_G"] == d["loss_G"]


def test_bench_forced_data_parallel_path_in_a_world_of_one_over_rccl():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-dist"] + _SMALL,
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    d = _line(r)
    assert d["world_size_seen"] == 1 and d["dist_backend"] == "nccl" and d["forced_dist_world_of_one"] is True
    # (under RCCL the collectives are captured with the step: one graph; CRANK_AMD_DP_GRAPH_COLLECTIVES=0: the chain)
    assert d["launch"].startswith(("one hip graph", "chain of")), d["launch"]
