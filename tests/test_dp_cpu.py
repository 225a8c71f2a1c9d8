"""CPU, world_size 2 over gloo: the data-parallel exchange of crank_amd/parallel.py makes
2 ranks x 2 utterances equal one process x 4 utterances: summed
gradients (C1), masked-mean re-weighting (C3), integer EMA statistics (C2)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import fill_models, make_batch
from crank_amd.utils import load_yaml

B, T, S = 4, 96, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _ReduceThenStep:
    """torch optimizer + C1 all-reduce; records the reduced gradient of the first step."""

    def __init__(self, opt, params, store, name):
        self.opt, self.params, self.store, self.name = opt, list(params), store, name

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self):
        from crank_amd import parallel

        for p in self.params:
            if p.grad is not None and parallel.is_dist():
                parallel.grad_allreduce(p.grad)
        if self.name not in self.store:
            self.store[self.name] = torch.cat([p.grad.reshape(-1) for p in self.params if p.grad is not None]).clone()
        self.opt.step()


def _run_trainer(conf, batch, store):
    from crank_amd.net.trainer import TrainerWrapper
    from oracle import modules as om

    torch.manual_seed(0)
    models = om.get_model(conf, S)
    fill_models(models)
    for m in models.values():
        m.train()
    opts = om.get_optimizer(conf, models)
    wrapped = {k: _ReduceThenStep(o, models[k].parameters(), store, k) for k, o in opts.items()}
    tr = TrainerWrapper(conf["trainer_type"], model=models, optimizer=wrapped, criterion=om.get_criterion(conf),
                        dataloader={"spkrs": {f"s{i}": i for i in range(S)}}, writer=None, expdir="/tmp/dp", conf=conf,
                        feat_conf=conf["feature"], scheduler=None, scaler=None, resume=0, device="cpu", n_jobs=1)
    return tr.train(batch)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from crank_amd import parallel

    parallel.init_from_env(backend="gloo")
    assert parallel.is_dist()
    conf = load_yaml(None, batch_size=B // world, batch_len=T)
    full = make_batch(B, T, S, seed=11)
    shard = parallel.shard_batch(full, rank, world)
    store = {}
    vals = _run_trainer(conf, shard, store)
    # C2: integer statistics sum exactly
    counts = torch.arange(8, dtype=torch.int32) * (rank + 1)
    sums = torch.arange(8, dtype=torch.int64) * (10 ** 12) * (rank + 1)
    parallel.ema_allreduce(counts, sums)
    # C2 as the generator sends it: every quantizer's statistics in ONE int64 message, int32 counts packed in pairs
    bucket = parallel.EmaBucket([(3, 4), (2, 6), (2, 5)], "cpu")  # (an odd codebook size pads its counts to a whole word)
    for i in range(3):
        c, sm = bucket.views(i)
        c.copy_(torch.arange(c.numel(), dtype=torch.int32) * (rank + 1) + 2 ** 30 * (i == 1))
        sm.copy_(-(torch.arange(sm.numel(), dtype=torch.int64) + 1) * (2 ** 40) * (rank + 1))
    bucket.reduce()
    packed = [t.clone().numpy() for i in range(3) for t in bucket.views(i)]
    if rank == 0:
        q.put(({k: v.numpy() for k, v in store.items()}, vals, counts.numpy(), sums.numpy(), packed))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    grads_dp, vals_dp, counts, sums, packed = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(counts, np.arange(8) * 3)
    np.testing.assert_array_equal(sums, np.arange(8, dtype=np.int64) * (10 ** 12) * 3)
    np.testing.assert_array_equal(packed[0], np.arange(4) * 3)
    np.testing.assert_array_equal(packed[1], -(np.arange(12, dtype=np.int64) + 1) * (2 ** 40) * 3)
    np.testing.assert_array_equal(packed[2].astype(np.uint32), (np.arange(6) * 3 + 2 ** 31).astype(np.uint32))
    np.testing.assert_array_equal(packed[3], -(np.arange(12, dtype=np.int64) + 1) * (2 ** 40) * 3)
    np.testing.assert_array_equal(packed[4], np.arange(5) * 3)
    np.testing.assert_array_equal(packed[5], -(np.arange(10, dtype=np.int64) + 1) * (2 ** 40) * 3)

    torch.set_num_threads(4)
    conf = load_yaml(None, batch_size=B, batch_len=T)
    store = {}
    vals_1 = _run_trainer(conf, make_batch(B, T, S, seed=11), store)
    for k in ["G", "SPKRADV", "C", "G_l1", "G_stft", "G_commit0", "G_spkradv_org", "C_real"]:
        np.testing.assert_allclose(vals_dp[k], vals_1[k], rtol=2e-5, err_msg=k)
    for k, g in store.items():
        ref = g.numpy()
        err = np.abs(grads_dp[k] - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < 1e-4, (k, err)


def test_shard_and_rescale_single_process():
    from crank_amd import parallel

    full = make_batch(4, 20, 3, seed=1)
    s0, s1 = parallel.shard_batch(full, 0, 2), parallel.shard_batch(full, 1, 2)
    assert torch.equal(torch.cat([s0["in_feats"], s1["in_feats"]]), full["in_feats"])
    assert s0["flbl"] + s1["flbl"] == full["flbl"]
    assert not parallel.is_dist()
    assert float(parallel.mean_rescale(torch.tensor(5.0))) == 1.0
    assert parallel.install() is None
