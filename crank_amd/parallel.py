"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests).  The reference is single-process; SURVEY.md
section 8e defines what has to be exchanged so that N ranks x B utterances equal one
batch of N*B:

C1  gradients   - each model's flat fp32 gradient block is summed with ONE all-reduce
                  right before that model's Adam step (messages: G 5.2 MB, D 1.6 MB,
                  C 0.6 MB, SPKRADV 0.16 MB: latency-bound on xGMI, hence one message
                  per model rather than per-layer buckets).
C2  VQ EMA      - per-code counts (int32) and feature sums (int64 fixed point) are
                  summed before the EMA blend; integer sums are order independent, so
                  codebooks stay bit-identical on every rank.
C3  loss means  - every loss is a mean over the rank's own masked elements; scaling each
                  rank's loss by count_local * world / count_global before backward makes
                  the summed gradients equal the single-process gradient.  The helper
                  below returns that factor from one tiny all-reduce.

Gradient sums (not means) are exchanged, so the per-rank losses are divided by world
size through the C3 factor (count_global already spans all ranks).
"""
import os
import random

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if is_dist() else 0


def world_size():
    return dist.get_world_size() if is_dist() else 1


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://")
    return rank, world, local


def grad_allreduce(flat_grad):
    """C1: in-place sum of one model's flat gradient block."""
    if is_dist():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)


def ema_allreduce(counts, sums):
    """C2: in-place sums of the integer EMA statistics, ONE message per quantizer call (the int32
    counts ride behind the int64 sums: these collectives are latency bound, 133 KB each)."""
    if is_dist():
        n = sums.numel()
        buf = torch.empty(n + counts.numel(), device=sums.device, dtype=torch.int64)
        buf[:n] = sums.reshape(-1)
        buf[n:] = counts
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        sums.reshape(-1).copy_(buf[:n])
        counts.copy_(buf[n:])


def mean_rescale(count_local):
    """C3: factor count_local / count_global for a masked-mean loss (tensor in, tensor out)."""
    if not is_dist():
        return torch.ones((), device=count_local.device)
    tot = count_local.detach().clone().float()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return count_local.float() / tot


def seed_shared_python_rng(seed=1234):
    """The cyclegan / stargan steps draw from Python's RNG inside the loss
    (trainer_cyclegan.py:166, trainer_stargan.py:91): all ranks must draw the same."""
    random.seed(seed)


def shard_batch(batch, rank, world):
    """Rank r takes utterances [r*B/world, (r+1)*B/world) of a global batch."""
    out = {}
    for k, v in batch.items():
        n = len(v)
        per = n // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out


class _DPLoss:
    """Wraps one criterion so that its value is this rank's share of the GLOBAL mean
    (C3).  kind: "masked" (x, y, mask=None, causal_size=0), "plain" (unmasked mean ->
    1/world) or "ce" (count = targets != ignore_index)."""

    def __init__(self, fn, kind, cache):
        self.fn, self.kind, self.cache = fn, kind, cache

    def _factor(self, key, count_fn):
        if key not in self.cache:
            self.cache[key] = mean_rescale(count_fn())
        return self.cache[key]

    def __call__(self, *args, **kwargs):
        v = self.fn(*args, **kwargs)
        world = dist.get_world_size()
        if self.kind == "plain":
            return v / world
        if self.kind == "ce":
            tgt = args[1]
            ign = getattr(self.fn, "ignore_index", -100)
            return v * self._factor(("ce", tgt.data_ptr(), tgt.numel()), lambda: (tgt != ign).sum())
        mask = kwargs.get("mask", args[2] if len(args) > 2 else None)
        if mask is None:
            return v / world
        cs = kwargs.get("causal_size", args[3] if len(args) > 3 else 0)
        if getattr(self.fn, "causal", False) and cs != 0:
            mask = mask[:, cs:] if cs > 0 else mask[:, :cs]
        return v * self._factor(("m", mask.data_ptr(), mask.numel(), cs), lambda: mask.sum())


def wrap_criterion(criterion):
    """DP view of the trainers' criterion dict; call ``criterion["_reset"]()`` once per step."""
    cache = {}
    kinds = {"mse": "plain", "l1": "plain", "kld": "plain", "ce": "ce", "fmse": "masked", "fl1": "masked",
             "fstft": "plain"}
    out = {k: _DPLoss(v, kinds[k], cache) for k, v in criterion.items() if k in kinds}
    out["_reset"] = cache.clear
    return out


def install(models=None):
    """Wire C1/C2 into the product: returns the grad-reduce callable for get_optimizer."""
    from .net.module import vqvae2

    vqvae2.set_ema_reduce_fn(ema_allreduce if is_dist() else None)
    return grad_allreduce if is_dist() else None
