"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests and for several ranks sharing one GPU).  The
reference is single-process; SURVEY.md section 8e defines what has to be exchanged so
that N ranks x B utterances equal one batch of N*B:

C1  gradients   - each model's flat fp32 gradient block is summed with ONE all-reduce
                  right before that model's Adam step (messages: G 5.2 MB, D 1.6 MB,
                  C 0.6 MB, SPKRADV 0.16 MB: latency-bound on xGMI, hence one message
                  per model rather than per-layer buckets).
C2  VQ EMA      - per-code counts (int32) and feature sums (int64 fixed point) of ALL
                  quantizers of a generator forward travel as ONE int64 message, issued
                  after the last quantizer of the forward (``EmaBucket``); the EMA blend of
                  every quantizer follows it.  Integer sums are order independent, so
                  codebooks stay bit-identical on every rank.  A codebook is not read
                  again before the next forward, so deferring the blend changes nothing.
C3  loss means  - every loss is a mean over the rank's own masked elements; scaling each
                  rank's loss by count_local / count_global before backward makes the
                  summed gradients equal the single-process gradient.  The element counts
                  of every mask / target a step can use are known when the batch arrives:
                  ``prepare_step`` collects them in a fixed order and they travel in the tail of
                  the step's first C2 message (``EmaBucket.RIDERS``; a step without one - no EMA
                  codebook - sums them with a small all-reduce of their own before the first loss).  A mask that was not announced
                  (none in the four trainers) falls back to its own all-reduce at the
                  point of use - legal because every rank runs the same sequence of
                  losses: the trainers draw their random choices from a generator that
                  is seeded identically on all ranks and shared with nothing else
                  (``BaseTrainer.rng``), never from the global ``random`` stream the
                  dataset draws from.

Gradient sums (not means) are exchanged, so the per-rank losses are divided by world
size through the C3 factor (count_global already spans all ranks).
"""
import random

import torch
import torch.distributed as dist

from . import config
from .config import cfg


def is_dist():
    """True when the step has to exchange with other ranks.  CRANK_AMD_FORCE_DIST=1 keeps the data-parallel code path
    on in a world of one (tests: the collectives of the captured step against RCCL on the single GPU of the test box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or cfg.force_dist


def rank():
    return dist.get_rank() if is_dist() else 0


def world_size():
    return dist.get_world_size() if is_dist() else 1


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).
    CRANK_AMD_DIST_BACKEND overrides the backend (gloo: several ranks on one GPU)."""
    rank, local, world = config.torchrun()
    forced = cfg.force_dist and rank is not None
    if world <= 1 and not forced:
        return 0, 1, 0
    if backend is None:
        backend = cfg.dist_backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://")
    return rank, world, local


# A step that is being captured as HIP graphs (BaseTrainer GraphedStep) registers itself here: every collective then
# closes the running capture, is issued from the host, and opens the next one - the step becomes a chain of graphs with
# the host-issued collectives between them (any backend; nothing but the collectives is enqueued per step).
_segmenter = None


def graph_collectives():
    """Backend nccl (RCCL): the collectives of a captured step are captured WITH it - RCCL's kernels become nodes of the step's
    ONE HIP graph (ProcessGroupNCCL supports stream capture: a captured collective is not handed to its watchdog; the same
    thing torch's graph-captured DDP and the serving stacks on MI300 do) - instead of host calls between a chain of graphs.
    In a world of one that takes the data-parallel step from + 18 % to + 7 % over the single-process step:
    what the chain costs is the stream hand-over at every boundary, and a captured collective has none.  gloo (CPU
    collectives; the multi-rank tests on one GPU) cannot be captured and keeps the chain.  CRANK_AMD_DP_GRAPH_COLLECTIVES=0:
    the chain for RCCL too (rounds 3 - 4; the fallback should a capture with collectives misbehave on some node)."""
    return (cfg.dp_graph_collectives and dist.is_initialized()
            and dist.get_backend() == "nccl")


def all_reduce_sum(t):
    """In-place sum over the ranks.  RCCL reduces device tensors directly; gloo (several ranks sharing one GPU
    in the tests, or CPU tensors) goes through a host copy for device tensors - not every gloo build takes
    them."""
    if _segmenter is not None and not (t.is_cuda and graph_collectives()):
        _segmenter.collective(t)
        return
    all_reduce_now(t)


def all_reduce_many(ts):
    """The in-place sums of several tensors at ONE point of the step (inside a captured step: one segment boundary for all
    of them): the gradient blocks of two small models travel together."""
    ts = [t for t in ts if t is not None]
    if not ts:
        return
    if _segmenter is not None and not (all(t.is_cuda for t in ts) and graph_collectives()):
        _segmenter.collective(ts)
        return
    for t in ts:
        all_reduce_now(t)


def all_reduce_now(t):
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def drain_backend_watchdog(seconds=0.35):
    """Called before a stream capture under the nccl (RCCL) backend.  ProcessGroupNCCL's watchdog thread polls the end events
    of collectives that were ISSUED BEFORE the capture (hipEventQuery, every 100 ms) until it has seen them complete; a
    query that lands inside another thread's global-mode capture fails with hipErrorStreamCaptureUnsupported, the
    watchdog rethrows and the process dies with SIGABRT (round 3's sporadic abort; profiles/round4_rccl_abort_root_cause.txt).
    GraphedStep captures thread-locally, which by the API's contract leaves other threads alone; this makes the hazard
    impossible rather than legal: complete everything, then give the watchdog three of its periods to retire its list -
    a capture issues no live collective (GraphedStep.collective), so the list stays empty until the first replay."""
    if is_dist() and dist.get_backend() == "nccl":
        import time

        torch.cuda.synchronize()
        time.sleep(seconds)


def agree_on_capture(ok):
    """True when EVERY rank captured its step (one small MIN all-reduce, issued by every rank right after its capture
    attempt, successful or not): the ranks replay together or step eagerly together."""
    if not is_dist():
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


class _Pending:
    """A collective that has been started (``all_reduce_start``) and must be completed with ``finish()`` before its tensor is read."""

    def __init__(self, tensor, work=None, deferred=False, segment_key=None):
        self.tensor, self.work, self.deferred, self.segment_key = tensor, work, deferred, segment_key

    def finish(self):
        if self.segment_key is not None:  # inside a captured step: the boundary at which a replay waits for the collective
            if _segmenter is not None:
                _segmenter.collective_finish(self.segment_key)
        elif self.deferred:  # backends / modes without an asynchronous form: the blocking collective, now
            all_reduce_sum(self.tensor)
        elif self.work is not None:
            self.work.wait()  # (RCCL: the current stream waits for the collective's stream; the host does not)
        self.work, self.deferred, self.segment_key = None, False, None


def all_reduce_async(t):
    """Issue the in-place sum of `t` now and return something with ``wait()`` (RCCL: the collective runs on its own stream,
    ``wait()`` makes the current stream wait for it), or None when the backend has no asynchronous form for `t` (gloo on a
    device tensor goes through the host: done when this returns)."""
    if t.is_cuda and dist.get_backend() == "nccl":
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
    all_reduce_now(t)
    return None


def all_reduce_start(t):
    """Begin the in-place sum of `t` over the ranks and return a handle; kernels enqueued before ``handle.finish()`` run
    beside the collective (RCCL works on a stream of its own).  Inside a captured step the start and the finish are two
    segment boundaries: a replay issues the collective at the first, replays the segment between them beside it and waits at
    the second.  With gloo or on CPU tensors the collective is issued by ``finish()`` - the same result, without the overlap."""
    if _segmenter is not None and not (t.is_cuda and graph_collectives()):
        return _Pending(t, segment_key=_segmenter.collective_start(t))
    if t.is_cuda and dist.get_backend() == "nccl":
        return _Pending(t, work=dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
    return _Pending(t, deferred=True)


def grad_allreduce(flat_grad):
    """C1: in-place sum of one model's flat gradient block."""
    if is_dist():
        all_reduce_sum(flat_grad)


def grad_allreduce_start(flat_grad):
    return all_reduce_start(flat_grad) if is_dist() else None


# ---------------------------------------------------------------------------- C2
class EmaBucket:
    """The integer EMA statistics of every quantizer of one generator as ONE int64 message.

    Layout: [sums_0 (D*K int64) | counts_0 (K int32 = K/2 int64 words) | sums_1 | counts_1 | ...].
    The int32 counts are summed as packed pairs: lo + hi * 2^32 per word.  Counts are non-negative and the
    global frame count is far below 2^32, so the low halves never carry into the high halves and the int64
    sum of the words IS the pair of int32 sums (little endian)."""

    RIDERS = 32  # int64 words behind the statistics: the step's mask / target element counts (C3) travel with the first message

    def __init__(self, dims, device):
        self.slots, off = [], 0
        for D, K in dims:
            self.slots.append((off, D * K, K))
            off += D * K + (K + 1) // 2  # (an odd codebook size leaves the last half word unused: it stays zero)
        self.tail = off
        self.buf = torch.zeros(off + self.RIDERS, device=device, dtype=torch.int64)

    def views(self, i):
        """(counts int32 (K), sums int64 (D*K)) of quantizer i: the kernels write straight into the message."""
        off, nsum, K = self.slots[i]
        sums = self.buf[off: off + nsum]
        counts = self.buf[off + nsum: off + nsum + (K + 1) // 2].view(torch.int32)[:K]
        return counts, sums

    def reduce(self):
        if is_dist():
            n = _step.board(self.buf[self.tail:])  # C3's counts of this step, if they still wait for a ride
            all_reduce_sum(self.buf)
            if n:
                _step.resolve(self.buf[self.tail: self.tail + n])


def ema_allreduce(counts, sums):
    """C2 for a single quantizer (kept for callers outside a generator forward): one message, the int32
    counts ride behind the int64 sums."""
    if is_dist():
        n = sums.numel()
        buf = torch.empty(n + counts.numel(), device=sums.device, dtype=torch.int64)
        buf[:n] = sums.reshape(-1)
        buf[n:] = counts
        all_reduce_sum(buf)
        sums.reshape(-1).copy_(buf[:n])
        counts.copy_(buf[n:])


# ---------------------------------------------------------------------------- C3
def mean_rescale(count_local):
    """Factor count_local / count_global for a masked-mean loss (tensor in, tensor out), own all-reduce."""
    if not is_dist():
        return torch.ones((), device=count_local.device)
    tot = count_local.detach().clone().float()
    all_reduce_sum(tot)
    return count_local.float() / tot


def _key(t):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()))


def causal_mask_view(mask, cs):
    """The frames of `mask` a causal-shifted feature loss keeps (crank/net/module/loss.py:33-43)."""
    if cs > 0:
        return mask[:, cs:]
    if cs < 0:
        return mask[:, :cs]
    return mask


_MASKS = ("encoder_mask", "decoder_mask", "cycle_encoder_mask", "cycle_decoder_mask")
_TARGETS = ("org_h", "cv_h")


class _StepCounts:
    """Per-step registry: tensor identity -> count_local / count_global.  The announced tensors are kept alive
    by the registry until the next step, so an address can not be reused by another tensor while its entry
    is valid."""

    def __init__(self):
        self.factors, self.keep, self.pending = {}, [], None

    def clear(self):
        self.factors, self.keep, self.pending = {}, [], None

    # The counts' all-reduce is a handful of integers: it rides in the first VQ-EMA message of the step (C2, issued in the
    # middle of the generator's forward - before any loss is evaluated) instead of being a collective, and in a captured step
    # a graph segment, of its own.  ``pending`` = (local counts fp32, factors fp32: filled in place when the sums arrive).
    def board(self, seats):
        """Write the waiting counts into `seats` (int64 words of a message about to be summed); their number, or 0."""
        if self.pending is None or self.pending[0].numel() > seats.numel() or self.pending[0].device != seats.device:
            return 0
        n = self.pending[0].numel()
        seats[:n].copy_(self.pending[0])  # (element counts: integers, exact in either type)
        return n

    def resolve(self, totals):
        local, fac = self.pending
        torch.div(local, totals.to(torch.float32).clamp_min(1.0), out=fac)
        self.pending = None

    def flush(self):
        """No ride came before the first loss (no EMA codebook, an evaluation pass): the counts' own all-reduce."""
        if self.pending is not None:
            tot = self.pending[0].clone()
            all_reduce_sum(tot)
            self.resolve(tot)


_step = _StepCounts()


def prepare_step(batch, conf):
    """C3: announce every mask / target of `batch` the step's losses can normalise by and sum their element
    counts over the ranks with ONE all-reduce (fixed order: causal shifts x masks, then speaker targets)."""
    _step.clear()
    if not is_dist():
        return
    shifts = [0]
    if conf.get("causal") and conf.get("causal_size"):
        cs = int(conf["causal_size"])
        shifts += [cs, 2 * cs]
    # groups of same-shaped tensors, one after the other: a group's element counts are ONE stack + ONE sum
    groups = []
    for cs in shifts:
        g = [causal_mask_view(batch[name], cs) for name in _MASKS if isinstance(batch.get(name), torch.Tensor)]
        by_shape = {}
        for v in g:
            by_shape.setdefault(tuple(v.shape), []).append(v)
        groups += [("m", vs) for vs in by_shape.values()]
    tg = [batch[name].reshape(-1) for name in _TARGETS  # .reshape(-1): what the trainers hand to the cross entropy
          if isinstance(batch.get(name), torch.Tensor) and batch[name].is_contiguous()]
    by_shape = {}
    for v in tg:
        by_shape.setdefault(tuple(v.shape), []).append(v)
    groups += [("t", vs) for vs in by_shape.values()]
    if not groups:
        return
    views, counts = [], []
    for kind, vs in groups:
        st = torch.stack([v.reshape(-1) for v in vs])
        counts.append(((st != -100) if kind == "t" else st).sum(1).to(torch.float32))
        views += vs
    local = counts[0] if len(counts) == 1 else torch.cat(counts)
    fac = torch.empty_like(local)
    for i, v in enumerate(views):
        _step.factors[_key(v)] = fac[i]  # (views of `fac`: valid once the global counts have arrived)
    _step.keep = [batch, views]
    _step.pending = (local, fac)


def seed_shared_python_rng(seed=1234):
    """Seed Python's global RNG identically on every rank BEFORE the trainer is built: the trainer copies the
    state into its own generator (``BaseTrainer.rng``) for the choices inside the cyclegan / stargan losses
    (crank/net/trainer/trainer_cyclegan.py:166, trainer_stargan.py:91)."""
    random.seed(seed)


def shard_batch(batch, rank, world):
    """Rank r takes utterances [r*B/world, (r+1)*B/world) of a global batch."""
    out = {}
    for k, v in batch.items():
        n = len(v)
        per = n // world
        out[k] = v[rank * per:(rank + 1) * per]
        if isinstance(out[k], torch.Tensor):
            out[k] = out[k].contiguous()
    return out


class _DPLoss:
    """Wraps one criterion so that its value is this rank's share of the GLOBAL mean
    (C3).  kind: "masked" (x, y, mask=None, causal_size=0), "plain" (unmasked mean ->
    1/world) or "ce" (count = targets != ignore_index)."""

    def __init__(self, fn, kind):
        self.fn, self.kind = fn, kind

    @staticmethod
    def _factor(view, count_fn):
        if _step.pending is not None:
            _step.flush()
        f = _step.factors.get(_key(view))
        return f if f is not None else mean_rescale(count_fn())

    def recon(self, x, y, mask, causal_size, stft):
        """The fused (L1, MSE, STFT) op of the wrapped criterion (CustomFeatureLoss.recon), each term as this rank's
        share of its global mean; None where the criterion has no such op."""
        inner = getattr(self.fn, "recon", None)
        three = inner(x, y, mask, causal_size, stft.fn if isinstance(stft, _DPLoss) else stft) if inner else None
        if three is None:
            return None
        m = mask
        if mask is not None and getattr(self.fn, "causal", False) and causal_size != 0:
            m = causal_mask_view(mask, causal_size)
        world = dist.get_world_size()
        fm = self._factor(m, lambda: m.sum()) if m is not None else 1.0 / world
        return three[0] * fm, three[1] * fm, three[2] / world

    @property
    def ignore_index(self):
        return getattr(self.fn, "ignore_index", None) if self.kind == "ce" else None

    def scale_ce(self, value, target):
        """C3 for a cross entropy that was formed outside this wrapper (the fused classifier + loss op)."""
        ign = getattr(self.fn, "ignore_index", -100)
        return value * self._factor(target, lambda: (target != ign).sum())

    def __call__(self, *args, **kwargs):
        v = self.fn(*args, **kwargs)
        world = dist.get_world_size()
        if self.kind == "plain":
            return v / world
        if self.kind == "ce":
            tgt = args[1]
            ign = getattr(self.fn, "ignore_index", -100)
            return v * self._factor(tgt, lambda: (tgt != ign).sum())
        mask = kwargs.get("mask", args[2] if len(args) > 2 else None)
        if mask is None:
            return v / world
        cs = kwargs.get("causal_size", args[3] if len(args) > 3 else 0)
        if getattr(self.fn, "causal", False) and cs != 0:
            mask = causal_mask_view(mask, cs)
        return v * self._factor(mask, lambda: mask.sum())


def scale_masked_mean(value, mask):
    """C3 for a masked mean that was not formed by a wrapped criterion (the commitment loss out of the quantizer op):
    this rank's share of the global mean."""
    if not is_dist():
        return value
    return value * _DPLoss._factor(mask, lambda: mask.sum())


def wrap_criterion(criterion):
    """DP view of the trainers' criterion dict (``prepare_step`` is called once per step by the trainer)."""
    kinds = {"mse": "plain", "l1": "plain", "kld": "plain", "ce": "ce", "fmse": "masked", "fl1": "masked",
             "fstft": "plain"}
    return {k: _DPLoss(v, kinds[k]) for k, v in criterion.items() if k in kinds}


def install(models=None):
    """Wire C1 into the product: returns the grad-reduce callable for get_optimizer.  (C2 lives in the
    generator's ``EmaBucket``, C3 in ``prepare_step`` / ``wrap_criterion``; both look at ``is_dist()``.)"""
    return grad_allreduce if is_dist() else None
