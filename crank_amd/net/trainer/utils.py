"""Factories with the reference's names and return shapes
(crank/net/trainer/utils.py:22-74): criterion dict, one optimizer and one StepLR-like
scheduler per model.  Optimizers act on a model's flat parameter block with one HIP
launch; under data parallelism they first all-reduce the flat gradient block.
"""
import torch
from torch import nn

from ... import ops, parallel
from ..module.loss import CrossEntropyLoss, CustomFeatureLoss, MeanLoss


def get_criterion(conf, device="cuda"):
    return {
        "mse": MeanLoss("mse"),
        "l1": MeanLoss("l1"),
        "ce": CrossEntropyLoss(ignore_index=-100),
        "kld": nn.KLDivLoss(reduction="mean"),  # never called by the trainers
        "fmse": CustomFeatureLoss(loss_type="mse", causal=conf["causal"], device=device),
        "fl1": CustomFeatureLoss(loss_type="l1", causal=conf["causal"], device=device),
        "fstft": CustomFeatureLoss(loss_type="stft", stft_params=conf["stft_params"], causal=conf["causal"],
                                   device=device),
    }


class FlatAdam:
    """torch.optim.Adam(lr) semantics (defaults betas (0.9,0.999), eps 1e-8) on a
    FlatModel.  lr and the step counter live on the device (no host sync per step)."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, grad_reduce_fn=None):
        self.model = model
        self.base_lr = float(lr)
        self.betas, self.eps = betas, eps
        dev = model.flat.device
        self.lr_dev = torch.tensor([lr], device=dev, dtype=torch.float32)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(model.flat.data)
        self.exp_avg_sq = torch.zeros_like(model.flat.data)
        self.grad_reduce_fn = grad_reduce_fn
        self._reduced = False  # the gradient block already holds the sum over the ranks (reduce_grads before a clip)
        self._inflight = None  # a started, not yet completed all-reduce of the gradient block (reduce_grads_start)
        # the update zeroes the gradient block as it reads it, so the next zero_grad() is free; set False to keep the
        # gradients readable after step() like torch.optim.Adam does
        self.clear_grads = True
        self.param_groups = [{"lr": float(lr), "params": [model.flat]}]

    def state_dict(self):
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.step_dev.clone(),
                "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        """Moments, step count and learning rate of a saved FlatAdam.  A state whose moments are not zero over an EMA
        codebook (a run that trained its codebooks by gradient) would move that codebook at every step although its
        gradient is zero: the model is told, so that every optimizer step ages its search images again."""
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_dev.copy_(sd["step"])
        self.set_lr(sd["lr"])
        for off, n in getattr(self.model, "ema_codebook_ranges", lambda: [])():
            if bool(self.exp_avg[off: off + n].any()) or bool(self.exp_avg_sq[off: off + n].any()):
                self.model._trained_codebooks = True

    def set_lr(self, lr):
        if float(lr) != self.param_groups[0]["lr"]:
            self.param_groups[0]["lr"] = float(lr)
            self.lr_dev.fill_(float(lr))

    def zero_grad(self, set_to_none=False):
        self.model.zero_grad()

    def reduce_grads_start(self):
        """Start C1 without waiting for it (RCCL: on the collective's own stream); ``reduce_grads`` / ``step`` complete it.
        What the caller enqueues in between - the update of a model that does not read this one's parameters - runs in the
        shadow of the all-reduce."""
        if self.grad_reduce_fn is not None and not self._reduced and self._inflight is None:
            self._inflight = parallel.grad_allreduce_start(self.model.grad_flat)

    def mark_reduced(self):
        """The caller has summed this model's gradient block over the ranks itself (together with another model's: one
        exchange for both); ``step`` / ``reduce_grads`` then leave it alone."""
        if self.grad_reduce_fn is not None:
            self._reduced = True

    def reduce_grads(self):
        """C1: sum this model's gradient block over the ranks, once per step.  ``step`` does it itself; a
        caller that needs the GLOBAL gradient before the update - gradient-norm clipping: N ranks x B utterances must
        clip like one batch of N*B - calls it first."""
        if self._inflight is not None:
            self._inflight.finish()
            self._inflight, self._reduced = None, True
        if self.grad_reduce_fn is not None and not self._reduced:
            self.grad_reduce_fn(self.model.grad_flat)
            self._reduced = True

    def step(self, defer_bump=False):
        """defer_bump: the caller advances the step count with the launch that follows
        (``model.prepare_nets(bump_step=optimizer.step_dev)``)."""
        m = self.model
        self.reduce_grads()
        self._reduced = False
        self._update(m, defer_bump)
        if self.clear_grads:
            m.grads_clean = True
        m.touch(by_optimizer=True)

    def _update(self, m, defer_bump):
        ops.adam_step(m.flat.data, m.grad_flat, self.exp_avg, self.exp_avg_sq, self.lr_dev, self.step_dev,
                      self.betas[0], self.betas[1], self.eps, clear_grads=self.clear_grads, defer_bump=defer_bump)


class FlatRAdam(FlatAdam):
    """torch_optimizer.RAdam(lr) semantics (crank/net/trainer/utils.py:44-45; defaults betas (0.9, 0.999), eps 1e-8, no
    weight decay) on a FlatModel: one launch, lr and step count on the device like FlatAdam's.  The package is absent from
    the reference tree; the update is its published one (``crk_radam_step``, ``oracle/optim.py::RAdam``)."""

    def _update(self, m, defer_bump):
        ops.radam_step(m.flat.data, m.grad_flat, self.exp_avg, self.exp_avg_sq, self.lr_dev, self.step_dev,
                       self.betas[0], self.betas[1], self.eps, clear_grads=self.clear_grads, defer_bump=defer_bump)


class FlatLamb(FlatAdam):
    """pytorch_lamb.Lamb(lr) semantics (crank/net/trainer/utils.py:46-47; defaults betas (0.9, 0.999), eps 1e-6, no weight
    decay, no bias correction, trust ratio per parameter tensor with the weight norm clamped to [0, 10]) on a FlatModel.
    The parameter tensors are the model's state-dict entries (the tensors ``model.parameters()`` yields in the reference:
    ``weight_g`` / ``weight_v`` / ``bias`` of every conv, embeddings, codebooks); whatever the flat block holds between
    them is a tensor of its own.  Two launches per step, fixed summation order (``crk_lamb_step``)."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-6, grad_reduce_fn=None):
        super().__init__(model, lr, betas=betas, eps=eps, grad_reduce_fn=grad_reduce_fn)
        n = model.flat.numel()
        spans, pos = [], 0
        for _, off, shp in sorted(model._entries, key=lambda e: e[1]):
            size = 1
            for d in shp:
                size *= int(d)
            if off > pos:
                spans.append((pos, off - pos))
            if size:
                spans.append((off, size))
            pos = max(pos, off + size)
        if pos < n:
            spans.append((pos, n - pos))
        tile = ops.lamb_tile()
        tiles, tensors = [], []
        for s, (off, size) in enumerate(spans):
            first = len(tiles)
            for o in range(0, size, tile):
                tiles.append((off + o, min(tile, size - o), s, 0))
            tensors.append((first, len(tiles) - first))
        dev = model.flat.device
        self.tensor_spans = spans
        self.tiles = torch.tensor(tiles, dtype=torch.int32, device=dev).reshape(-1, 4).contiguous()
        self.tensors = torch.tensor(tensors, dtype=torch.int32, device=dev).reshape(-1, 2).contiguous()
        self.upd = torch.zeros(n, device=dev, dtype=torch.float32)
        self.part = torch.zeros(2 * max(1, len(tiles)), device=dev, dtype=torch.float32)
        self.trust_ratio = torch.ones(max(1, len(tensors)), device=dev, dtype=torch.float32)

    def _update(self, m, defer_bump):
        ops.lamb_step(m.flat.data, m.grad_flat, self.exp_avg, self.exp_avg_sq, self.upd, self.tiles, self.tensors, self.part,
                      self.trust_ratio, self.lr_dev, self.step_dev, self.betas[0], self.betas[1], self.eps,
                      clear_grads=self.clear_grads, defer_bump=defer_bump)


_OPTIMIZERS = {"adam": FlatAdam, "radam": FlatRAdam, "lamb": FlatLamb}


def get_optimizer(conf, model, grad_reduce_fn=None):
    """crank/net/trainer/utils.py:40-58: adam / radam / lamb per model, anything else is the reference's ValueError."""
    optimizer = {}
    for m in ["G", "D", "C", "SPKRADV"]:
        if m in model:
            t = conf["optim"][m]["type"]
            if t not in _OPTIMIZERS:
                raise ValueError("Invalid optimizer type")
            optimizer[m] = _OPTIMIZERS[t](model[m], conf["optim"][m]["lr"], grad_reduce_fn=grad_reduce_fn)
    return optimizer


class StepLR:
    """StepLR stepped with an explicit step count, as the reference drives it
    (crank/net/trainer/basetrainer.py:84-90,239-247): lr = base * gamma ** (steps // size)."""

    def __init__(self, optimizer, step_size, gamma):
        self.optimizer, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.last_epoch = 0

    def step(self, steps=None):
        self.last_epoch = self.last_epoch + 1 if steps is None else int(steps)
        self.optimizer.set_lr(self.optimizer.base_lr * self.gamma ** (self.last_epoch // self.step_size))

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]


def get_scheduler(conf, optimizer):
    scheduler = {}
    for m in ["G", "D", "C", "SPKRADV"]:
        if m in optimizer:
            scheduler[m] = StepLR(optimizer[m], conf["optim"][m]["decay_step_size"], conf["optim"][m]["decay_size"])
    return scheduler


def clip_grad_norm(model, max_norm):
    """torch.nn.utils.clip_grad_norm_ on the flat gradient block (plumbing, rarely on:
    clip_grad_norm defaults to 0.0 in every recipe)."""
    g = model.grad_flat
    total = torch.linalg.vector_norm(g)
    g.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
    return total


def get_dataloader(conf, scp, scaler, flag="train", n_jobs=0, reader=None, device="cuda"):
    """crank/net/trainer/utils.py:77-106 over device-resident corpora: the same dict
    ({"spkrs", "train", "dev", "eval"}) with loaders that assemble each batch in HBM
    (crank_amd/net/trainer/dataset.py).  ``n_jobs`` is accepted and unused: there are no
    worker processes.  For decoding flags the batch is re-shaped like the reference does:
    batch_len = longest utterance, batch_size = tokens // batch_len."""
    from .dataset import BaseDataset, DeviceLoader, calculate_maxflen

    if flag in ["train", "reconstruction"]:
        feats = list(scp["train"]["feats"].values()) + list(scp["dev"]["feats"].values())
    elif flag in ["eval"]:
        feats = list(scp["eval"]["feats"].values())
    else:
        raise ValueError(f"unknown flag {flag}")
    if flag in ["reconstruction", "eval"]:
        token_size = conf["batch_len"] * conf["batch_size"]
        conf["batch_len"] = calculate_maxflen(feats, reader=reader)
        conf["batch_size"] = token_size // conf["batch_len"]
    spkrs = dict(zip(scp["train"]["spkrs"], range(len(scp["train"]["spkrs"]))))
    out = {"spkrs": spkrs}
    for phase, shuffle in (("train", True), ("dev", True), ("eval", False)):
        if phase in scp and scp[phase].get("feats"):
            dset = BaseDataset(conf, scp, scaler, phase=phase, reader=reader, device=device)
            out[phase] = DeviceLoader(dset, conf["batch_size"], shuffle=shuffle, rank=parallel.rank() if phase != "eval" else 0,
                                      world_size=parallel.world_size() if phase != "eval" else 1)
    return out
