"""Batch assembly from a corpus resident in HBM.

Mirror of the reference's ``BaseDataset`` + ``DataLoader`` pair (crank/net/trainer/dataset.py:22-198,
crank/net/trainer/utils.py:77-106) with the work moved to where the data is: the reference reads
HDF5 files in worker processes, normalises / crops / pads every sample in numpy, collates on the
host and copies ~25 MB per batch over PCIe; here every utterance is read ONCE, packed frame-major
into HBM (VCC2020's training set is ~0.2 GB of the 288 GB), normalised once by
``crk_scaler_apply`` and each batch is one ``crk_collate_batch`` launch that writes the collated,
padded batch dict directly in device memory.  The kernels reproduce numpy / sklearn bit for bit
(tests/test_gpu_dataset.py against the reference-generated tests/golden/dataset.npz).

The two random draws of a sample -- the conversion-target speaker (dataset.py:84-86) and the first
kept frame of an over-long utterance (dataset.py:161) -- are made on the host with Python's
``random`` in the reference's order, so a seeded run sees the same stream as the reference with
``num_workers=0``.

With ``use_raw`` the waveform of every row is padded / cropped like ``padding_raw`` (dataset.py:261-285),
as float32 (the reference's dtype depends on the branch taken).

``cache_dataset``: the corpus is resident anyway; what the reference's sample cache freezes - the drawn
conversion target and crop start of an utterance (quirk Q9) - is memoised per utterance.  Not reproduced: ``spec_augment`` (the reference raises
NotImplementedError, dataset.py:114) and the never-taken "excit" branch (dataset.py:111-112).
"""
import ctypes
import random
from pathlib import Path

import numpy as np
import torch

from crank_amd import _lib
from crank_amd._lib import check, ptr, stream_ptr

MASK_KEYS = ("encoder_mask", "decoder_mask", "cycle_encoder_mask", "cycle_decoder_mask")


def read_feature(h5f, ext="mlfb"):
    """dataset.py:229-236.  HDF5 needs h5py, which this image does not carry: pass ``reader=`` to
    BaseDataset to serve features from elsewhere."""
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - environment dependent
        raise RuntimeError("reading HDF5 feature files needs h5py; pass reader=callable(h5f, ext) instead") from e
    with h5py.File(h5f, "r") as fp:
        data = fp[ext][:]
    return data[:, np.newaxis] if data.ndim == 1 else data


def _f64(a, device):
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1)), device=device)


class ScalerStats:
    """The numbers batch assembly and decoding need from the reference's ``scaler.pkl`` (a dict of
    sklearn StandardScalers: ``scaler[feat]`` and ``scaler[spkr]["lcf0"]``), on the device as float64."""

    def __init__(self, scaler, spkrlist, feat_types, ignore, device):
        self.device = device
        self.feat = {}
        for k in feat_types:
            if scaler is not None and k not in ("uv", "cap") and k not in ignore:
                self.feat[k] = (_f64(scaler[k].mean_, device), _f64(scaler[k].scale_, device))
        self.lcf0_host = None
        if "lcf0" in self.feat:
            self.lcf0_host = (float(np.asarray(scaler["lcf0"].mean_).reshape(-1)[0]), float(np.asarray(scaler["lcf0"].scale_).reshape(-1)[0]))
        if scaler is not None:
            self.spk_mean = _f64([np.asarray(scaler[s]["lcf0"].mean_).reshape(-1)[0] for s in spkrlist], device)
            # np.sqrt of the float64 variance is taken on the host, as convert_f0 does (dataset.py:288-293)
            self.spk_std = _f64([np.sqrt(np.asarray(scaler[s]["lcf0"].var_).reshape(-1)[0]) for s in spkrlist], device)
        else:
            self.spk_mean = self.spk_std = None


def scaler_apply(x, mean, scale, inverse=False):
    """(N, D) float32 device tensor -> new tensor, sklearn's float32 transform / inverse_transform."""
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    y = torch.empty_like(x2)
    if x2.shape[-1] != mean.numel():
        raise ValueError(f"scaler has {mean.numel()} dimensions, features have {x2.shape[-1]}")
    check(_lib.lib().crk_scaler_apply(ptr(x2), x2.shape[1], ptr(y), y.shape[1], x2.shape[0], x2.shape[1], ptr(mean), ptr(scale),
                                      1 if inverse else 0, stream_ptr()), "scaler_apply")
    return y.reshape(x.shape)


class BaseDataset:
    """``BaseDataset(conf, scp, scaler, phase)`` of the reference; ``reader(h5f, ext) -> ndarray``
    replaces its HDF5 access.  ``assemble(indices)`` returns the collated batch dict on the device;
    ``dataset[idx]`` returns the corresponding single sample (tensors without the batch axis)."""

    def __init__(self, conf, scp, scaler, phase="train", reader=None, device="cuda"):
        if conf.get("spec_augment"):
            raise NotImplementedError("SpecAugument currently disabled.")  # dataset.py:114
        self.conf, self.device = conf, torch.device(device)
        self.h5list = list(scp[phase]["feats"].values())
        self.spkrlist = list(scp["train"]["spkrs"])
        self.spkrdict = dict(zip(self.spkrlist, range(len(self.spkrlist))))
        self.n_spkrs = len(self.spkrdict)
        self.batch_len = conf["batch_len"]
        self.in_type, self.out_type = conf["input_feat_type"], conf["output_feat_type"]
        reader = reader or read_feature
        ignore = list(conf.get("ignore_scaler", []))
        types = [self.in_type] + ([self.out_type] if self.out_type != self.in_type else [])
        self.has_cap = "mcep" in types
        self.stats = ScalerStats(scaler, self.spkrlist, types + ["lcf0"], ignore, self.device)

        # ---- read every utterance once and pack the corpus ----
        cols = {k: [] for k in types + ["lcf0", "uv"] + (["cap"] if self.has_cap else [])}
        lens, spk = [], []
        self.flbl, self.org_names = [], []
        for f in self.h5list:
            f = Path(f)
            for k in cols:
                a = np.asarray(reader(str(f), ext=k), dtype=np.float32)
                cols[k].append(a[:, None] if a.ndim == 1 else a)
            lens.append(cols[self.in_type][-1].shape[0])
            self.flbl.append(str(Path(f.parent.stem) / f.stem))  # dataset.py:80-82
            self.org_names.append(str(f.parent.stem))
            spk.append(self.spkrdict[self.org_names[-1]])
        self.lens = lens
        dev = self.device
        self.utt_start = torch.as_tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), device=dev)
        self.utt_spk = torch.as_tensor(np.asarray(spk, dtype=np.int32), device=dev)
        self.packed = {}
        for k, parts in cols.items():
            raw = torch.as_tensor(np.ascontiguousarray(np.concatenate(parts)), device=dev)
            if k == "lcf0":
                self.lcf0_raw = raw.reshape(-1).contiguous()
            self.packed[k] = scaler_apply(raw, *self.stats.feat[k]) if k in self.stats.feat else raw
        self.drop_0th = "mcep" in types and not conf.get("use_mcep_0th", False)
        self.use_raw = bool(conf.get("use_raw"))
        if self.use_raw:  # waveforms, packed sample after sample (dataset.py:42-43)
            waves = [np.asarray(reader(str(f), ext="raw"), dtype=np.float32).reshape(-1) for f in self.h5list]
            self.raw = torch.as_tensor(np.ascontiguousarray(np.concatenate(waves)), device=dev)
            self.raw_start = torch.as_tensor(np.concatenate([[0], np.cumsum([w.size for w in waves])]).astype(np.int64), device=dev)
            self.fftl, self.hop = int(conf["feature"]["fftl"]), int(conf["feature"]["hop_size"])

    def __len__(self):
        return len(self.h5list)

    # the reference's draws, in its order (dataset.py:84-86 then :161 for each sample).  With cache_dataset
    # (default true) the reference keeps the finished sample of an utterance after its first visit
    # (dataset.py:59-60,72-73), i.e. conversion target and crop start are drawn ONCE per utterance and reused in
    # every later epoch (quirk Q9): the drawn pair is memoised the same way.
    def _draw(self, idx):
        memo = getattr(self, "_draw_memo", None)
        if memo is None:
            memo = self._draw_memo = {}
        cached = bool(self.conf.get("cache_dataset", False)) if hasattr(self, "conf") else False
        if cached and idx in memo:
            return memo[idx]
        org = self.org_names[idx]
        cv_name = random.choice([s for s in list(self.spkrdict.keys()) if s != org])
        diff = self.batch_len - self.lens[idx]
        p = random.choice(range(0, abs(diff))) if diff < 0 else 0
        if cached:
            memo[idx] = (cv_name, p)
        return cv_name, p

    def assemble(self, indices, draws=None):
        """Collated batch for the utterances ``indices``; ``draws`` = [(cv speaker name, p)] overrides the RNG."""
        indices = [int(i) for i in indices]
        for i in indices:
            if not 0 <= i < len(self):
                raise IndexError(f"utterance index {i} out of range")
        draws = [self._draw(i) for i in indices] if draws is None else list(draws)
        B, T, S, dev = len(indices), self.batch_len, self.n_spkrs, self.device
        cv_idx = [self.spkrdict[c] for c, _ in draws]
        for i, (_, p) in zip(indices, draws):
            if p < 0 or (p and p + T > self.lens[i]):
                raise ValueError(f"crop start {p} does not fit utterance {i} of {self.lens[i]} frames")
        picks = torch.as_tensor(np.asarray([indices, [p for _, p in draws], cv_idx], dtype=np.int32), device=dev)

        f32 = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
        batch, streams = {}, []

        def stream(key, src, col0, ncols):
            batch[key] = f32(B, T, ncols)
            streams.append((src, src.shape[1], col0, ncols, batch[key]))

        for role, k in (("in_feats", self.in_type), ("out_feats", self.out_type)):
            src = self.packed[k]
            first = 1 if (k == "mcep" and self.drop_0th) else 0
            stream(role, src, first, src.shape[1] - first)
        if self.drop_0th:
            stream("mcep_0th", self.packed["mcep"], 0, 1)
        stream("lcf0", self.packed["lcf0"], 0, 1)
        stream("uv", self.packed["uv"], 0, 1)
        if self.has_cap:
            stream("cap", self.packed["cap"], 0, self.packed["cap"].shape[1])
        if self.in_type != self.out_type:  # the reference keeps the input feature under its own name (dataset.py:133-136)
            batch[self.in_type] = batch["in_feats"]

        desc = _lib.CollateDesc()
        desc.n_streams = len(streams)
        for j, (src, ld, c0, nc, dst) in enumerate(streams):
            desc.streams[j] = _lib.CollateStream(ptr(src), ld, c0, nc, ptr(dst))
        desc.utt_start, desc.utt_spk, desc.n_utt, desc.n_spk = ptr(self.utt_start), ptr(self.utt_spk), len(self), S
        have_f0 = self.stats.spk_mean is not None
        if have_f0:
            desc.lcf0_raw, desc.spk_lcf0_mean, desc.spk_lcf0_std = ptr(self.lcf0_raw), ptr(self.stats.spk_mean), ptr(self.stats.spk_std)
            batch["cv_lcf0"] = f32(B, T, 1)
        batch["org_h"] = torch.empty(B, T, device=dev, dtype=torch.int64)
        batch["cv_h"] = torch.empty(B, T, device=dev, dtype=torch.int64)
        batch["org_h_onehot"], batch["cv_h_onehot"] = f32(B, T, S), f32(B, T, S)
        mask = torch.empty(B, T, 1, device=dev, dtype=torch.bool)
        batch["flen"] = torch.empty(B, device=dev, dtype=torch.int64)
        if self.use_raw:  # padding_raw, dataset.py:261-285: fftl + hop * T - 1 samples per row
            desc.raw, desc.raw_start, desc.fftl, desc.hop = ptr(self.raw), ptr(self.raw_start), self.fftl, self.hop
            batch["raw"] = f32(B, self.fftl + self.hop * T - 1)
        check(_lib.lib().crk_collate_batch(ctypes.byref(desc), ptr(picks), B, T, ptr(batch.get("cv_lcf0")), ptr(batch["org_h"]),
                                           ptr(batch["cv_h"]), ptr(batch["org_h_onehot"]), ptr(batch["cv_h_onehot"]), ptr(mask),
                                           ptr(batch["flen"]), ptr(batch.get("raw")), stream_ptr()), "collate_batch")
        for k in MASK_KEYS:  # four independent copies in the reference (dataset.py:118-125)
            batch[k] = mask.clone()
        batch["flbl"] = [self.flbl[i] for i in indices]
        batch["org_spkr_name"] = [self.org_names[i] for i in indices]
        batch["cv_spkr_name"] = [c for c, _ in draws]
        return batch

    def __getitem__(self, idx):
        b = self.assemble([idx])
        return {k: (v[0] if isinstance(v, (torch.Tensor, list)) else v) for k, v in b.items()}


class DeviceLoader:
    """What ``DataLoader(dataset, batch_size, shuffle)`` is to the reference (utils.py:94-105): an
    iterable of collated batches, one pass over the utterances per iteration, the last batch short.

    Data parallel: ``batch_size`` is the PER-RANK batch; every rank draws the same
    permutation (seed + epoch, a generator of its own: no hidden coupling to the global torch RNG) and
    rank r assembles utterances [r * batch_size, (r + 1) * batch_size) of each global batch of
    world_size * batch_size utterances.  A trailing global batch that cannot give every rank at least
    one utterance is dropped, so all ranks run the same number of steps (collectives stay matched)."""

    def __init__(self, dataset, batch_size, shuffle=False, drop_last=False, rank=0, world_size=1, seed=0):
        if not 0 <= rank < world_size:
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0

    def _global_batches(self):
        n, gb = len(self.dataset), self.batch_size * self.world_size
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        out = []
        for i in range(0, n, gb):
            chunk = order[i : i + gb]
            if len(chunk) < gb and (self.drop_last or len(chunk) < self.world_size):
                break
            out.append(chunk)
        return out

    def __len__(self):
        return len(self._global_batches())

    def __iter__(self):
        batches = self._global_batches()
        self.epoch += 1
        for chunk in batches:
            if len(chunk) == self.batch_size * self.world_size:
                mine = chunk[self.rank * self.batch_size : (self.rank + 1) * self.batch_size]
            else:  # short last batch: spread as evenly as it goes
                per, extra = divmod(len(chunk), self.world_size)
                a = self.rank * per + min(self.rank, extra)
                mine = chunk[a : a + per + (1 if self.rank < extra else 0)]
            yield self.dataset.assemble(mine)


def calculate_maxflen(flist, reader=None):
    """dataset.py:279-285: the longest utterance of a list (decode batches are padded to it)."""
    reader = reader or read_feature
    return max(reader(str(f), ext="mlfb").shape[0] for f in flist)
