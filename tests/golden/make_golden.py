#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the authoring container (needs /root/reference).  Nothing from the
reference is copied: the fixtures hold inputs and expected outputs only.

Absent third-party packages are replaced by import-only placeholders with no
arithmetic, except ``parallel_wavegan.models`` which is bound to the repo's own
restatement ``oracle/pwg.py`` (the package is un-vendored and not installable
here).  Consequently:

* quantizer.npz / losses.npz / stft_layer.npz / misc.npz pin ``oracle/modules.py``
  (and the HIP kernels) against the reference's OWN classes
  (``crank.net.module.vqvae2.Quantizer``, ``crank.net.module.loss.*``,
  ``crank.net.module.mlfb.STFTLayer/MLFBScalerLayer``, ``GradientReversalLayer``,
  ``BaseTrainer._get_dec_h``, StepLR stepping).
* step_*.npz come from the reference's own ``VQVAE2`` / ``SpeakerAdversarialNetwork``
  / ``get_model`` / trainer classes running one optimisation step, with the conv
  stacks supplied by ``oracle/pwg.py``: they pin the wiring, loss algebra and update
  order of the reference, not the third-party conv code.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import copy
import os
import random
import sys
import types
import warnings

sys.dont_write_bytecode = True
warnings.simplefilter("ignore")

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from oracle import pwg as oracle_pwg  # noqa: E402
from crank_amd.synthetic import deterministic_state, make_batch  # noqa: E402


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return lambda *a, **kw: None


pw = _placeholder("parallel_wavegan")
pw.models = _placeholder(
    "parallel_wavegan.models",
    ParallelWaveGANGenerator=oracle_pwg.ParallelWaveGANGenerator,
    ParallelWaveGANDiscriminator=oracle_pwg.ParallelWaveGANDiscriminator,
    ResidualParallelWaveGANDiscriminator=oracle_pwg.ResidualParallelWaveGANDiscriminator,
)
_placeholder("parallel_wavegan.bin")
_placeholder("parallel_wavegan.bin.preprocess", logmelfilterbank=None)
lib = _placeholder("librosa")
# the mel basis (librosa absent): the repo's restatement of librosa.filters.mel's published defaults
# (oracle/modules.py slaney_mel_basis) - only the use_raw step case reaches it
from oracle.modules import slaney_mel_basis as _oracle_mel  # noqa: E402
lib.filters = _placeholder("librosa.filters", mel=lambda sr, n_fft, n_mels, fmin, fmax: _oracle_mel(sr, n_fft, n_mels, fmin, fmax))
_placeholder("soundfile")
_placeholder("h5py")
sp = _placeholder("sprocket")
sp.util = _placeholder("sprocket.util", HDF5=_Dummy)
sp.speech = _placeholder("sprocket.speech", Synthesizer=_Dummy, FeatureExtractor=_Dummy)
_placeholder("typeguard", check_argument_types=lambda: True)
_placeholder("tensorboardX", SummaryWriter=_Dummy)
_placeholder("torch_optimizer")
_placeholder("pytorch_lamb", Lamb=_Dummy)

from crank.net.module.vqvae2 import Quantizer, VQVAE2  # noqa: E402
from crank.net.module import loss as ref_loss  # noqa: E402
from crank.net.module.mlfb import STFTLayer, MLFBScalerLayer  # noqa: E402
from crank.net.module.spkradv import GradientReversalLayer  # noqa: E402
from crank.net.trainer import TrainerWrapper  # noqa: E402
from crank.net.trainer.utils import get_criterion, get_optimizer, get_scheduler  # noqa: E402
from crank.bin.train import get_model  # noqa: E402


def np_(t):
    return t.detach().cpu().numpy().copy()


# ----------------------------------------------------------------------------
def gen_quantizer():
    out = {}
    torch.manual_seed(11)
    K, D, B, T = 512, 64, 3, 40
    q = Quantizer(D, K, ema_flag=True, bdt_flag=True)
    q.train()
    out["init_weight"] = np_(q.embedding.weight)
    out["init_ema_w"] = np_(q.ema_w)
    out["init_ema_size"] = np_(q.ema_size)
    rs = np.random.RandomState(5)
    for it in range(3):
        x = torch.from_numpy((rs.standard_normal((B, D, T)) * (0.01 if it == 0 else 1.0)).astype(np.float32))
        if it == 2:
            x[:, :, -7:] = 0.0  # all-pad frames
        e, qx, idx = q(x, use_ema=True)
        out[f"x{it}"], out[f"e{it}"], out[f"qx{it}"], out[f"idx{it}"] = np_(x), np_(e), np_(qx), np_(idx)
        out[f"w{it}"], out[f"ema_w{it}"], out[f"ema_size{it}"] = np_(q.embedding.weight), np_(q.ema_w), np_(q.ema_size)
    x = torch.from_numpy(rs.standard_normal((B, D, T)).astype(np.float32))
    e, qx, idx = q(x, use_ema=False)
    out["x3"], out["e3"], out["qx3"], out["idx3"] = np_(x), np_(e), np_(qx), np_(idx)
    out["w3"] = np_(q.embedding.weight)

    # adversarial codebooks: exact ties (duplicate rows), near ties, btd layout
    q2 = Quantizer(D, K, ema_flag=False, bdt_flag=False)
    q2.eval()
    w = rs.standard_normal((K, D)).astype(np.float32) * 0.5
    w[100] = w[7]  # exact duplicates -> lowest index wins
    w[300] = w[7]
    w[411] = w[20] + 1e-3 * rs.standard_normal(D).astype(np.float32)  # near tie
    q2.embedding.weight.data.copy_(torch.from_numpy(w))
    x = rs.standard_normal((2, 33, D)).astype(np.float32)
    x[0, :5] = w[7] + 1e-2 * rs.standard_normal((5, D)).astype(np.float32)
    x[0, 5:9] = w[300]
    x[1, :6] = 0.5 * (w[20] + w[411]) + 1e-2 * rs.standard_normal((6, D)).astype(np.float32)
    x[1, 6:8] = 0.0
    e, qx, idx = q2(torch.from_numpy(x))
    out["tie_w"], out["tie_x"], out["tie_idx"], out["tie_e"] = w, x, np_(idx), np_(e)
    np.savez_compressed(os.path.join(HERE, "quantizer.npz"), **out)
    print("quantizer.npz", {k: v.shape for k, v in list(out.items())[:4]})


def gen_quantizer_full():
    """SURVEY 8(c).1 at the benchmark size: 32 000 frames through the reference Quantizer (K=512, D=64) whose
    codebook has gone through two EMA updates (unused codes at ~1e5, quirk Q2).  Only what cannot be regenerated
    is stored: the codebook and the int16 indices; the frames come from RandomState(1234) in the test."""
    torch.manual_seed(21)
    K, D, B, T = 512, 64, 64, 500
    q = Quantizer(D, K, ema_flag=True, bdt_flag=True)
    q.train()
    rs = np.random.RandomState(4321)
    # a codebook in use: 448 codes with trained-like statistics (ema_w = size * code), 64 codes never used so far
    # (zero size, randn ema_w: the reference's initial state) - the first EMA update throws those to ~1e5 (Q2).
    # From the reference's raw init a handful of updates collapses every frame onto one code, which would pin nothing.
    w0 = (0.8 * rs.standard_normal((K, D))).astype(np.float32)
    size0 = np.where(np.arange(K) < 448, 20.0, 0.0).astype(np.float32)
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(w0))
        q.ema_size.copy_(torch.from_numpy(size0))
        q.ema_w[:, :448] = torch.from_numpy((w0[:448] * size0[:448, None]).T.copy())
    for it in range(2):
        x = torch.from_numpy(rs.standard_normal((8, D, T)).astype(np.float32))
        q(x, use_ema=True)
    x = torch.from_numpy(np.random.RandomState(1234).standard_normal((B, D, T)).astype(np.float32))
    e, qx, idx = q(x, use_ema=False)
    w = np_(q.embedding.weight)
    idx = np_(idx)
    assert idx.shape == (B, T) and idx.max() < 2 ** 15
    d64 = ((w.astype(np.float64) ** 2).sum(1)[None] - 2 * x.numpy().transpose(0, 2, 1).reshape(-1, D).astype(np.float64) @ w.astype(np.float64).T)
    print("quantizer_full.npz: distinct codes used", len(np.unique(idx)), "| agreement with an fp64 argmin",
          float((d64.argmin(1) == idx.reshape(-1)).mean()), "| largest |w|", float(np.abs(w).max()))
    np.savez_compressed(os.path.join(HERE, "quantizer_full.npz"), codebook=w, idx=idx.astype(np.int16),
                        x_seed=np.array(1234), x_shape_BDT=np.array([B, D, T]))


# ----------------------------------------------------------------------------
def gen_losses():
    out = {}
    rs = np.random.RandomState(6)
    B, T, Dm = 3, 120, 80
    x = torch.from_numpy(rs.standard_normal((B, T, Dm)).astype(np.float32)).requires_grad_(True)
    y = torch.from_numpy(rs.standard_normal((B, T, Dm)).astype(np.float32))
    flen = np.array([120, 77, 101])
    mask = torch.from_numpy((np.arange(T)[None, :] < flen[:, None])[:, :, None])
    out["x"], out["y"], out["mask"] = np_(x), np_(y), np_(mask)
    stft_params = {"fft_sizes": [64, 128], "win_sizes": [64, 128], "hop_sizes": [16, 32], "logratio": 0}
    for causal in [False, True]:
        for cs in ([0] if not causal else [-8, -2, 0, 2, 8]):
            for lt in ["l1", "mse", "stft"]:
                kw = dict(stft_params=stft_params, device="cpu") if lt == "stft" else dict(device="cpu")
                crit = ref_loss.CustomFeatureLoss(loss_type=lt, causal=causal, **kw)
                if x.grad is not None:
                    x.grad = None
                m = None if lt == "stft" else mask
                v = crit(x, y, mask=m, causal_size=cs)
                v.backward()
                tag = f"{lt}_c{int(causal)}_cs{cs}"
                out[f"val_{tag}"] = np_(v)
                out[f"grad_{tag}"] = np_(x.grad)
    # unmasked l1/mse
    for lt in ["l1", "mse"]:
        crit = ref_loss.CustomFeatureLoss(loss_type=lt, causal=False, device="cpu")
        x.grad = None
        v = crit(x, y)
        v.backward()
        out[f"val_{lt}_nomask"], out[f"grad_{lt}_nomask"] = np_(v), np_(x.grad)
    # directly-constructed STFTLoss (double swap cancels) and logratio != 0
    sl = ref_loss.STFTLoss(fft_size=32, win_size=20, hop_size=10, logratio=0.3, device="cpu")
    x.grad = None
    v = sl(x, y)
    v.backward()
    out["val_stftloss_direct"], out["grad_stftloss_direct"] = np_(v), np_(x.grad)
    ms = ref_loss.MultiSizeSTFTLoss(fft_sizes=[32, 64], win_sizes=[32, 64], hop_sizes=[8, 16], logratio=0.25, device="cpu")
    x.grad = None
    v = ms(x, y)
    v.backward()
    out["val_ms_log"], out["grad_ms_log"] = np_(v), np_(x.grad)
    # CE with ignore_index, masked MSE vs constant (LSGAN)
    logits = torch.from_numpy(rs.standard_normal((B * T, 14)).astype(np.float32)).requires_grad_(True)
    tgt = rs.randint(0, 14, size=(B, T))
    tgt[~np_(mask)[:, :, 0]] = -100
    tgt = torch.from_numpy(tgt.reshape(-1))
    ce = torch.nn.CrossEntropyLoss(ignore_index=-100)(logits, tgt)
    ce.backward()
    out["ce_logits"], out["ce_target"], out["ce_val"], out["ce_grad"] = np_(logits), np_(tgt), np_(ce), np_(logits.grad)
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses.npz", len(out))


# ----------------------------------------------------------------------------
def gen_stft_layer():
    import wave

    import joblib

    out = {}
    with wave.open(os.path.join(REF, "test/data/SF1_10001.wav"), "rb") as w:
        fs = w.getframerate()
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    x = (pcm.astype(np.float32) / 32768.0)[: fs]  # first second keeps the fixture small
    out["fs"] = np.array(fs)
    out["wav"] = x
    xt = torch.from_numpy(x)[None]
    for center in [False, True]:
        layer = STFTLayer(fs=fs, hop_size=128, fft_size=1024, win_length=1024, window="hann", center=center)
        s = layer(xt)
        amp = torch.sqrt(s[..., 0] ** 2 + s[..., 1] ** 2)
        out[f"amp_center{int(center)}"] = np_(amp)[:, ::8].astype(np.float32)  # every 8th frame
    scaler = joblib.load(os.path.join(REF, "test/data/scaler.pkl"))
    sl = MLFBScalerLayer(scaler["mlfb"])
    z = torch.from_numpy(np.random.RandomState(3).standard_normal((2, 9, 80)).astype(np.float32))
    out["scaler_in"], out["scaler_out"] = np_(z), np_(sl(z))
    out["scaler_mean"], out["scaler_var"] = scaler["mlfb"].mean_.astype(np.float64), scaler["mlfb"].var_.astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "stft_layer.npz"), **out)
    print("stft_layer.npz", {k: np.asarray(v).shape for k, v in out.items()})


# ----------------------------------------------------------------------------
def load_conf(**over):
    with open(os.path.join(REF, "egs/vaevc/template/conf/default.yml")) as fp:
        conf = yaml.safe_load(fp)
    for k, v in over.items():
        if isinstance(v, dict) and k in conf:
            conf[k].update(v)
        else:
            conf[k] = v
    return conf


def fill(models, seed=4321):
    for i, m in enumerate(sorted(models)):
        sd = models[m].state_dict()
        vals = deterministic_state({k: tuple(v.shape) for k, v in sd.items()}, seed + i)
        # the on-the-fly feature layer (use_raw) holds constants - mel basis, scaler statistics - not weights
        models[m].load_state_dict({k: (sd[k] if k.startswith("preprocess_layer.") else torch.from_numpy(vals[k])) for k in sd})


class _MlfbScaler:
    """What MLFBScalerLayer reads from a fitted sklearn StandardScaler (crank/net/module/mlfb.py:116-131)."""

    def __init__(self, dim=80):
        self.mean_ = (-3.0 + 0.02 * np.arange(dim)).astype(np.float64)
        self.var_ = (0.4 + 0.01 * np.arange(dim)).astype(np.float64)


def summarize_state(models):
    out = {}
    for m in sorted(models):
        for k, v in models[m].state_dict().items():
            a = np_(v).astype(np.float64).reshape(-1)
            out[f"post/{m}/{k}"] = np.array([a.sum(), np.abs(a).sum(), a[0], a[-1], a[a.size // 2]])
    return out


MCEP = {"input_feat_type": "mcep", "output_feat_type": "mcep", "input_size": 34, "output_size": 34, "use_mcep_0th": False,
        "ignore_scaler": ["mcep"]}


def run_step(trainer_type, tag, conf_over, B=2, T=96, n_spkrs=2, seed=77, steps=1, pyseed=1234, full_length=False):
    random.seed(pyseed)
    np.random.seed(pyseed)
    torch.manual_seed(pyseed)
    conf_over = dict(conf_over)
    clip = conf_over.pop("_clip", None)
    conf = load_conf(trainer_type=trainer_type, batch_size=B, batch_len=T, **conf_over)
    if clip is not None:
        for m in conf["optim"]:
            conf["optim"][m]["clip_grad_norm"] = clip
    scaler = {"mlfb": _MlfbScaler(conf["feature"]["mlfb_dim"])} if conf["use_raw"] and conf["use_preprocessed_scaler"] else None
    models = get_model(conf, spkr_size=n_spkrs, device="cpu", scaler=scaler)
    fill(models)
    for m in models.values():
        m.train()
    optimizer = get_optimizer(conf, models)
    criterion = get_criterion(conf, device="cpu")
    scheduler = get_scheduler(conf, optimizer)
    spkrs = {f"spk{i}": i for i in range(n_spkrs)}
    writer = {"train": _Dummy(), "dev": _Dummy()}
    trainer = TrainerWrapper(
        conf["trainer_type"], model=models, optimizer=optimizer, criterion=criterion,
        dataloader={"spkrs": spkrs}, writer=writer, expdir="/tmp/golden_exp", conf=conf,
        feat_conf=conf["feature"], scheduler=scheduler, scaler=scaler, resume=0, device="cpu", n_jobs=1,
    )
    trainer.tqdm.close()
    out = {}
    dim = conf["input_size"]
    for s in range(steps):
        batch = make_batch(B, T, n_spkrs, in_dim=dim, seed=seed + s, full_length=full_length, use_raw=conf["use_raw"],
                           fftl=conf["feature"]["fftl"], hop_size=conf["feature"]["hop_size"])
        # the reference enters GAN / cycle phases from trainer.steps
        trainer.steps = conf_over.get("_force_steps", 1)
        trainer.check_custom_start()
        vals = trainer.train(batch, phase="train")
        for k, v in vals.items():
            out[f"loss{s}/{k}"] = np.array(v, dtype=np.float64)
    # a forward after the update(s): decoded features and code indices
    with torch.no_grad():
        batch = make_batch(B, T, n_spkrs, in_dim=dim, seed=seed, full_length=full_length, use_raw=conf["use_raw"],
                           fftl=conf["feature"]["fftl"], hop_size=conf["feature"]["hop_size"])
        enc_h = trainer._get_enc_h(batch)
        dec_h, spkrvec = trainer._get_dec_h(batch)
        o = models["G"].forward(batch["raw"] if conf["use_raw"] else batch["in_feats"], enc_h, dec_h, spkrvec=spkrvec, use_ema=False)
        out["post_decoded"] = np_(o["decoded"])
        out["post_qidx0"], out["post_qidx1"] = np_(o["qidx"][0]), np_(o["qidx"][1])
        out["dec_h"] = np_(dec_h)
        out["spkrvec"] = np_(spkrvec)
    if scaler is not None:
        out["mlfb_scaler_mean"], out["mlfb_scaler_var"] = scaler["mlfb"].mean_, scaler["mlfb"].var_
    out.update({k: v for k, v in summarize_state(models).items() if "preprocess_layer." not in k})
    out["meta_B_T_nspk_seed_steps"] = np.array([B, T, n_spkrs, seed, steps])
    np.savez_compressed(os.path.join(HERE, f"step_{tag}.npz"), **out)
    print(f"step_{tag}.npz", {k: float(v) for k, v in out.items() if k.startswith("loss0/")})


def gen_misc():
    out = {}
    g = GradientReversalLayer(scale=0.1)
    x = torch.from_numpy(np.random.RandomState(1).standard_normal((2, 5, 8)).astype(np.float32)).requires_grad_(True)
    y = g(x)
    (y * torch.arange(8.0)).sum().backward()
    out["grl_x"], out["grl_y"], out["grl_grad"] = np_(x), np_(y), np_(x.grad)
    # StepLR stepped with an explicit step count (basetrainer.py:84-90,239-247)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=2e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=200000, gamma=0.5)
    lrs = []
    for s in [0, 199999, 200000, 400000]:
        sch.step(s)
        lrs.append(opt.param_groups[0]["lr"])
    out["steplr_steps"], out["steplr_lr"] = np.array([0, 199999, 200000, 400000]), np.array(lrs)
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **out)
    print("misc.npz", lrs)


# ----------------------------------------------------------------------------
class _RecordingRandom:
    """Stands in for the ``random`` module inside the reference's dataset module: same draws
    (delegates to the real, seeded module), but remembers what ``choice`` returned."""

    def __init__(self):
        self.log = []

    def choice(self, seq):
        v = random.choice(seq)
        self.log.append(v)
        return v

    def __getattr__(self, k):
        return getattr(random, k)


def _synthetic_corpus(rs, spkrs, lens, dim, with_cap):
    """Raw (unscaled) features per utterance, shaped like what the reference's HDF5 files hold."""
    from pathlib import Path

    corpus, files = [], {}
    for i, flen in enumerate(lens):
        spk = spkrs[i % len(spkrs)]
        si = spkrs.index(spk)
        u = {
            "feat": (rs.standard_normal((flen, dim)) * (1.0 + 0.3 * np.arange(dim)) + 0.5 * si - 2.0).astype(np.float32),
            "lcf0": (4.6 + 0.25 * si + 0.2 * rs.standard_normal(flen)).astype(np.float32),  # read_feature adds the axis
            "uv": (rs.uniform(size=flen) < 0.7).astype(np.float32),
            "spk": si,
        }
        if with_cap:
            u["cap"] = rs.standard_normal((flen, 2)).astype(np.float32)
        path = Path(f"/nonexistent/h5/{spk}/utt{i:03d}.h5")
        corpus.append(u)
        files[str(path)] = u
    return corpus, files


def _ref_getitem(dset, idx, drop_0th):
    """BaseDataset.__getitem__ (dataset.py:58-74) through the reference's own _pre_getitem,
    _transform, _zero_padding and _post_getitem.  _middle_getitem itself cannot run under
    numpy >= 1.25 (dataset.py:111 compares an ndarray with a str inside ``if``), so its three
    remaining statements -- 0th-coefficient split (:108-110), the four mask copies (:118-125) --
    are carried out here between the reference's calls."""
    sample = dset._pre_getitem(idx, str(dset.h5list[idx]))
    sample = dset._transform(sample)
    if drop_0th:
        sample["mcep_0th"] = sample["mcep"][..., :1]
        sample["mcep"] = sample["mcep"][..., 1:]
    sample = dset._zero_padding(sample)
    for ed in ["encoder_mask", "decoder_mask", "cycle_encoder_mask", "cycle_decoder_mask"]:
        sample[ed] = np.copy(sample["mask"])
    del sample["mask"]
    return dset._post_getitem(sample)


def gen_dataset():
    """BaseDataset.__getitem__ + default collate, convert_f0, the sklearn scalers, and the
    decode-side BaseTrainer._store_features / _get_cvf0, all run from the reference's own code
    on an in-memory corpus (read_feature is pointed at a dict instead of HDF5 files)."""
    from pathlib import Path
    from types import SimpleNamespace

    from sklearn.preprocessing import StandardScaler
    from torch.utils.data.dataloader import default_collate

    from crank.net.trainer import dataset as ref_ds
    from crank.net.trainer.basetrainer import BaseTrainer

    out = {}
    rs = np.random.RandomState(2024)
    spkrs = ["SF1", "SM1", "TF1", "TM2"]
    blen = 40
    lens = [17, 40, 41, 97, 1, 39, 64, 40, 250, 33, 58, 12]
    import joblib

    pkl = joblib.load(os.path.join(REF, "test", "data", "scaler.pkl"))  # the reference's own test fixture (VCC2018, 12 speakers)
    pkl_spkrs = sorted(k for k, v in pkl.items() if isinstance(v, dict))
    for case, (ftype, dim, drop) in {"mlfb": ("mlfb", 8, False), "mcep": ("mcep", 7, True), "pkl": ("mlfb", 80, False)}.items():
        if case == "pkl":
            # features drawn around the statistics of the reference's scaler.pkl, normalised with that very object
            spkrs, scaler = pkl_spkrs, pkl
            corpus, files = _synthetic_corpus(rs, spkrs, lens, dim, with_cap=False)
            for u in corpus:
                z = rs.standard_normal(u["feat"].shape)
                u["feat"][:] = (z * pkl["mlfb"].scale_ + pkl["mlfb"].mean_).astype(np.float32)
                own = pkl[spkrs[u["spk"]]]["lcf0"]
                u["lcf0"][:] = (own.mean_[0] + np.sqrt(own.var_[0]) * rs.standard_normal(u["lcf0"].shape)).astype(np.float32)
        else:
            spkrs = ["SF1", "SM1", "TF1", "TM2"]
            corpus, files = _synthetic_corpus(rs, spkrs, lens, dim, with_cap=(ftype == "mcep"))
            # scalers fitted like crank/bin/extract_statistics.py does: one per feature, one lcf0 scaler per speaker
            scaler = {ftype: StandardScaler().fit(np.concatenate([u["feat"] for u in corpus])),
                      "lcf0": StandardScaler().fit(np.concatenate([u["lcf0"][:, None] for u in corpus]))}
            for si, spk in enumerate(spkrs):
                scaler[spk] = {"lcf0": StandardScaler().fit(np.concatenate([u["lcf0"][:, None] for u in corpus if u["spk"] == si]))}

        def read_feature(h5f, ext="mlfb", files=files, ftype=ftype):
            u = files[str(h5f)]
            data = u["feat"] if ext == ftype else u[ext]
            return data[:, np.newaxis] if data.ndim == 1 else data

        conf = {"batch_len": blen, "input_feat_type": ftype, "output_feat_type": ftype, "use_raw": False,
                "cache_dataset": False, "ignore_scaler": [], "use_mcep_0th": not drop, "spec_augment": False,
                "feature": {"fftl": 1024, "hop_size": 128}}
        scp = {"train": {"feats": {Path(f).stem: Path(f) for f in files}, "spkrs": spkrs}}
        rec = _RecordingRandom()
        ref_ds.read_feature, ref_ds.random = read_feature, rec
        dset = ref_ds.BaseDataset(conf, scp, scaler, phase="train")
        random.seed(99)
        samples, draws = [], []
        for idx in range(len(dset)):
            rec.log.clear()
            samples.append(_ref_getitem(dset, idx, drop))
            cv_name = rec.log[0]
            p = rec.log[1] if len(rec.log) > 1 else 0
            draws.append((idx, spkrs.index(cv_name), p))
            assert samples[-1]["cv_spkr_name"] == cv_name
        ref_ds.random = random
        # utterances of exactly batch_len frames come back unconverted (cv_lcf0 float64, dataset.py:243-249): cast
        # those to the dtype every other sample has so that the reference's own collate can stack them
        for s in samples:
            for k, v in s.items():
                if isinstance(v, np.ndarray) and v.dtype == np.float64:
                    s[k] = v.astype(np.float32)
        batch = default_collate(samples)
        pre = f"{case}/"
        out[pre + "draws_utt_cv_p"] = np.array(draws, dtype=np.int64)
        out[pre + "lens"] = np.array(lens, dtype=np.int64)
        out[pre + "utt_spk"] = np.array([u["spk"] for u in corpus], dtype=np.int64)
        out[pre + "feat"] = np.concatenate([u["feat"] for u in corpus])
        out[pre + "lcf0"] = np.concatenate([u["lcf0"] for u in corpus])
        out[pre + "uv"] = np.concatenate([u["uv"] for u in corpus])
        out[pre + "feat_mean"], out[pre + "feat_scale"] = scaler[ftype].mean_, scaler[ftype].scale_
        out[pre + "lcf0_mean"], out[pre + "lcf0_scale"] = scaler["lcf0"].mean_, scaler["lcf0"].scale_
        out[pre + "spk_lcf0_mean"] = np.array([scaler[s]["lcf0"].mean_[0] for s in spkrs])
        out[pre + "spk_lcf0_var"] = np.array([scaler[s]["lcf0"].var_[0] for s in spkrs])
        keys = ["in_feats", "out_feats", "lcf0", "uv", "cv_lcf0", "org_h", "cv_h", "org_h_onehot", "cv_h_onehot",
                "encoder_mask", "decoder_mask", "cycle_encoder_mask", "cycle_decoder_mask", "flen"]
        if drop:
            keys.append("mcep_0th")
        for k in keys:
            out[pre + "batch/" + k] = np_(batch[k])
        out[pre + "batch_dtypes"] = np.array([f"{k}:{batch[k].dtype}" for k in keys])

        # ---- decode side: BaseTrainer._store_features / _get_cvf0 on the same batch ----
        me = SimpleNamespace(conf=conf, scaler=scaler, device="cpu")
        decoded = torch.from_numpy(rs.standard_normal(tuple(batch["out_feats"].shape)).astype(np.float32))
        flen_cut = torch.clamp(batch["flen"], max=blen)  # what a decode batch carries (batch_len = max length there)
        b2 = dict(batch)
        b2["flen"] = flen_cut
        if ftype == "mcep":
            b2["cap"] = batch["cap"]
        target = "TM2"
        out[pre + "spkrs"] = np.array(spkrs)
        feats = BaseTrainer._store_features(me, b2, {"decoded": decoded}, target, Path("/nonexistent/out"))
        out[pre + "decoded"] = np_(decoded)
        out[pre + "target_spk"] = np.array(spkrs.index(target))
        for n, (path, f) in enumerate(feats.items()):
            for k in ["feats", "lcf0", "uv", "f0", "normed_lcf0", "normed_feat"] + (["rmcep"] if drop else []):
                out[pre + f"store/{n}/{k}"] = np.asarray(f[k])
        out[pre + "cvf0"] = np_(BaseTrainer._get_cvf0(me, batch, target))
    # ---- padding_raw (dataset.py:261-285), the reference's own function: reflect padding (incl. pads longer than the
    # signal), the unpadded branch, zero extension, the cropped branch with and without a cut ----
    fftl, hop, blen = 16, 4, 10
    cases = [(20, 6, 0), (5, 2, 0), (1, 1, 0), (45, 9, 0), (39, 8, 0), (38, 8, 0), (100, 25, 0), (100, 25, 7), (60, 14, 3),
             (50, 12, 2), (64, 10, 0)]
    for k, (n, flen, p_) in enumerate(cases):
        x = rs.standard_normal(n).astype(np.float32)  # what the fixture stores; the reference works on it in float64
        y = ref_ds.padding_raw(x.astype(np.float64), blen - flen, blen, fftl, hop, value=0.0, p=p_)
        out[f"raw/x{k}"], out[f"raw/out{k}"] = x, np.asarray(y)
        out[f"raw/args{k}"] = np.array([flen, blen, fftl, hop, p_])
    out["raw/n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
    print("dataset.npz", len(out), "arrays;", {k: out[k].shape for k in list(out)[:3]})


def gen_convert(B=4, T=200, n_spkrs=14, seed=11):
    """Conversion on GIVEN parameters by the reference's own VQVAE2 (crank/net/module/vqvae2.py:84-152 forward, eval mode,
    conversion-target speaker / converted F0 as BaseTrainer._get_dec_h(use_cvfeats=True) builds them,
    crank/net/trainer/basetrainer.py:289-308): decoded features and code indices.  The fixture bench.py's parity gates and
    tests/test_gpu_step.py::test_conversion_matches_reference_golden compare against - no oracle needed on the GPU box."""
    torch.manual_seed(1234)
    conf = load_conf()
    models = get_model(conf, spkr_size=n_spkrs, device="cpu", scaler=None)
    fill({"G": models["G"]})  # (the generator alone, seed 4321: what tests.helpers.fill_models({"G": net}) loads)
    G = models["G"].eval()
    batch = make_batch(B, T, n_spkrs, in_dim=conf["input_size"], seed=seed)
    dec_h = torch.cat([batch["cv_lcf0"], batch["uv"]], -1)
    h = batch["cv_h"].clone()
    h[:, :] = h[:, 0:1]
    with torch.no_grad():
        o = G.forward(batch["in_feats"], None, dec_h, spkrvec=h, use_ema=False)
    out = {"decoded": np_(o["decoded"]), "qidx0": np_(o["qidx"][0]), "qidx1": np_(o["qidx"][1]),
           "meta_B_T_nspk_seed": np.array([B, T, n_spkrs, seed])}
    np.savez_compressed(os.path.join(HERE, "convert_vqvae.npz"), **out)
    print("convert_vqvae.npz", out["decoded"].shape, float(np.abs(out["decoded"]).max()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["quantizer", "quantizer_full", "losses", "stft", "misc", "dataset", "steps", "convert"]
    if "convert" in which:
        gen_convert()
    if "quantizer" in which:
        gen_quantizer()
    if "quantizer_full" in which:
        gen_quantizer_full()
    if "losses" in which:
        gen_losses()
    if "stft" in which:
        gen_stft_layer()
    if "misc" in which:
        gen_misc()
    if "dataset" in which:
        gen_dataset()
    if "steps" in which:
        nodrop = {"discriminator_dropout": 0.0}
        run_step("vqvae", "vqvae", {}, steps=2)
        run_step("vqvae", "vqvae_cycle", {"use_cyclic_training": True, "n_steps_cycle_start": 0})
        run_step("lsgan", "lsgan", dict(nodrop, n_steps_gan_start=0))
        run_step("cyclegan", "cyclegan", dict(nodrop, n_steps_gan_start=0, use_cyclic_training=True, n_steps_cycle_start=0))
        run_step("stargan", "stargan", dict(nodrop, n_steps_gan_start=0, use_cyclic_training=True, n_steps_cycle_start=0))
    if "branches" in which or "steps" in which:
        # configuration branches of the reference trainers, one step each (tests/helpers.py STEP_CASES):
        # G.forward on raw audio (test/test_vqvae.py:34-66), ACGAN head (trainer_lsgan.py:172-181), F0-conditioned
        # encoder (basetrainer.py:253-258), causal stacks + causal_size (trainer_vqvae.py:171-176), gradient
        # clipping (trainer_vqvae.py:203-206), dictionary loss without EMA (trainer_vqvae.py:233-237)
        nodrop = {"discriminator_dropout": 0.0}
        run_step("vqvae", "vqvae_raw", {"use_raw": True, "use_preprocessed_scaler": True})
        run_step("lsgan", "lsgan_acgan", dict(nodrop, n_steps_gan_start=0, acgan_flag=True))
        run_step("vqvae", "vqvae_encf0", {"encoder_f0": True}, steps=2)
        run_step("vqvae", "vqvae_causal", {"causal": True, "causal_size": 4}, T=160)
        run_step("vqvae", "vqvae_clip", {"_clip": 0.5}, steps=2)
        run_step("vqvae", "vqvae_noema", {"ema_flag": False}, steps=2)
        # BASELINE configs[4]: stargan on 34-dim mel-cepstra (egs/vaevc/template/conf/mcep_vqvae_22050.yml:17-25;
        # dataset.py:108-110 drops the 0th coefficient), 12 speakers (VCC2018), D input 34 + 1 + 32 = 67 -> 1
        # (crank/bin/train.py:94-118), update_D conditioned on the conversion target (trainer_stargan.py:82-118)
        run_step("stargan", "stargan_mcep", dict(MCEP, **nodrop, n_steps_gan_start=0, use_cyclic_training=True,
                                                 n_steps_cycle_start=0), B=3, T=128, n_spkrs=12)
