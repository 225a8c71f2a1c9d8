"""Oracle (TEST INFRASTRUCTURE): PyTorch-CPU fp32 restatement of crank's own step
arithmetic.  Each function/class cites the reference lines it follows; the golden
fixtures under ``tests/golden`` (made by ``tests/golden/make_golden.py`` from the
imported reference classes) pin it.

Never imported by the product package.
"""

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pwg


# ----------------------------------------------------------------------------
# VQ codebook (crank/net/module/vqvae2.py:286-347)
# ----------------------------------------------------------------------------
def vq_nearest(x_flat, codebook):
    """vqvae2.py:338-347.  dist = sum(W^2) - 2 X W^T + sum(X^2), fp32, argmin
    along the code axis (first index wins ties)."""
    w2 = (codebook * codebook).sum(dim=1)
    x2 = (x_flat * x_flat).sum(dim=1, keepdim=True)
    dist = w2 - 2 * (x_flat @ codebook.t()) + x2
    return dist.argmin(dim=1)


def vq_ema_update(x_btd, idx, ema_size, ema_w, decay=0.99, eps=1e-5):
    """vqvae2.py:315-330.  Every frame counts (no mask); the Laplace-smoothed
    cluster size is what gets stored; returns (ema_size, ema_w, codebook)."""
    K = ema_size.numel()
    D = x_btd.size(-1)
    counts = torch.bincount(idx.reshape(-1), minlength=K).to(x_btd.dtype)
    onehot = F.one_hot(idx, K).to(x_btd.dtype)  # (B,T,K)
    # reference sums per-utterance x^T.onehot products over the batch
    embed_sum = torch.matmul(x_btd.transpose(1, 2), onehot).sum(dim=0)  # (D,K)
    ema_size = decay * ema_size + (1 - decay) * counts
    ema_w = decay * ema_w + (1 - decay) * embed_sum
    n = ema_size.sum()
    ema_size = (ema_size + eps) / (n + K * eps) * n
    codebook = (ema_w / ema_size.unsqueeze(0)).t().contiguous()
    assert codebook.shape == (K, D)
    return ema_size, ema_w, codebook


class OracleQuantizer(nn.Module):
    """vqvae2.py:286-336 with bdt_flag semantics ((B,D,T) in when bdt_flag)."""

    def __init__(self, emb_dim, emb_size, decay=0.99, eps=1e-5, ema_flag=False, bdt_flag=False):
        super().__init__()
        self.emb_dim, self.emb_size = emb_dim, emb_size
        self.ema_flag, self.bdt_flag = ema_flag, bdt_flag
        self.embedding = nn.Embedding(emb_size, emb_dim)
        self.embedding.weight.data.uniform_(-1.0 / emb_size, 1.0 / emb_size)
        if ema_flag:
            self.decay, self.eps = decay, eps
            w0 = torch.randn(emb_dim, emb_size)
            self.register_buffer("ema_size", torch.zeros(emb_size))
            self.register_buffer("ema_w", w0.clone())

    def forward(self, x, use_ema=True):
        if self.bdt_flag:
            x = x.transpose(1, 2)
        B, T, D = x.shape
        idx = vq_nearest(x.reshape(-1, D), self.embedding.weight).view(B, T)
        # lookup goes through the OLD codebook (vqvae2.py:311-313); one_hot @ W
        e = torch.matmul(F.one_hot(idx, self.emb_size).float(), self.embedding.weight)
        if self.training and self.ema_flag and use_ema:
            s, w, cb = vq_ema_update(x, idx, self.ema_size, self.ema_w.data, self.decay, self.eps)
            self.ema_size = s
            self.ema_w.data = w
            self.embedding.weight.data.copy_(cb)
        qx = x + (e - x).detach()  # straight-through, vqvae2.py:333
        if self.bdt_flag:
            qx = qx.transpose(1, 2)
        return e, qx, idx


# ----------------------------------------------------------------------------
# Hierarchical VQ-VAE (crank/net/module/vqvae2.py:38-283)
# ----------------------------------------------------------------------------
class OracleVQVAE2(nn.Module):
    def __init__(self, conf, spkr_size=0, scaler=None):
        super().__init__()
        self.conf, self.spkr_size = conf, spkr_size
        self.encoder_receptive_size = 0
        self.decoder_receptive_size = 0
        self.encoders = nn.ModuleList()
        self.decoders = nn.ModuleList()
        self.quantizers = nn.ModuleList()
        nst = conf["n_vq_stacks"]
        for n in range(nst):
            if n == 0:  # vqvae2.py:218-231
                e_in, e_out = conf["input_size"], conf["emb_dim"][0]
                e_aux = 2 if conf["encoder_f0"] else 0
                d_in = sum(conf["emb_dim"][i] for i in range(nst))
                d_out = conf["output_size"]
                d_aux = 2 if conf["decoder_f0"] else 0
                d_aux += conf["spkr_embedding_size"] if conf["use_spkr_embedding"] else spkr_size
            else:  # vqvae2.py:232-238
                e_in, e_out, e_aux = conf["emb_dim"][n - 1], conf["emb_dim"][n], 0
                d_in, d_out, d_aux = conf["emb_dim"][n], conf["emb_dim"][n - 1], 0
            common = dict(
                kernel_size=conf["kernel_size"][n],
                layers=conf["n_layers"][n] * conf["n_layers_stacks"][n],
                stacks=conf["n_layers_stacks"][n],
                residual_channels=64,
                gate_channels=128,
                skip_channels=64,
                aux_context_window=0,
                dropout=0.0,
                bias=True,
                use_weight_norm=True,
                use_causal_conv=conf["causal"],
                upsample_conditional_features=False,
            )
            self.encoders.append(
                pwg.ParallelWaveGANGenerator(in_channels=e_in, out_channels=e_out, aux_channels=e_aux, **common)
            )
            self.decoders.append(
                pwg.ParallelWaveGANGenerator(in_channels=d_in, out_channels=d_out, aux_channels=d_aux, **common)
            )
            self.encoder_receptive_size += self.encoders[-1].receptive_field_size
            self.decoder_receptive_size += self.decoders[-1].receptive_field_size
            self.quantizers.append(
                OracleQuantizer(conf["emb_dim"][n], conf["emb_size"][n], ema_flag=conf["ema_flag"], bdt_flag=True)
            )
        if conf["use_spkr_embedding"]:
            self.spkr_embedding = nn.Embedding(spkr_size, conf["spkr_embedding_size"])
        if conf["use_raw"]:
            ms = scaler["mlfb"] if conf["use_preprocessed_scaler"] else None
            f = conf["feature"]
            self.preprocess_layer = OracleLogMel(
                fs=f["fs"], hop_size=f["hop_size"], fft_size=f["fftl"], win_length=f["win_length"],
                window=conf["raw_window_type"], center=False, n_mels=f["mlfb_dim"],
                fmin=f["fmin"], fmax=f["fmax"], scaler=ms,
            )
        elif conf["use_sinc_conv"]:
            raise NotImplementedError("use_sinc_conv is a dead branch in the reference (SURVEY Q10)")

    # vqvae2.py:154-158
    def _cond(self, dec_h, spkrvec):
        if spkrvec is not None:
            emb = self.spkr_embedding(spkrvec)
            dec_h = emb if dec_h is None else torch.cat([dec_h, emb], dim=-1)
        return dec_h

    @staticmethod
    def _t(h):
        return h.transpose(1, 2) if h is not None else None

    def _pre(self, x):
        if self.conf["use_raw"]:
            return self.preprocess_layer(x)
        return x

    def encode(self, x, enc_h=None):  # vqvae2.py:160-169
        out = []
        for n in range(self.conf["n_vq_stacks"]):
            cur = self.encoders[n](x, c=enc_h) if n == 0 else self.encoders[n](cur, c=None)
            out.append(cur)
        return out

    def decode(self, enc, dec_h, use_ema=True, detach=False):  # vqvae2.py:171-190
        dec = 0
        emb_idxs, qxs, qidxs = [], [], []
        for n in reversed(range(self.conf["n_vq_stacks"])):
            enc[n] = enc[n] + dec  # mutates the caller's list (quirk Q6)
            e, qx, qi = self.quantizers[n](enc[n], use_ema=use_ema)
            if detach:
                qx = qx.detach()
            emb_idxs.append(e)
            qxs.append(qx)
            qidxs.append(qi)
            if n != 0:
                dec = self.decoders[n](qx, c=None)
            else:
                dec = self.decoders[n](torch.cat(qxs, dim=1), c=dec_h)
        return enc, dec, emb_idxs, qxs, qidxs

    @staticmethod
    def make_dict(enc, dec, emb_idxs, qidxs, enc_unmod):  # vqvae2.py:197-209
        return {
            "encoded": [e.transpose(1, 2) for e in enc],
            "encoded_unmod": [e.transpose(1, 2) for e in enc_unmod] if enc_unmod is not None else None,
            "decoded": dec.transpose(1, 2),
            "emb_idx": emb_idxs[::-1],
            "qidx": qidxs[::-1],
        }

    def forward(self, x, enc_h, dec_h, spkrvec=None, use_ema=True, encoder_detach=False):
        x = self._pre(x).transpose(1, 2)
        dec_h = self._t(self._cond(dec_h, spkrvec))
        enc = self.encode(x, enc_h=self._t(enc_h))
        unmod = [e.clone() for e in enc]
        enc, dec, embs, _, qidxs = self.decode(enc, dec_h, use_ema=use_ema, detach=encoder_detach)
        return self.make_dict(enc, dec, embs, qidxs, unmod)

    def cycle_forward(self, x, org_enc_h, org_dec_h, cv_enc_h, cv_dec_h, org_spkrvec, cv_spkrvec):
        # vqvae2.py:101-152
        x = self._pre(x).transpose(1, 2)
        o_dec_h = self._t(self._cond(org_dec_h, org_spkrvec))
        c_dec_h = self._t(self._cond(cv_dec_h, cv_spkrvec))
        o_enc_h, c_enc_h = self._t(org_enc_h), self._t(cv_enc_h)
        outs = []
        for _ in range(self.conf["n_cycles"]):
            enc = self.encode(x, enc_h=o_enc_h)
            o_unmod = [e.clone() for e in enc]
            c_unmod = [e.clone() for e in enc]
            o_enc, o_dec, o_emb, _, o_q = self.decode(enc, o_dec_h)
            c_enc, c_dec, c_emb, _, c_q = self.decode(enc, c_dec_h)  # same list object (Q6)
            enc = self.encode(c_dec, enc_h=c_enc_h)
            r_unmod = [e.clone() for e in enc]
            r_enc, r_dec, r_emb, _, r_q = self.decode(enc, o_dec_h)
            outs.append(
                {
                    "org": self.make_dict(o_enc, o_dec, o_emb, o_q, o_unmod),
                    "cv": self.make_dict(c_enc, c_dec, c_emb, c_q, c_unmod),
                    "recon": self.make_dict(r_enc, r_dec, r_emb, r_q, r_unmod),
                }
            )
            x = r_dec.clone().detach()
        return outs


# ----------------------------------------------------------------------------
# Gradient reversal + speaker adversarial net (crank/net/module/spkradv.py)
# ----------------------------------------------------------------------------
class _GRL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -ctx.scale * g, None


class OracleSpeakerAdversarialNetwork(nn.Module):
    def __init__(self, conf, spkr_size=0):
        super().__init__()
        self.conf, self.spkr_size = conf, spkr_size
        self.scale = float(conf["spkradv_lambda"])
        self.classifier = pwg.ParallelWaveGANDiscriminator(  # spkradv.py:49-60
            in_channels=sum(conf["emb_dim"][: conf["n_vq_stacks"]]),
            out_channels=spkr_size,
            kernel_size=conf["spkradv_kernel_size"],
            layers=conf["n_spkradv_layers"],
            conv_channels=64,
            dilation_factor=1,
            nonlinear_activation="LeakyReLU",
            nonlinear_activation_params={"negative_slope": 0.2},
            bias=True,
            use_weight_norm=True,
        )

    def forward(self, x, detach=False):  # spkradv.py:27-33
        x = torch.cat(x, dim=-1)
        if detach:
            x = x.detach()
        x = _GRL.apply(x, self.scale).transpose(1, 2)
        return self.classifier(x).transpose(1, 2)


# ----------------------------------------------------------------------------
# Losses (crank/net/module/loss.py)
# ----------------------------------------------------------------------------
def stft_mag(x_btd, n_fft, hop_length, win_length, window):
    """loss.py:50-60 as torch.stft finally sees its arguments."""
    rows = x_btd.transpose(1, 2).reshape(-1, x_btd.size(1))
    spec = torch.stft(rows, n_fft, hop_length, win_length, window, return_complex=True)
    p = torch.clamp(spec.real ** 2 + spec.imag ** 2, min=1e-7).transpose(2, 1)
    return torch.sqrt(p)


def multi_stft_loss(x, y, fft_sizes, win_sizes, hop_sizes, logratio=0.0):
    """loss.py:88-114 including the argument shuffle: through
    MultiSizeSTFTLoss torch.stft gets hop_length = cfg win_sizes[i],
    win_length = cfg hop_sizes[i] and a hann window of cfg hop_sizes[i] taps."""
    total = 0.0
    for f, w, h in zip(fft_sizes, win_sizes, hop_sizes):
        hop_eff, win_eff = w, h
        win = torch.hann_window(win_eff, dtype=x.dtype, device=x.device)
        xm = stft_mag(x, f, hop_eff, win_eff, win)
        ym = stft_mag(y, f, hop_eff, win_eff, win)
        mag = F.l1_loss(xm, ym)
        lmag = F.l1_loss(xm.log(), ym.log())
        total = total + (1 - logratio) * mag + logratio * lmag
    return total / len(fft_sizes)


class OracleFeatureLoss(nn.Module):
    """loss.py:18-47."""

    def __init__(self, loss_type="l1", causal=False, stft_params=None):
        super().__init__()
        self.loss_type, self.causal = loss_type, causal
        self.stft_params = stft_params or {}

    def forward(self, x, y, mask=None, causal_size=0):
        if self.causal:
            if causal_size > 0:
                x, y = x[:, causal_size:], y[:, :-causal_size]
                mask = mask[:, causal_size:] if mask is not None else None
            elif causal_size < 0:
                cs = -causal_size
                y, x = y[:, cs:], x[:, :-cs]
                mask = mask[:, :-cs] if mask is not None else None
        if mask is not None:
            x, y = x.masked_select(mask), y.masked_select(mask)
        if self.loss_type == "l1":
            return F.l1_loss(x, y)
        if self.loss_type == "mse":
            return F.mse_loss(x, y)
        return multi_stft_loss(x, y, **self.stft_params)


def get_criterion(conf):
    """crank/net/trainer/utils.py:22-37."""
    return {
        "mse": nn.MSELoss(),
        "l1": nn.L1Loss(),
        "ce": nn.CrossEntropyLoss(ignore_index=-100),
        "kld": nn.KLDivLoss(reduction="mean"),
        "fmse": OracleFeatureLoss("mse", causal=conf["causal"]),
        "fl1": OracleFeatureLoss("l1", causal=conf["causal"]),
        "fstft": OracleFeatureLoss("stft", causal=conf["causal"], stft_params=conf["stft_params"]),
    }


# ----------------------------------------------------------------------------
# On-the-fly log-mel front end (crank/net/module/mlfb.py)
# ----------------------------------------------------------------------------
def slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel defaults (htk=False, norm='slaney'), restated from the
    published formula; float32 (n_mels, 1+n_fft//2)."""

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        freqs = f_sp * m
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


class OracleLogMel(nn.Module):
    """mlfb.py:134-171 for the fixed-window variants ("hann" etc.)."""

    def __init__(self, fs=22050, hop_size=256, fft_size=1024, win_length=None, window="hann",
                 center=True, pad_mode="reflect", n_mels=80, fmin=None, fmax=None, scaler=None, eps=1e-10):
        super().__init__()
        self.hop_size, self.fft_size = hop_size, fft_size
        self.win_length = fft_size if win_length is None else win_length
        self.window, self.center, self.pad_mode, self.eps = window, center, pad_mode, eps
        fmin = 0 if fmin is None else fmin
        fmax = fs / 2 if fmax is None else fmax
        basis = slaney_mel_basis(fs, fft_size, n_mels, fmin, fmax)
        self.register_buffer("mel_basis", torch.from_numpy(basis.T.copy()).float())
        if scaler is not None:
            self.register_buffer("mean", torch.from_numpy(np.asarray(scaler.mean_)).float())
            self.register_buffer("std", torch.from_numpy(np.asarray(scaler.var_)).float().sqrt())
        else:
            self.mean = None

    def stft(self, x):  # mlfb.py:92-113
        win = getattr(torch, f"{self.window}_window")(self.win_length, dtype=x.dtype, device=x.device)
        s = torch.stft(x, n_fft=self.fft_size, win_length=self.win_length, hop_length=self.hop_size,
                       window=win, center=self.center, pad_mode=self.pad_mode, return_complex=True)
        return torch.view_as_real(s).transpose(1, 2).float()

    def forward(self, x):  # mlfb.py:165-171
        s = self.stft(x)
        amp = torch.sqrt(s[..., 0] ** 2 + s[..., 1] ** 2)
        m = torch.clamp(torch.matmul(amp, self.mel_basis), min=self.eps).log10()
        if self.mean is not None:
            m = (m - self.mean) / self.std
        return m


# ----------------------------------------------------------------------------
# Model factory (crank/bin/train.py:56-131)
# ----------------------------------------------------------------------------
def get_model(conf, spkr_size=0, scaler=None):
    models = {"G": OracleVQVAE2(conf, spkr_size=spkr_size, scaler=scaler)}
    if conf["use_spkradv_training"]:
        models["SPKRADV"] = OracleSpeakerAdversarialNetwork(conf, spkr_size)
    if conf["use_spkr_classifier"]:
        models["C"] = pwg.ParallelWaveGANDiscriminator(
            in_channels=conf["input_size"], out_channels=spkr_size,
            kernel_size=conf["spkr_classifier_kernel_size"], layers=conf["n_spkr_classifier_layers"],
            conv_channels=64, dilation_factor=1, nonlinear_activation="LeakyReLU",
            nonlinear_activation_params={"negative_slope": 0.2}, bias=True, use_weight_norm=True,
        )
    if conf["trainer_type"] in ["lsgan", "cyclegan", "stargan"]:
        cin = conf["input_size"] + (1 if conf["use_D_uv"] else 0)
        if conf["use_D_spkrcode"]:
            cin += conf["spkr_embedding_size"] if conf["use_spkr_embedding"] else spkr_size
        if conf["gan_type"] != "lsgan":
            raise ValueError("only gan_type lsgan is defined (SURVEY Q10)")
        cout = 1 + (spkr_size if conf["acgan_flag"] else 0)
        if not conf["use_residual_network"]:
            raise NotImplementedError("non-residual D is broken in the reference (train.py:121)")
        models["D"] = pwg.ResidualParallelWaveGANDiscriminator(
            in_channels=cin, out_channels=cout, kernel_size=conf["discriminator_kernel_size"],
            layers=conf["n_discriminator_layers"] * conf["n_discriminator_stacks"],
            stacks=conf["n_discriminator_stacks"], dropout=conf["discriminator_dropout"],
        )
    return models


def get_optimizer(conf, model):
    """crank/net/trainer/utils.py:40-58 (radam / lamb: the restatements of oracle/optim.py - their packages are absent)."""
    from .optim import make_optimizer

    out = {}
    for m in ["G", "D", "C", "SPKRADV"]:
        if m in model:
            out[m] = make_optimizer(conf["optim"][m]["type"], model[m].parameters(), conf["optim"][m]["lr"])
    return out


def steplr_value(base_lr, steps, step_size, gamma):
    """StepLR stepped with an explicit epoch (basetrainer.py:239-247)."""
    return base_lr * gamma ** (steps // step_size)
