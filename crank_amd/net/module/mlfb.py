"""Waveform -> log-mel: the on-the-fly layer (crank/net/module/mlfb.py:134-171, used when the
recipe sets ``use_raw``; ``center=False``) and the offline extraction of the recipe's stage 2
(``logmelfilterbank``: crank/feature/feature.py:126-145 calls the third-party
``parallel_wavegan.bin.preprocess.logmelfilterbank``; ``center=True``,
reflect padding).  STFT (LDS FFT), magnitude, mel projection, log10 and the optional
standardisation run in one HIP kernel (mlfb_kernels.hip).

Only the fixed-window variants are implemented ("hann", "hamming", ... =
``torch.<name>_window``); the trainable "param" / "conv" window variants of the
reference's STFTLayer (mlfb.py:64-90) are rejected.
"""
import numpy as np
import torch

from ... import ops


def slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """Mel filterbank with librosa.filters.mel's defaults (Slaney scale: linear below
    1 kHz at 200/3 Hz per mel, log above with step ln(6.4)/27; triangular filters on
    linspace(0, sr/2, 1+n_fft/2); area normalisation 2/(f[i+2]-f[i])), restated from
    the published definition.  Returns (n_mels, n_bins) f32."""
    f_sp = 200.0 / 3
    brk_hz = 1000.0
    brk_mel = brk_hz / f_sp
    step = np.log(6.4) / 27.0

    def to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= brk_hz, brk_mel + np.log(np.maximum(f, 1e-10) / brk_hz) / step, f / f_sp)

    def to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= brk_mel, brk_hz * np.exp(step * (m - brk_mel)), f_sp * m)

    bins = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    edges = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    rel = edges[:, None] - bins[None, :]
    fb = np.zeros((n_mels, bins.size))
    for i in range(n_mels):
        fb[i] = np.maximum(0.0, np.minimum(-rel[i] / width[i], rel[i + 2] / width[i + 1]))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb.astype(np.float32)


class LogMelFilterBankLayer(torch.nn.Module):
    def __init__(self, fs=22050, hop_size=256, fft_size=1024, win_length=None, window="hann", center=True,
                 pad_mode="reflect", n_mels=80, fmin=None, fmax=None, scaler=None, eps=1.0e-10, device="cuda"):
        super().__init__()
        if window in ("param", "conv"):
            raise NotImplementedError("trainable STFT windows (mlfb.py:64-90) are not implemented")
        if center and pad_mode != "reflect":
            raise NotImplementedError("center=True is implemented for pad_mode='reflect' only")
        self.center = bool(center)
        self.hop_size, self.fft_size = hop_size, fft_size
        self.win_length = fft_size if win_length is None else win_length
        self.eps = eps
        fmin = 0 if fmin is None else fmin
        fmax = fs / 2 if fmax is None else fmax
        basis = slaney_mel_basis(fs, fft_size, n_mels, fmin, fmax)  # (n_mels, n_bins)
        self.mel_basis = torch.from_numpy(np.ascontiguousarray(basis.T)).to(device)  # (n_bins, n_mels)
        self.window = getattr(torch, f"{window}_window")(self.win_length, dtype=torch.float32, device=device)
        self.mean = self.std = None
        if scaler is not None:  # MLFBScalerLayer, mlfb.py:116-131
            self.mean = torch.from_numpy(np.asarray(scaler.mean_)).float().to(device)
            self.std = torch.from_numpy(np.asarray(scaler.var_)).float().sqrt().to(device)

    def forward(self, x):
        """x: (B, n_samples) -> (B, T, n_mels); T = 1 + (n_samples - fft_size)//hop, or 1 + n_samples//hop
        when centred (torch.stft's frame count over the padded signal)."""
        if self.center:
            if x.shape[1] <= self.fft_size // 2:
                raise ValueError(f"reflect padding of {self.fft_size // 2} samples needs a longer signal than {x.shape[1]}")
            T = 1 + x.shape[1] // self.hop_size
        else:
            T = 1 + (x.shape[1] - self.fft_size) // self.hop_size
        return ops.logmel(x, T, self.fft_size, self.hop_size, self.win_length, self.window, self.mel_basis, self.eps,
                          self.mean, self.std, center=self.center)


_LAYERS = {}


def logmelfilterbank(audio, sampling_rate, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80,
                     fmin=None, fmax=None, eps=1e-10, device="cuda"):
    """Offline log-mel of one utterance with the signature of the function the reference's feature
    extraction calls (crank/feature/feature.py:134-145): audio (n_samples,) -> (1 + n_samples // hop, num_mels)
    float32 ndarray.  The waveform crosses PCIe once; everything else happens in one kernel."""
    key = (sampling_rate, fft_size, hop_size, win_length, window, num_mels, fmin, fmax, eps, str(device))
    if key not in _LAYERS:
        _LAYERS[key] = LogMelFilterBankLayer(fs=sampling_rate, hop_size=hop_size, fft_size=fft_size, win_length=win_length,
                                             window=window, center=True, n_mels=num_mels, fmin=fmin, fmax=fmax, eps=eps,
                                             device=device)
    x = torch.as_tensor(np.asarray(audio, dtype=np.float32), device=device)[None]
    return _LAYERS[key](x)[0].cpu().numpy()
