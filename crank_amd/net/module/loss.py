"""Loss modules with the reference's call surface (crank/net/module/loss.py) on the
HIP loss kernels: masked / causal-shifted L1 and MSE feature losses and the
multi-resolution STFT-magnitude loss along the frame axis.
"""
import torch
import torch.nn as nn

from ... import ops


class MeanLoss(nn.Module):
    """nn.L1Loss / nn.MSELoss replacement (mean reduction) for same-shaped tensors."""

    def __init__(self, mode):
        super().__init__()
        self.mode = mode

    def forward(self, x, y):
        return ops.masked_mean_loss(x, y, None, self.mode)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(ignore_index=-100) over (N,C) logits / (N,) int64 targets
    (crank/net/trainer/utils.py:26)."""

    def __init__(self, ignore_index=-100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        return ops.cross_entropy(logits, target, self.ignore_index)


class MultiSizeSTFTLoss(nn.Module):
    """crank/net/module/loss.py:88-114.  The reference passes (fft, hop, win) into
    STFTLoss(fft, win, hop) and stft() passes (fft, win, hop) into
    torch.stft(n_fft, hop_length, win_length); the net effect (verified
    by spying on torch.stft) is hop_length = cfg win_sizes[i], win_length =
    cfg hop_sizes[i], window = hann(cfg hop_sizes[i]).  Reproduced, not fixed."""

    def __init__(self, fft_sizes=[32, 128, 256], win_sizes=[20, 80, 160], hop_sizes=[10, 20, 30], logratio=0.0,
                 device="cuda"):
        super().__init__()
        self.logratio = float(logratio)
        self.resolutions = [(f, w, h) for f, w, h in zip(fft_sizes, win_sizes, hop_sizes)]  # (n_fft, hop_eff, win_eff)
        self.windows = [torch.hann_window(h, dtype=torch.float32, device=device) for (_, _, h) in self.resolutions]

    def forward(self, x, y):
        return ops.stft_loss(x, y, self.resolutions, self.windows, self.logratio)


class STFTLoss(nn.Module):
    """crank/net/module/loss.py:63-85 constructed directly: the two swaps cancel, so
    torch.stft sees (fft_size, hop_size, win_size) as named."""

    def __init__(self, fft_size=32, win_size=20, hop_size=10, logratio=0.0, device="cuda"):
        super().__init__()
        self.logratio = float(logratio)
        self.resolutions = [(fft_size, hop_size, win_size)]
        self.windows = [torch.hann_window(win_size, dtype=torch.float32, device=device)]

    def forward(self, x, y):
        return ops.stft_loss(x, y, self.resolutions, self.windows, self.logratio)


class CustomFeatureLoss(nn.Module):
    """crank/net/module/loss.py:18-47: optional causal shift of x / y / mask, optional
    frame mask, then L1 / MSE mean or the STFT loss (which ignores the mask)."""

    def __init__(self, loss_type="l1", causal=False, stft_params={}, device="cuda"):
        super().__init__()
        self.loss_type = loss_type
        self.causal = causal
        if loss_type == "stft":
            self.loss_func = MultiSizeSTFTLoss(**stft_params, device=device)
        elif loss_type not in ("l1", "mse"):
            raise ValueError(f"unknown loss_type {loss_type}")

    def both(self, x, y, mask=None, causal_size=0):
        """(L1, MSE) of the same arguments in one pass - what two calls of an "l1" and an "mse" instance return."""
        x, y, mask = self._shift(x, y, mask, causal_size)
        return ops.masked_both_loss(x, y, mask)

    def recon(self, x, y, mask, causal_size, stft):
        """(L1, MSE, STFT loss) of the same arguments: what "l1", "mse" and the "stft" instance `stft` return, with one
        backward launch for all of them.  None where the STFT configuration is outside the fused kernels' range."""
        f = getattr(stft, "loss_func", None)
        if not isinstance(f, MultiSizeSTFTLoss) or len(f.resolutions) > 4 or any(r[2] > 64 for r in f.resolutions):
            return None
        x, y, mask = self._shift(x, y, mask, causal_size)
        if x.dim() != 3:
            return None
        return ops.recon_loss(x, y, mask, f.resolutions, f.windows, f.logratio)

    def _shift(self, x, y, mask, causal_size):
        if self.causal:
            if causal_size > 0:
                x = x[:, causal_size:]
                y = y[:, :-causal_size]
                mask = mask[:, causal_size:] if mask is not None else None
            elif causal_size < 0:
                cs = -causal_size
                y = y[:, cs:]
                x = x[:, :-cs]
                mask = mask[:, :-cs] if mask is not None else None
        return x, y, mask

    def forward(self, x, y, mask=None, causal_size=0):
        x, y, mask = self._shift(x, y, mask, causal_size)
        if self.loss_type == "stft":
            return self.loss_func(x, y)
        return ops.masked_mean_loss(x, y, mask, self.loss_type)
