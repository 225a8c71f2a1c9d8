"""On-the-fly log-mel front end (logmel_kernel; crank/net/module/mlfb.py:165-171) at the use_raw benchmark
shape: B = 64 utterances x 65 023 samples (= 500 frames of hop 128 + one FFT window, center=False) -> (64, 500, 80).
    python tools/prof_logmel.py            HIP-event time per call, GB/s against the algorithmic bytes
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python tools/prof_logmel.py   (kernel-only durations)
Algorithmic bytes per call: raw samples once (B x n_samples x 4) + B x T x 80 x 4 written (the hop / n_fft overlap of the
windows and the 164 KB filterbank are served by L2)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from crank_amd.net.module.mlfb import LogMelFilterBankLayer  # noqa: E402

B, hop, nfft, T = 64, 128, 1024, 500
ns = (T - 1) * hop + nfft - 1 + 128  # 65 023: the longest waveform that still gives T frames
layer = LogMelFilterBankLayer(fs=22050, hop_size=hop, fft_size=nfft, win_length=nfft, window="hann", center=False,
                              n_mels=80, fmin=80, fmax=7600, device="cuda")
g = torch.Generator().manual_seed(0)
x = (0.1 * torch.randn(B, ns, generator=g)).cuda()
for _ in range(3):
    y = layer(x)
torch.cuda.synchronize()
assert y.shape == (B, T, 80), y.shape
import hashlib  # noqa: E402

print("output sha1 (bit-identity across library builds):", hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16])
iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(iters):
    layer(x)
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) / iters * 1e3
by = B * ns * 4 + B * T * 80 * 4
print(f"logmel_kernel B={B} x {ns} samples -> {T} frames x 80: {us:.1f} us per call (events, incl. launch); "
      f"algorithmic {by / 1e6:.1f} MB -> {by / us / 1e3:.1f} GB/s = {by / us / 1e3 / 8000 * 100:.2f} % of the 8 TB/s HBM peak; "
      f"{B * T / us:.2f} frames/us")
