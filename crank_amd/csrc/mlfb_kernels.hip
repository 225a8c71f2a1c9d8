// On-the-fly waveform -> log-mel front end, used only when the recipe
// sets use_raw: crank/net/module/mlfb.py:134-171 (LogMelFilterBankLayer =
// STFTLayer(center=False) -> sqrt(re^2+im^2) -> MLFBLayer matmul/clamp/log10 ->
// MLFBScalerLayer), wrapped by raw_preprocessing at crank/net/module/vqvae2.py:23-35.
//
// One 256-thread workgroup owns a run of frames of one utterance.  Per frame: the
// windowed samples are staged into LDS in bit-reversed order, a radix-2 complex FFT
// runs in LDS (twiddles from an LDS table built once per workgroup), the one-sided
// magnitudes stay in LDS and the mel projection is a dense matvec against the
// L2-resident filterbank (only each filter's nonzero run of bins).  HBM traffic is the raw samples once (hop/n_fft overlap is
// served by L2) plus n_mels floats per frame.
#include "common.h"

#define MLFB_MAX_FFT 2048

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ raw, int ld_raw, int n_samples, int T,
                                                     int n_fft, int log2n, int hop, int win, const float* __restrict__ window,
                                                     const float* __restrict__ mel, int n_mels, float eps,
                                                     const float* __restrict__ mean, const float* __restrict__ stdv,
                                                     float* __restrict__ out, int ldo, int frames_per_block, int center) {
  __shared__ float re[MLFB_MAX_FFT], im[MLFB_MAX_FFT];
  __shared__ float twr[MLFB_MAX_FFT / 2], twi[MLFB_MAX_FFT / 2];
  __shared__ float wnd[MLFB_MAX_FFT];
  __shared__ int mel_lo[256], mel_hi[256];  // nonzero bin range of each (triangular) mel filter
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int lpad = (n_fft - win) / 2;
  for (int k = tid; k < n_fft / 2; k += 256) {
    float s, c;
    sincospif(-2.0f * (float)k / (float)n_fft, &s, &c);
    twr[k] = c; twi[k] = s;
  }
  for (int j = tid; j < n_fft; j += 256) wnd[j] = (j >= lpad && j < lpad + win) ? window[j - lpad] : 0.f;
  const int n_bins = n_fft / 2 + 1;
  // A mel filter is a triangle over a short run of bins (80 filters over 513 bins: 4 - 40 bins each); the dense matvec read
  // the whole 164 KB basis from L2 per frame.  The products with the zero weights outside [lo, hi) add exact zeros to a
  // non-negative sum, so skipping them leaves every output bit unchanged.
  for (int m = tid; m < n_mels; m += 256) {
    int lo = n_bins, hi = 0;
    for (int k = 0; k < n_bins; k++)
      if (mel[(long)k * n_mels + m] != 0.f) { lo = min(lo, k); hi = k + 1; }
    mel_lo[m] = lo; mel_hi[m] = hi;
  }
  const int t_begin = blockIdx.x * frames_per_block;
  const int t_end = min(T, t_begin + frames_per_block);
  for (int t = t_begin; t < t_end; t++) {
    __syncthreads();
    const float* src = raw + (long)b * ld_raw;
    for (int j = tid; j < n_fft; j += 256) {
      const int r = (int)(__brev((unsigned)j) >> (32 - log2n));
      long pos = (long)t * hop + j;
      if (center) {  // torch.stft / librosa center=True, pad_mode="reflect": n_fft/2 mirrored samples either side, edge not repeated
        pos -= n_fft / 2;
        if (pos < 0) pos = -pos;
        if (pos >= n_samples) pos = 2L * (n_samples - 1) - pos;
      }
      re[r] = (pos >= 0 && pos < n_samples) ? src[pos] * wnd[j] : 0.f;
      im[r] = 0.f;
    }
    __syncthreads();
    for (int st = 0; st < log2n; st++) {
      const int half = 1 << st;
      const int tstride = (n_fft >> 1) >> st;
      for (int bf = tid; bf < n_fft / 2; bf += 256) {
        const int grp = bf >> st, pos = bf & (half - 1);
        const int i0 = (grp << (st + 1)) + pos, i1 = i0 + half;
        const float wr = twr[pos * tstride], wi = twi[pos * tstride];
        const float xr = re[i1] * wr - im[i1] * wi;
        const float xi = re[i1] * wi + im[i1] * wr;
        const float ar = re[i0], ai = im[i0];
        re[i0] = ar + xr; im[i0] = ai + xi;
        re[i1] = ar - xr; im[i1] = ai - xi;
      }
      __syncthreads();
    }
    for (int k = tid; k < n_bins; k += 256) re[k] = sqrtf(re[k] * re[k] + im[k] * im[k]);
    __syncthreads();
    for (int m = tid; m < n_mels; m += 256) {
      float acc = 0.f;
      const int k_hi = mel_hi[m];
      for (int k = mel_lo[m]; k < k_hi; k++) acc += re[k] * mel[(long)k * n_mels + m];
      float v = log10f(fmaxf(acc, eps));
      if (mean) v = (v - mean[m]) / stdv[m];
      out[((long)b * T + t) * ldo + m] = v;
    }
  }
}

extern "C" int crk_logmel_fwd(const float* raw, int ld_raw, int B, int n_samples, int T, int n_fft, int hop,
                              int win_length, const float* window, const float* mel_basis, int n_mels, float eps,
                              const float* mean, const float* stdv, float* out, int ldo, int center, void* stream) {
  if (!raw || !window || !mel_basis || !out || n_fft > MLFB_MAX_FFT || (n_fft & (n_fft - 1)) || win_length > n_fft ||
      n_mels > 256)
    return CRK_ERR_ARG;
  if (center && n_samples <= n_fft / 2) return CRK_ERR_ARG;  // reflect padding needs pad < length, like torch.stft
  int log2n = 0;
  while ((1 << log2n) < n_fft) log2n++;
  const int fpb = 8;
  dim3 grid((T + fpb - 1) / fpb, B), block(256);
  hipLaunchKernelGGL(logmel_kernel, grid, block, 0, (hipStream_t)stream, raw, ld_raw, n_samples, T, n_fft, log2n, hop,
                     win_length, window, mel_basis, n_mels, eps, mean, stdv, out, ldo, fpb, center);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

extern "C" const char* crk_version(void) { return "crank_hip 0.1 (gfx950)"; }
