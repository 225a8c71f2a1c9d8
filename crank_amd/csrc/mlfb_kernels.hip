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
void conv_prof_begin(int cls, double flops, hipStream_t s);  // (conv_kernels.hip: bench.py's per-class HIP-event timing; class 8)
void conv_prof_end(int cls, hipStream_t s);
void conv_prof_bytes(int cls, double bytes);

#define MLFB_MAX_FFT 2048

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ raw, int ld_raw, int n_samples, int T,
                                                     int n_fft, int log2n, int hop, int win, const float* __restrict__ window,
                                                     const float* __restrict__ mel, int n_mels, float eps,
                                                     const float* __restrict__ mean, const float* __restrict__ stdv,
                                                     float* __restrict__ out, int ldo, int frames_per_block, int center) {
  __shared__ float re[MLFB_MAX_FFT], im[MLFB_MAX_FFT];
  __shared__ float twr[MLFB_MAX_FFT], twi[MLFB_MAX_FFT];  // stage st's twiddles at [2^st, 2^(st+1)): consecutive lanes, consecutive words
  __shared__ float wnd[MLFB_MAX_FFT];
  __shared__ int mel_lo[256], mel_hi[256];  // nonzero bin range of each (triangular) mel filter
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int lpad = (n_fft - win) / 2;
  // exp(-2 pi i k / n_fft) for k = pos * (n_fft / 2 >> st), stored per STAGE (index 2^st + pos): the lanes of a butterfly
  // instruction read consecutive words (one table indexed by k has them n_fft / 2^(st+1) words apart).  Same values, same
  // output bits; measured at the benchmark shape it is worth nothing by itself (profiles/round5_logmel_kernel.txt: the frame
  // loop is bound by its twelve barriers and two exposed global round trips per frame, not by LDS bandwidth).
  for (int i = tid + 1; i < n_fft; i += 256) {
    const int st = 31 - __clz(i), pos = i - (1 << st);
    const int k = pos * ((n_fft >> 1) >> st);
    float s, c;
    sincospif(-2.0f * (float)k / (float)n_fft, &s, &c);
    twr[i] = c; twi[i] = s;
  }
  for (int j = tid; j < n_fft; j += 256) wnd[j] = (j >= lpad && j < lpad + win) ? window[j - lpad] : 0.f;
  const int n_bins = n_fft / 2 + 1;
  // A mel filter is a triangle over a short run of bins (80 filters over 513 bins: 4 - 40 bins each); the dense matvec read
  // the whole 164 KB basis from L2 per frame.  The products with the zero weights outside [lo, hi) add exact zeros to a
  // non-negative sum, so skipping them leaves every output bit unchanged.
  for (int m = tid; m < n_mels; m += 256) { mel_lo[m] = n_bins; mel_hi[m] = 0; }
  __syncthreads();
  for (int i = tid; i < n_bins * n_mels; i += 256)  // (the whole workgroup walks the basis once, coalesced)
    if (mel[i] != 0.f) {
      const int k = i / n_mels, m = i - k * n_mels;
      atomicMin(&mel_lo[m], k);
      atomicMax(&mel_hi[m], k + 1);
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
      for (int bf = tid; bf < n_fft / 2; bf += 256) {
        const int grp = bf >> st, pos = bf & (half - 1);
        const int i0 = (grp << (st + 1)) + pos, i1 = i0 + half;
        const float wr = twr[half + pos], wi = twi[half + pos];
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
  // frames per workgroup: the per-workgroup tables (1 023 twiddles through sincospif, window, filter ranges) cost as much as
  // several frames - as few workgroups as still fill the machine four times over
  int fpb = (int)(((long long)T * B) / 1024);
  fpb = fpb < 8 ? 8 : (fpb > 64 ? 64 : fpb);
  dim3 grid((T + fpb - 1) / fpb, B), block(256);
  // algorithmic bytes: every sample once (the hop / n_fft overlap of the windows is served by L2) + n_mels floats per frame
  conv_prof_bytes(8, 4.0 * B * n_samples + 4.0 * B * T * n_mels);
  conv_prof_begin(8, (double)B * T * (5.0 * n_fft * log2n + 2.0 * (n_fft / 2 + 1) * n_mels), (hipStream_t)stream);
  hipLaunchKernelGGL(logmel_kernel, grid, block, 0, (hipStream_t)stream, raw, ld_raw, n_samples, T, n_fft, log2n, hop,
                     win_length, window, mel_basis, n_mels, eps, mean, stdv, out, ldo, fpb, center);
  conv_prof_end(8, (hipStream_t)stream);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

extern "C" const char* crk_version(void) { return "crank_hip 0.1 (gfx950)"; }
