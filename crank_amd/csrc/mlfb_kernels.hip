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
#include "switches.h"
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

// ---------------------------------------------------------------------------------------------------------------------
// n_fft = 1024 (every recipe of the reference): one WAVE transforms TWO consecutive frames as one complex FFT
// (z = frame_a + i frame_b; the two real spectra are separated afterwards), 16 points per lane in registers:
//   lane l holds z[64 j + l], j = 0 .. 15 (coalesced loads)          -> DFT-16 over j in registers          (k1)
//   x W_1024^(k1 l); DFT-4 over the lane bits 5, 4 as two radix-2 exchanges                                  (r)
//   x W_64^(q r), q = l & 15; 16 x 16 transpose inside each row of 16 lanes through LDS; DFT-16 over q      (s)
//   -> register s of lane l holds Z[(l & 15) + 16 r + 64 s]: no workgroup barrier anywhere in the frame loop (the radix-2
//   kernel above has twelve per frame and two exposed global round trips), the samples of the next pair are requested
//   before the current pair's transform.  Magnitudes of both frames go to a per-wave LDS strip; the mel projection reads each
//   triangular filter's run of bins against a compact weight table in LDS built once per workgroup.
// profiles/round5_logmel_kernel.txt: 617 us -> see there, B = 64 x 65 023 samples.
struct lm_c { float re, im; };
__device__ __forceinline__ lm_c lm_add(lm_c a, lm_c b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ lm_c lm_sub(lm_c a, lm_c b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ lm_c lm_mul(lm_c a, lm_c b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ lm_c lm_mnegi(lm_c a) { return {a.im, -a.re}; }   // x (-i)
__device__ __forceinline__ lm_c lm_mposi(lm_c a) { return {-a.im, a.re}; }   // x (+i)
// forward DFT-4 (W_4 = -i) of (x0, x1, x2, x3) -> (y0, y1, y2, y3)
#define LM_DFT4(x0, x1, x2, x3, y0, y1, y2, y3)                                  \
  {                                                                              \
    const lm_c s02_ = lm_add(x0, x2), d02_ = lm_sub(x0, x2), s13_ = lm_add(x1, x3), d13_ = lm_sub(x1, x3); \
    y0 = lm_add(s02_, s13_); y1 = lm_add(d02_, lm_mnegi(d13_)); y2 = lm_sub(s02_, s13_); y3 = lm_add(d02_, lm_mposi(d13_)); \
  }
// in-register forward DFT-16, natural order in and out: n = 4 a + b, k = c + 4 d
__device__ __forceinline__ void lm_fft16(lm_c (&v)[16]) {
  lm_c y[16];
#pragma unroll
  for (int b = 0; b < 4; b++) LM_DFT4(v[b], v[4 + b], v[8 + b], v[12 + b], y[b], y[4 + b], y[8 + b], y[12 + b])  // y[4 c + b]
  // x W_16^(b c): (cos, -sin) of 2 pi b c / 16
  const lm_c w1 = {0.92387953251128674f, -0.38268343236508977f}, w2 = {0.70710678118654752f, -0.70710678118654752f};
  const lm_c w3 = {0.38268343236508977f, -0.92387953251128674f}, w6 = {-0.70710678118654752f, -0.70710678118654752f};
  const lm_c w9 = {-0.92387953251128674f, 0.38268343236508977f};
  y[4 + 1] = lm_mul(y[4 + 1], w1); y[4 + 2] = lm_mul(y[4 + 2], w2); y[4 + 3] = lm_mul(y[4 + 3], w3);
  y[8 + 1] = lm_mul(y[8 + 1], w2); y[8 + 2] = lm_mnegi(y[8 + 2]);    y[8 + 3] = lm_mul(y[8 + 3], w6);
  y[12 + 1] = lm_mul(y[12 + 1], w3); y[12 + 2] = lm_mul(y[12 + 2], w6); y[12 + 3] = lm_mul(y[12 + 3], w9);
#pragma unroll
  for (int c = 0; c < 4; c++) LM_DFT4(y[4 * c], y[4 * c + 1], y[4 * c + 2], y[4 * c + 3], v[c], v[c + 4], v[c + 8], v[c + 12])
}
// exp(-2 pi i m / 1024), m = 0 .. 1023: a constant of the library, (re)written by EVERY call's lm_prep_kernel on the call's own
// stream and device (the same 8 KB every time).  As first written it was filled once per process by a launch on the first
// caller's stream behind a plain static flag: a second GPU of the process never got its table, a first call inside a stream
// capture recorded the fill instead of running it, and nothing ordered it against calls on other streams.
__device__ lm_c lm_tw[1024];
__device__ __forceinline__ lm_c lm_w1024(int m) { return lm_tw[m]; }
// Per call, BEFORE the transform kernel: each filter's nonzero run of bins [lo, hi) and a copy of its weights (runs of up to
// LM_RUN bins; a longer run is read from the basis itself).  One 64-thread workgroup per filter; the tables live in one of
// LM_NSCR library-owned slots handed out round robin (calls in flight at the same time never share a slot unless more than
// LM_NSCR of them overlap).  (Built per WORKGROUP of the transform kernel, as first written, the scan of the 513 x 80 basis cost
// more than the transforms: 270 us per call.)
#define LM_RUN 64
#define LM_NSCR 8
__device__ float lm_melw[LM_NSCR][256 * LM_RUN];
__device__ int lm_mello[LM_NSCR][256], lm_melhi[LM_NSCR][256];
__global__ __launch_bounds__(64) void lm_prep_kernel(const float* __restrict__ mel, int n_mels, int n_bins, int slot) {
  const int m = blockIdx.x, lane = threadIdx.x;
  for (int i = m * 64 + lane; i < 1024; i += gridDim.x * 64) {  // the twiddle table: this call's transform kernel reads it
    float sn, cs;
    sincospif(-2.0f * (float)i / 1024.0f, &sn, &cs);
    lm_tw[i] = {cs, sn};
  }
  int lo = n_bins, hi = 0;
  for (int k = lane; k < n_bins; k += 64)
    if (mel[(long)k * n_mels + m] != 0.f) { lo = min(lo, k); hi = max(hi, k + 1); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
  if (lane == 0) { lm_mello[slot][m] = lo; lm_melhi[slot][m] = hi; }
  if (hi - lo <= LM_RUN && lo + lane < hi) lm_melw[slot][m * LM_RUN + lane] = mel[(long)(lo + lane) * n_mels + m];
}
#define LM_WTAB 2048
#define LM_ZS 1088  // complex slots per wave: 4 rows x 16 x 17 (padded transposes) >= 1024 (the spectrum)
__global__ __launch_bounds__(256) void logmel_wave_kernel(const float* __restrict__ raw, int ld_raw, int n_samples, int T, int hop,
                                                          int win, const float* __restrict__ window, const float* __restrict__ mel,
                                                          int n_mels, float eps, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, float* __restrict__ out, int ldo,
                                                          int pairs_per_wave, int center, int slot) {
  constexpr int N = 1024, NB = 513;
  __shared__ lm_c zb[4][LM_ZS];
  __shared__ float mag[4][2][520];
  __shared__ float wtab[LM_WTAB];
  __shared__ int mlo[256], mhi[256], moff[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
  // ---- per workgroup: the filters' runs (lm_prep_kernel) packed side by side into LDS ----
  for (int m = tid; m < n_mels; m += 256) { mlo[m] = lm_mello[slot][m]; mhi[m] = lm_melhi[slot][m]; }
  __syncthreads();
  if (tid == 0) {
    int off = 0;
    for (int m = 0; m < n_mels; m++) {
      const int len = mhi[m] > mlo[m] ? mhi[m] - mlo[m] : 0;
      moff[m] = (len <= LM_RUN && off + len <= LM_WTAB) ? off : -1;  // -1: this filter's weights are read from the basis itself
      if (moff[m] >= 0) off += len;
    }
  }
  __syncthreads();
  for (int i = tid; i < n_mels * LM_RUN; i += 256) {
    const int m = i / LM_RUN, j = i - m * LM_RUN;
    if (moff[m] >= 0 && mlo[m] + j < mhi[m]) wtab[moff[m] + j] = lm_melw[slot][i];
  }
  __syncthreads();
  // ---- per lane: window taps and twiddles ----
  const int lpad = (N - win) / 2;
  float wn[16];
  lm_c twa[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int n = 64 * j + lane;
    wn[j] = (n >= lpad && n < lpad + win) ? window[n - lpad] : 0.f;
    twa[j] = lm_w1024(j * lane);  // W_1024^(k1 l), k1 = j
  }
  const int q = lane & 15, r = ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1);  // r: the DFT-4 output this lane ends up with
  const lm_c twb = lm_w1024(16 * q * r);                                   // W_64^(q r)
  const bool hi5 = (lane >> 5) & 1, hi4 = (lane >> 4) & 1;
  lm_c* zw = zb[wave];
  const float* src = raw + (long)b * ld_raw;
  const int pair0 = (blockIdx.x * 4 + wave) * pairs_per_wave;

  auto sample = [&](int t, int n) -> float {
    long pos = (long)t * hop + n;
    if (center) {  // torch.stft / librosa center=True, pad_mode="reflect"
      pos -= N / 2;
      if (pos < 0) pos = -pos;
      if (pos >= n_samples) pos = 2L * (n_samples - 1) - pos;
    }
    return (t < T && pos >= 0 && pos < n_samples) ? src[pos] : 0.f;
  };
  float xa[16], xb[16];
  if (2 * pair0 < T) {
#pragma unroll
    for (int j = 0; j < 16; j++) { xa[j] = sample(2 * pair0, 64 * j + lane); xb[j] = sample(2 * pair0 + 1, 64 * j + lane); }
  }
  for (int pi = 0; pi < pairs_per_wave; pi++) {
    const int ta = 2 * (pair0 + pi), tb = ta + 1;
    if (ta >= T) break;  // (wave-uniform)
    lm_c v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = {xa[j] * wn[j], xb[j] * wn[j]};
    if (pi + 1 < pairs_per_wave && ta + 2 < T) {  // the next pair's samples: in flight under this pair's transform
#pragma unroll
      for (int j = 0; j < 16; j++) { xa[j] = sample(ta + 2, 64 * j + lane); xb[j] = sample(ta + 3, 64 * j + lane); }
    }
    lm_fft16(v);  // over j -> k1
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
      lm_c u = lm_mul(v[k1], twa[k1]);
      // DFT-4 over the lane bits (p = 2 p1 + p0, p1 = bit 5, p0 = bit 4): radix 2 across bit 5, then across bit 4
      lm_c o = {__shfl_xor(u.re, 32, 64), __shfl_xor(u.im, 32, 64)};
      u = hi5 ? lm_sub(o, u) : lm_add(u, o);           // F[s = p1][p0] = u_{p0} + (-1)^s u_{p0 + 2}
      if (hi4 && hi5) u = lm_mnegi(u);                 // the odd branch of s = 1 carries W_4^1 = -i
      o = {__shfl_xor(u.re, 16, 64), __shfl_xor(u.im, 16, 64)};
      u = hi4 ? lm_sub(o, u) : lm_add(u, o);           // C[r = s + 2 p0] = E[s] +- W_4^s O[s]
      v[k1] = lm_mul(u, twb);
    }
    // 16 x 16 transpose inside each row of 16 lanes (row stride 17 complex: conflict-free 8-byte accesses)
    const int rowb = (lane >> 4) * (16 * 17);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) zw[rowb + k1 * 17 + q] = v[k1];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
    for (int qq = 0; qq < 16; qq++) v[qq] = zw[rowb + q * 17 + qq];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    lm_fft16(v);  // over q -> s
#pragma unroll
    for (int sx = 0; sx < 16; sx++) zw[q + 16 * r + 64 * sx] = v[sx];  // Z[k1 + 16 k2], k2 = r + 4 s
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // the two real spectra: Xa = (Z[k] + conj Z[N-k]) / 2, Xb = (Z[k] - conj Z[N-k]) / 2i -> magnitudes
    for (int k = lane; k < NB; k += 64) {
      const lm_c z = zw[k], y = zw[(N - k) & (N - 1)];
      const float ar = z.re + y.re, ai = z.im - y.im, br = z.im + y.im, bi = z.re - y.re;
      mag[wave][0][k] = 0.5f * sqrtf(ar * ar + ai * ai);
      mag[wave][1][k] = 0.5f * sqrtf(br * br + bi * bi);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int m = lane; m < n_mels; m += 64) {
      const int lo = mlo[m], hi = mhi[m];
      float sa = 0.f, sb = 0.f;
      if (moff[m] >= 0) {
        const float* wp = wtab + moff[m] - lo;
        for (int k = lo; k < hi; k++) { const float w = wp[k]; sa += mag[wave][0][k] * w; sb += mag[wave][1][k] * w; }
      } else {
        for (int k = lo; k < hi; k++) { const float w = mel[(long)k * n_mels + m]; sa += mag[wave][0][k] * w; sb += mag[wave][1][k] * w; }
      }
      float va = log10f(fmaxf(sa, eps)), vb = log10f(fmaxf(sb, eps));
      if (mean) { va = (va - mean[m]) / stdv[m]; vb = (vb - mean[m]) / stdv[m]; }
      out[((long)b * T + ta) * ldo + m] = va;
      if (tb < T) out[((long)b * T + tb) * ldo + m] = vb;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the strips are rewritten by the next pair
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
  static int wave_env = -1;  // CRK_LOGMEL_WAVE=0: the radix-2 workgroup-per-frame kernel for every size (A/B measurements)
  if (wave_env < 0) wave_env = crk_sw().logmel_wave;
  if (n_fft == 1024 && wave_env) {
    static unsigned next_slot = 0;
    const int slot = (int)(next_slot++ % LM_NSCR);
    conv_prof_bytes(8, 4.0 * B * n_samples + 4.0 * B * T * n_mels);
    conv_prof_begin(8, (double)B * T * (5.0 * n_fft * log2n + 2.0 * (n_fft / 2 + 1) * n_mels), (hipStream_t)stream);  // (both launches)
    hipLaunchKernelGGL(lm_prep_kernel, dim3(n_mels), dim3(64), 0, (hipStream_t)stream, mel_basis, n_mels, n_fft / 2 + 1, slot);
    const int npairs = (T + 1) / 2;
    int ppw = (int)(((long long)npairs * B + 4 * 512 - 1) / (4 * 512));  // pairs per wave: one round of 2 x 256 workgroups
    ppw = ppw < 1 ? 1 : (ppw > 32 ? 32 : ppw);
    dim3 grid((npairs + 4 * ppw - 1) / (4 * ppw), B), block(256);
    hipLaunchKernelGGL(logmel_wave_kernel, grid, block, 0, (hipStream_t)stream, raw, ld_raw, n_samples, T, hop, win_length, window,
                       mel_basis, n_mels, eps, mean, stdv, out, ldo, ppw, center, slot);
    conv_prof_end(8, (hipStream_t)stream);
    CRK_CHECK_LAUNCH();
    return CRK_OK;
  }
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
