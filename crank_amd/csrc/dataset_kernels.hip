// Batch assembly from an HBM-resident corpus and decode-side post-processing (gfx950).
//
// Replaces the per-sample numpy work of the reference's DataLoader workers plus the
// host->device copy of every batch (crank/net/trainer/dataset.py:58-198,229-258,288-293;
// crank/net/trainer/basetrainer.py:311-320,340-386).  All of it is HBM-bound byte moving
// with a little float64 arithmetic; results are bit-identical to numpy / sklearn because
// every operation is evaluated in the precision numpy uses and rounded where numpy rounds
// (no fused multiply-add: contraction is switched off for this file).
#include "common.h"
#include "../../include/crank_hip.h"

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------
// StandardScaler.transform / inverse_transform on float32 rows.  sklearn works in place on a
// float32 copy with float64 statistics: X -= mean_; X /= scale_  (inverse: X *= scale_; X += mean_),
// i.e. each step in float64, rounded to float32 after it.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scaler_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                    long long N, int D, const double* __restrict__ mean,
                                                    const double* __restrict__ scale, int inverse) {
  const long long total = N * D;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / D;
    const int d = (int)(i - n * D);
    const double v = (double)x[n * ldx + d];
    float r;
    if (!inverse) {
      const float a = (float)(v - mean[d]);
      r = (float)((double)a / scale[d]);
    } else {
      const float a = (float)(v * scale[d]);
      r = (float)((double)a + mean[d]);
    }
    y[n * ldy + d] = r;
  }
}

extern "C" int crk_scaler_apply(const float* x, int ldx, float* y, int ldy, long long N, int D, const double* mean,
                                const double* scale, int inverse, void* stream) {
  if (!x || !y || !mean || !scale || N < 0 || D <= 0 || ldx < D || ldy < D) return CRK_ERR_ARG;
  if (N == 0) return CRK_OK;
  const long long total = N * D;
  const int nb = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(scaler_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, N, D, mean, scale, inverse);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// convert_f0 (dataset.py:288-293) in float64, left to right like the numpy expression
__device__ __forceinline__ double convert_f0(double l, double m_org, double s_org, double m_cv, double s_cv) {
  return (l - m_org) / s_org * s_cv + m_cv;
}

// ------------------------------------------------------------------------------
// collate: one workgroup assembles CL_FR frames of one batch row.
//   frames t <  min(flen, T): copied from the utterance (from frame p on when flen > T)
//   frames t >= flen        : 0.0 / False / -100 (dataset.py:176-190)
// ------------------------------------------------------------------------------
#define CL_FR 32

struct CollateArgs {
  crk_collate_desc d;
  const int* picks;  // [3][B]: utterance, first kept frame p, conversion-target speaker
  int B, T;
  float* cv_lcf0;
  long long *org_h, *cv_h;
  float *org_onehot, *cv_onehot;
  unsigned char* mask;
  long long* flen;
};

__global__ __launch_bounds__(256) void collate_kernel(const CollateArgs a) {
  const int b = blockIdx.y, t0 = blockIdx.x * CL_FR, tid = threadIdx.x;
  const int T = a.T;
  const int u = a.picks[b], p = a.picks[a.B + b], cv = a.picks[2 * a.B + b];
  const long long s0 = a.d.utt_start[u];
  const long long len = a.d.utt_start[u + 1] - s0;
  const int org = a.d.utt_spk[u];
  const bool crop = len > T;
  const long long first = s0 + (crop ? p : 0) + t0;  // corpus frame of this block's frame 0
  const int real = crop ? T : (int)len;              // frames [0, real) of the row come from the utterance
  const int nfr = min(CL_FR, T - t0);
  const int nv = max(0, min(real - t0, nfr));
  const long long row0 = (long long)b * T + t0;

  for (int s = 0; s < a.d.n_streams; ++s) {
    const crk_collate_stream st = a.d.streams[s];
    const float* src = st.src + first * st.ld + st.col0;
    float* dst = st.dst + row0 * st.ncols;
    const bool vec = ((st.ncols | st.ld | st.col0) & 3) == 0 && ((((uintptr_t)st.src) | ((uintptr_t)st.dst)) & 15) == 0;
    if (vec) {
      const int c4 = st.ncols >> 2;
      for (int i = tid; i < nfr * c4; i += 256) {
        const int t = i / c4, c = i - t * c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < nv) v = *(const f32x4*)(src + (long long)t * st.ld + 4 * c);
        *(f32x4*)(dst + 4 * i) = v;
      }
    } else {
      for (int i = tid; i < nfr * st.ncols; i += 256) {
        const int t = i / st.ncols, c = i - t * st.ncols;
        dst[i] = t < nv ? src[(long long)t * st.ld + c] : 0.f;
      }
    }
  }
  if (tid < nfr) {
    const bool valid = tid < nv;
    const long long o = row0 + tid;
    if (a.cv_lcf0) {
      float r = 0.f;
      if (valid)
        r = (float)convert_f0((double)a.d.lcf0_raw[first + tid], a.d.spk_lcf0_mean[org], a.d.spk_lcf0_std[org],
                              a.d.spk_lcf0_mean[cv], a.d.spk_lcf0_std[cv]);
      a.cv_lcf0[o] = r;
    }
    if (a.org_h) a.org_h[o] = valid ? org : -100;
    if (a.cv_h) a.cv_h[o] = valid ? cv : -100;
    if (a.mask) a.mask[o] = valid ? 1 : 0;
  }
  const int S = a.d.n_spk;
  if (a.org_onehot || a.cv_onehot) {
    for (int i = tid; i < nfr * S; i += 256) {
      const int t = i / S, c = i - t * S;
      const bool valid = t < nv;
      if (a.org_onehot) a.org_onehot[row0 * S + i] = (valid && c == org) ? 1.f : 0.f;
      if (a.cv_onehot) a.cv_onehot[row0 * S + i] = (valid && c == cv) ? 1.f : 0.f;
    }
  }
  if (a.flen && blockIdx.x == 0 && tid == 0) a.flen[b] = len;  // the utterance's own length, also when cropped (dataset.py:88)
}

// raw waveform of a batch row, padded / cropped like padding_raw (dataset.py:261-285): target length
// fftl + hop * T - 1;
//   short utterance, or crop start p == 0: the waveform is reflect-padded by fftl/2 on both sides when it is
//     shorter than target - fftl, and taken as it is otherwise (the reference's quirk: no left padding then);
//   p > 0: fftl/2 zeros, then the waveform from sample p * hop on;
// then zeros up to the target length, or cut.
__global__ __launch_bounds__(256) void collate_raw_kernel(const crk_collate_desc d, const int* __restrict__ picks, int B,
                                                         int T, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int u = picks[b], p = picks[B + b];
  const long long r0 = d.raw_start[u];
  const long long len = d.raw_start[u + 1] - r0;
  const long long flen = d.utt_start[u + 1] - d.utt_start[u];
  const long long target = (long long)d.fftl + (long long)d.hop * T - 1;
  const int hf = d.fftl / 2;
  const bool both = (T - flen > 0) || p == 0;
  const bool reflect = both && len < target - d.fftl;
  const float* x = d.raw + r0;
  float* o = out + (long long)b * target;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < target; s += (long long)gridDim.x * 256) {
    float v = 0.f;
    if (both) {
      if (reflect) {
        if (s < len + 2 * hf) {
          long long q = s - hf, idx;
          if (len == 1) idx = 0;
          else {
            const long long m = 2 * (len - 1);
            long long t = q % m;
            if (t < 0) t += m;
            idx = t < len ? t : m - t;
          }
          v = x[idx];
        }
      } else if (s < len) {
        v = x[s];
      }
    } else if (s >= hf) {
      const long long q = (long long)p * d.hop + s - hf;
      if (q < len) v = x[q];
    }
    o[s] = v;
  }
}

extern "C" int crk_collate_batch(const crk_collate_desc* desc, const int* picks, int B, int T, float* cv_lcf0,
                                 long long* org_h, long long* cv_h, float* org_onehot, float* cv_onehot,
                                 unsigned char* mask, long long* flen, float* raw_out, void* stream) {
  if (!desc || !picks || B <= 0 || T <= 0) return CRK_ERR_ARG;
  if (desc->n_streams < 0 || desc->n_streams > CRK_COLLATE_MAX_STREAMS || !desc->utt_start || !desc->utt_spk) return CRK_ERR_ARG;
  for (int s = 0; s < desc->n_streams; ++s) {
    const crk_collate_stream& st = desc->streams[s];
    if (!st.src || !st.dst || st.ncols <= 0 || st.col0 < 0 || st.ld < st.col0 + st.ncols) return CRK_ERR_ARG;
  }
  if (cv_lcf0 && (!desc->lcf0_raw || !desc->spk_lcf0_mean || !desc->spk_lcf0_std)) return CRK_ERR_ARG;
  if ((org_onehot || cv_onehot) && desc->n_spk <= 0) return CRK_ERR_ARG;
  CollateArgs a;
  a.d = *desc;
  a.picks = picks, a.B = B, a.T = T, a.cv_lcf0 = cv_lcf0, a.org_h = org_h, a.cv_h = cv_h;
  a.org_onehot = org_onehot, a.cv_onehot = cv_onehot, a.mask = mask, a.flen = flen;
  hipLaunchKernelGGL(collate_kernel, dim3((T + CL_FR - 1) / CL_FR, B), dim3(256), 0, (hipStream_t)stream, a);
  if (raw_out) {
    if (!desc->raw || !desc->raw_start || desc->fftl <= 1 || desc->hop <= 0) return CRK_ERR_ARG;
    const long long target = (long long)desc->fftl + (long long)desc->hop * T - 1;
    const int gx = (int)((target + 1023) / 1024 < 256 ? (target + 1023) / 1024 : 256);
    hipLaunchKernelGGL(collate_raw_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, *desc, picks, B, T, raw_out);
  }
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// ------------------------------------------------------------------------------
// decode side (basetrainer.py:311-320, 372-385): for every frame of a padded batch
//   org_cf0     = fl32(fl32(fl64(lcf0) * scale) + mean)      inverse of the global lcf0 scaler (float32 in, float32 out)
//   cv_cf0      = convert_f0(org_cf0)                          float64
//   f0          = exp(cv_cf0) * uv                             float64
//   normed_lcf0 = (cv_cf0 - mean) / scale                      float64
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_f0_kernel(const float* __restrict__ lcf0, const float* __restrict__ uv, int B, int T,
                                                       const int* __restrict__ org_spk, const int* __restrict__ cv_spk,
                                                       double g_mean, double g_scale, int has_global,
                                                       const double* __restrict__ spk_mean, const double* __restrict__ spk_std,
                                                       double* __restrict__ cv_cf0, double* __restrict__ f0,
                                                       double* __restrict__ normed) {
  const long long total = (long long)B * T;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / T);
    const int o = org_spk[b], c = cv_spk[b];
    float l = lcf0[i];
    if (has_global) {
      const float s = (float)((double)l * g_scale);
      l = (float)((double)s + g_mean);
    }
    const double cvv = convert_f0((double)l, spk_mean[o], spk_std[o], spk_mean[c], spk_std[c]);
    if (cv_cf0) cv_cf0[i] = cvv;
    if (f0) f0[i] = exp(cvv) * (double)uv[i];
    if (normed) normed[i] = has_global ? (cvv - g_mean) / g_scale : cvv;
  }
}

extern "C" int crk_decode_f0(const float* lcf0, const float* uv, int B, int T, const int* org_spk, const int* cv_spk,
                             double lcf0_mean, double lcf0_scale, int has_lcf0_scaler, const double* spk_lcf0_mean,
                             const double* spk_lcf0_std, double* cv_lcf0, double* f0, double* normed_lcf0, void* stream) {
  if (!lcf0 || !org_spk || !cv_spk || !spk_lcf0_mean || !spk_lcf0_std || B <= 0 || T <= 0) return CRK_ERR_ARG;
  if (f0 && !uv) return CRK_ERR_ARG;
  const long long total = (long long)B * T;
  const int nb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(decode_f0_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, lcf0, uv, B, T, org_spk, cv_spk, lcf0_mean,
                     lcf0_scale, has_lcf0_scaler, spk_lcf0_mean, spk_lcf0_std, cv_lcf0, f0, normed_lcf0);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
