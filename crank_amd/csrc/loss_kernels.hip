// Loss kernels and small pointwise helpers.
//   masked L1 / MSE means      - crank/net/module/loss.py:30-47 (CustomFeatureLoss),
//                                commitment / LSGAN MSE of trainer_vqvae.py:227-237,
//                                trainer_lsgan.py:154-170 expressed as masked means
//   cross entropy, ignore -100 - crank/net/trainer/utils.py:26 + trainer_vqvae.py:177-198
//   multi-resolution STFT loss - crank/net/module/loss.py:50-114
// All reductions are two-stage (per-workgroup partials, then one finishing block) so
// results are deterministic; there are no host synchronisations.
#include "common.h"

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// Finishing step of the two-stage reductions: ONE workgroup sums the partials in a fixed order.  (A "last workgroup
// finishes" scheme was measured and dropped: a device-scope fence per workgroup costs an L2 write-back on this
// multi-XCD part - 1024 of them turned a 12 us kernel into a 34 us one, a finishing launch costs ~4 us.)
__device__ __forceinline__ void mean_from_partials(const float* part, int nblocks, float* out, float* sh) {
  float s = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) { s += part[2 * i]; c += part[2 * i + 1]; }
  s = block_sum_256(s, sh);
  c = block_sum_256(c, sh);
  if (threadIdx.x == 0) { out[0] = s / c; out[1] = c; }  // 0/0 -> NaN like torch's mean of an empty tensor
}

// ------------------------------------------------------------------------------
// masked mean of |x-y| or (x-y)^2 over elements whose frame mask is set.
// x,y: [N, D] with row strides; mask: [N] bytes (nullptr = all ones);
// out[0] = mean, out[1] = count (elements).  mode 0 = L1, 1 = MSE.
// y may be nullptr with a constant target yconst (LSGAN real->1 / fake->0).
// ------------------------------------------------------------------------------
#define LOSS_MAX_BLOCKS 1024
#define LOSS_MAX_RES 4                                  // STFT resolutions of one fused launch

__global__ __launch_bounds__(256) void masked_loss_partial(const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ y, int ldy, float yconst,
                                                           const unsigned char* __restrict__ mask, long N, int D,
                                                           int mode, float* __restrict__ part) {
  __shared__ float sh[4];
  float s = 0.f, c = 0.f;
  const long total = N * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / D;
    const int d = (int)(i - n * D);
    if (mask && !mask[n]) continue;
    const float yv = y ? y[n * ldy + d] : yconst;
    const float df = x[n * ldx + d] - yv;
    s += mode == 0 ? fabsf(df) : df * df;
    c += 1.f;
  }
  s = block_sum_256(s, sh);
  c = block_sum_256(c, sh);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = c; }
}

// four consecutive channels per thread (16-byte loads): D, the row strides and the pointers are multiples of 4 floats,
// y is a tensor.  MODE 0 / 1: partials {sum, count}; 2: {sum |d|, sum d^2, count} (masked_loss_both_partial's).
template <int MODE>
__global__ __launch_bounds__(256) void masked_loss_partial4(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ y, int ldy,
                                                            const unsigned char* __restrict__ mask, long N, int D4,
                                                            float* __restrict__ part) {
  __shared__ float sh[4];
  float s1 = 0.f, s2 = 0.f, c = 0.f;
  const long total = N * D4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / D4;
    const int d = (int)(i - n * D4) * 4;
    if (mask && !mask[n]) continue;
    const float4 xv = *reinterpret_cast<const float4*>(x + n * ldx + d);
    const float4 yv = *reinterpret_cast<const float4*>(y + n * ldy + d);
    const float df[4] = {xv.x - yv.x, xv.y - yv.y, xv.z - yv.z, xv.w - yv.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (MODE != 1) s1 += fabsf(df[j]);
      if (MODE != 0) s2 += df[j] * df[j];
    }
    c += 4.f;
  }
  if (MODE != 1) s1 = block_sum_256(s1, sh);
  if (MODE != 0) s2 = block_sum_256(s2, sh);
  c = block_sum_256(c, sh);
  if (threadIdx.x == 0) {
    if (MODE == 2) { part[3 * blockIdx.x] = s1; part[3 * blockIdx.x + 1] = s2; part[3 * blockIdx.x + 2] = c; }
    else { part[2 * blockIdx.x] = MODE == 0 ? s1 : s2; part[2 * blockIdx.x + 1] = c; }
  }
}
__device__ __host__ inline bool loss_vec4_ok(const void* a, const void* b, const void* c, int D, int l0, int l1, int l2) {
  return !((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) && !((D | l0 | l1 | l2) & 3);
}

__global__ __launch_bounds__(256) void masked_loss_final(const float* __restrict__ part, int nblocks,
                                                         float* __restrict__ out) {
  __shared__ float sh[4];
  mean_from_partials(part, nblocks, out, sh);
}

// L1 and MSE of the same pair in one pass (the trainers ask for both on the decoded features): partials {sum|d|, sum d^2,
// count}; out4 = {L1 mean, count, MSE mean, count}, each half what the single-mode entry points write.  Same loop and
// reduction order as the single-mode kernels: identical values.
__global__ __launch_bounds__(256) void masked_loss_both_partial(const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ y, int ldy,
                                                                const unsigned char* __restrict__ mask, long N, int D,
                                                                float* __restrict__ part) {
  __shared__ float sh[4];
  float s1 = 0.f, s2 = 0.f, c = 0.f;
  const long total = N * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / D;
    const int d = (int)(i - n * D);
    if (mask && !mask[n]) continue;
    const float df = x[n * ldx + d] - y[n * ldy + d];
    s1 += fabsf(df);
    s2 += df * df;
    c += 1.f;
  }
  s1 = block_sum_256(s1, sh);
  s2 = block_sum_256(s2, sh);
  c = block_sum_256(c, sh);
  if (threadIdx.x == 0) { part[3 * blockIdx.x] = s1; part[3 * blockIdx.x + 1] = s2; part[3 * blockIdx.x + 2] = c; }
}
__global__ __launch_bounds__(256) void masked_loss_both_final(const float* __restrict__ part, int nblocks,
                                                              float* __restrict__ out) {
  __shared__ float sh[4];
  float s1 = 0.f, s2 = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) { s1 += part[3 * i]; s2 += part[3 * i + 1]; c += part[3 * i + 2]; }
  s1 = block_sum_256(s1, sh);
  s2 = block_sum_256(s2, sh);
  c = block_sum_256(c, sh);
  if (threadIdx.x == 0) { out[0] = s1 / c; out[1] = c; out[2] = s2 / c; out[3] = c; }
}
extern "C" int crk_masked_loss_both_fwd(const float* x, int ldx, const float* y, int ldy, const unsigned char* mask,
                                        long long N, int D, float* out4, float* scratch, void* stream) {
  if (!x || !y || !out4 || !scratch || N < 0 || D <= 0) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  long b = (N * D + 255) / 256;
  const int nb = (int)(b > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : (b < 1 ? 1 : b));
  if (loss_vec4_ok(x, y, nullptr, D, ldx, ldy, 0)) {
    long b4 = (N * (D / 4) + 255) / 256;
    const int nb4 = (int)(b4 > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : (b4 < 1 ? 1 : b4));
    hipLaunchKernelGGL(masked_loss_partial4<2>, dim3(nb4), dim3(256), 0, s, x, ldx, y, ldy, mask, (long)N, D / 4, scratch);
    hipLaunchKernelGGL(masked_loss_both_final, dim3(1), dim3(256), 0, s, scratch, nb4, out4);
    CRK_CHECK_LAUNCH();
    return CRK_OK;
  }
  hipLaunchKernelGGL(masked_loss_both_partial, dim3(nb), dim3(256), 0, s, x, ldx, y, ldy, mask, (long)N, D, scratch);
  hipLaunchKernelGGL(masked_loss_both_final, dim3(1), dim3(256), 0, s, scratch, nb, out4);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

__global__ __launch_bounds__(256) void masked_loss_bwd(const float* __restrict__ x, int ldx,
                                                       const float* __restrict__ y, int ldy, float yconst,
                                                       const unsigned char* __restrict__ mask, long N, int D,
                                                       int mode, const float* __restrict__ stat,
                                                       const float* __restrict__ gout, float* __restrict__ dx,
                                                       int lddx, float* __restrict__ dy, int lddy,
                                                       const float* __restrict__ add, int ldadd,
                                                       const float* __restrict__ add_scale) {
  const float g = gout[0] / stat[1];
  const float as = add_scale ? add_scale[0] : 1.f;
  const long total = N * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / D;
    const int d = (int)(i - n * D);
    float r = 0.f;
    if (!mask || mask[n]) {
      const float yv = y ? y[n * ldy + d] : yconst;
      const float df = x[n * ldx + d] - yv;
      r = mode == 0 ? (df > 0.f ? g : (df < 0.f ? -g : 0.f)) : 2.f * df * g;
    }
    if (dx) dx[n * lddx + d] = add ? add[n * ldadd + d] * as + r : r;
    if (dy) dy[n * lddy + d] = -r;
  }
}

// the same, four consecutive channels of a frame per thread (16-byte accesses): D, the row strides and the pointers are
// multiples of 4 floats, y is a tensor, only dx is wanted.  Per-element arithmetic unchanged: identical values.
__global__ __launch_bounds__(256) void masked_loss_bwd4(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ y, int ldy,
                                                        const unsigned char* __restrict__ mask, long N, int D4,
                                                        int mode, const float* __restrict__ stat,
                                                        const float* __restrict__ gout, float* __restrict__ dx,
                                                        int lddx, const float* __restrict__ add, int ldadd,
                                                        const float* __restrict__ add_scale) {
  const float g = gout[0] / stat[1];
  const float as = add_scale ? add_scale[0] : 1.f;
  const long total = N * D4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / D4;
    const int d = (int)(i - n * D4) * 4;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!mask || mask[n]) {
      const float4 xv = *reinterpret_cast<const float4*>(x + n * ldx + d);
      const float4 yv = *reinterpret_cast<const float4*>(y + n * ldy + d);
      const float df[4] = {xv.x - yv.x, xv.y - yv.y, xv.z - yv.z, xv.w - yv.w};
      float rr[4];
#pragma unroll
      for (int j = 0; j < 4; j++) rr[j] = mode == 0 ? (df[j] > 0.f ? g : (df[j] < 0.f ? -g : 0.f)) : 2.f * df[j] * g;
      r = make_float4(rr[0], rr[1], rr[2], rr[3]);
    }
    if (add) {
      const float4 a = *reinterpret_cast<const float4*>(add + n * ldadd + d);
      r = make_float4(a.x * as + r.x, a.y * as + r.y, a.z * as + r.z, a.w * as + r.w);
    }
    *reinterpret_cast<float4*>(dx + n * lddx + d) = r;
  }
}

// The quantizer's input gradient where several gradients of the same tensor meet (vqvae2.py:171-190: the top stack's
// straight-through value feeds its decoder AND the last decoder's concatenation; the encoder output feeds the quantizer AND
// the speaker-adversarial net, spkradv.py:74-76): t = (a1 + a2) + 2 (x - e) g  goes to the tensor that was added to x inside
// the op (dsum, optional), t + a3 to x itself (dx).  Every addition is the one autograd's accumulation would have made
// (fp32 additions commute: the values are bit-identical to separate launches), a1 / a2 / a3 optional, four channels per
// thread.
__global__ __launch_bounds__(256) void masked_loss_join4(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ y, int ldy,
                                                         const unsigned char* __restrict__ mask, long N, int D4,
                                                         const float* __restrict__ stat, const float* __restrict__ gout,
                                                         float* __restrict__ dx, int lddx, float* __restrict__ dsum,
                                                         int ldsum, const float* __restrict__ a1, int ld1,
                                                         const float* __restrict__ a2, int ld2,
                                                         const float* __restrict__ a3, int ld3) {
#pragma clang fp contract(off)  // every product and sum below is rounded on its own, as the separate launches round them
  const float g = gout[0] / stat[1];
  const long total = N * D4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / D4;
    const int d = (int)(i - n * D4) * 4;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (!mask || mask[n]) {
      const float4 xv = *reinterpret_cast<const float4*>(x + n * ldx + d);
      const float4 yv = *reinterpret_cast<const float4*>(y + n * ldy + d);
      const float df[4] = {xv.x - yv.x, xv.y - yv.y, xv.z - yv.z, xv.w - yv.w};
#pragma unroll
      for (int j = 0; j < 4; j++) r[j] = 2.f * df[j] * g;
    }
    if (a1 || a2) {
      float4 s = *reinterpret_cast<const float4*>((a1 ? a1 + n * ld1 : a2 + n * ld2) + d);
      if (a1 && a2) {
        const float4 b = *reinterpret_cast<const float4*>(a2 + n * ld2 + d);
        s = make_float4(s.x + b.x, s.y + b.y, s.z + b.z, s.w + b.w);
      }
      // (masked_loss_bwd4 forms fma(s, scale, r) with its scale at 1: one rounding, of s + r)
      r[0] = s.x + r[0]; r[1] = s.y + r[1]; r[2] = s.z + r[2]; r[3] = s.w + r[3];
    }
    if (dsum) *reinterpret_cast<float4*>(dsum + n * ldsum + d) = make_float4(r[0], r[1], r[2], r[3]);
    if (a3) {
      const float4 c = *reinterpret_cast<const float4*>(a3 + n * ld3 + d);
      r[0] += c.x; r[1] += c.y; r[2] += c.z; r[3] += c.w;
    }
    *reinterpret_cast<float4*>(dx + n * lddx + d) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

static int loss_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > LOSS_MAX_BLOCKS) b = LOSS_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int crk_masked_loss_fwd(const float* x, int ldx, const float* y, int ldy, float yconst,
                                   const unsigned char* mask, long long N, int D, int mode, float* out2,
                                   float* scratch, void* stream) {
  if (!x || !out2 || !scratch || N < 0 || D <= 0) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (y && loss_vec4_ok(x, y, nullptr, D, ldx, ldy, 0)) {
    const int nb4 = loss_blocks(N * (D / 4));
    if (mode == 0) hipLaunchKernelGGL(masked_loss_partial4<0>, dim3(nb4), dim3(256), 0, s, x, ldx, y, ldy, mask, (long)N, D / 4, scratch);
    else hipLaunchKernelGGL(masked_loss_partial4<1>, dim3(nb4), dim3(256), 0, s, x, ldx, y, ldy, mask, (long)N, D / 4, scratch);
    hipLaunchKernelGGL(masked_loss_final, dim3(1), dim3(256), 0, s, scratch, nb4, out2);
    CRK_CHECK_LAUNCH();
    return CRK_OK;
  }
  const int nb = loss_blocks(N * D);
  hipLaunchKernelGGL(masked_loss_partial, dim3(nb), dim3(256), 0, s, x, ldx, y, ldy, yconst, mask, (long)N, D, mode, scratch);
  hipLaunchKernelGGL(masked_loss_final, dim3(1), dim3(256), 0, s, scratch, nb, out2);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

extern "C" int crk_masked_loss_bwd_acc(const float* x, int ldx, const float* y, int ldy, float yconst,
                                       const unsigned char* mask, long long N, int D, int mode, const float* stat2,
                                       const float* gout, float* dx, int lddx, float* dy, int lddy, const float* add,
                                       int ldadd, const float* add_scale, void* stream) {
  if (!x || !stat2 || !gout) return CRK_ERR_ARG;
  const bool al16 = !((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dx) | ((uintptr_t)add)) & 15);
  if (y && dx && !dy && al16 && !((D | ldx | ldy | lddx | (add ? ldadd : 0)) & 3)) {
    const int nb4 = loss_blocks(N * (D / 4));
    hipLaunchKernelGGL(masked_loss_bwd4, dim3(nb4), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, mask, (long)N, D / 4, mode,
                       stat2, gout, dx, lddx, add, ldadd, add_scale);
    CRK_CHECK_LAUNCH();
    return CRK_OK;
  }
  const int nb = loss_blocks(N * D);
  hipLaunchKernelGGL(masked_loss_bwd, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, yconst, mask, (long)N,
                     D, mode, stat2, gout, dx, lddx, dy, lddy, add, ldadd, add_scale);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

extern "C" int crk_vq_commit_bwd(const float* x, int ldx, const float* e, int lde, const unsigned char* mask, long long N,
                                 int D, const float* stat2, const float* gout, float* dx, int lddx, float* dsum, int ldsum,
                                 const float* a1, int ld1, const float* a2, int ld2, const float* a3, int ld3,
                                 void* stream) {
  if (!x || !e || !stat2 || !gout || !dx || N < 0 || D <= 0) return CRK_ERR_ARG;
  const uintptr_t al = ((uintptr_t)x) | ((uintptr_t)e) | ((uintptr_t)dx) | ((uintptr_t)dsum) | ((uintptr_t)a1) |
                       ((uintptr_t)a2) | ((uintptr_t)a3);
  const int lds_ = D | ldx | lde | lddx | (dsum ? ldsum : 0) | (a1 ? ld1 : 0) | (a2 ? ld2 : 0) | (a3 ? ld3 : 0);
  if ((al & 15) || (lds_ & 3)) return CRK_ERR_UNSUPPORTED;  // (the caller then joins the gradients with separate additions)
  if (N == 0) return CRK_OK;
  hipLaunchKernelGGL(masked_loss_join4, dim3(loss_blocks(N * (D / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, e, lde, mask,
                     (long)N, D / 4, stat2, gout, dx, lddx, dsum, ldsum, a1, ld1, a2, ld2, a3, ld3);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

extern "C" int crk_masked_loss_bwd(const float* x, int ldx, const float* y, int ldy, float yconst,
                                   const unsigned char* mask, long long N, int D, int mode, const float* stat2,
                                   const float* gout, float* dx, int lddx, float* dy, int lddy, void* stream) {
  return crk_masked_loss_bwd_acc(x, ldx, y, ldy, yconst, mask, N, D, mode, stat2, gout, dx, lddx, dy, lddy, nullptr, 0, nullptr,
                                 stream);
}

extern "C" int crk_loss_scratch_floats() { return 2 * LOSS_MAX_BLOCKS * LOSS_MAX_RES + 8; }

// ------------------------------------------------------------------------------
// cross entropy over frames with ignore_index: one thread per frame (C <= 64).
// forward also emits the un-normalised gradient softmax - onehot (0 on ignored rows)
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_partial(const float* __restrict__ logits, int ldl,
                                                  const long long* __restrict__ target, long N, int C, int ignore,
                                                  float* __restrict__ dlogits, float* __restrict__ part) {
  __shared__ float sh[4];
  float s = 0.f, c = 0.f;
  for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < N; n += (long)gridDim.x * 256) {
    const long long tg = target[n];
    const float* lp = logits + n * ldl;
    if (tg == ignore) {
      if (dlogits) for (int j = 0; j < C; j++) dlogits[n * C + j] = 0.f;
      continue;
    }
    float m = -INFINITY;
    for (int j = 0; j < C; j++) m = fmaxf(m, lp[j]);
    float se = 0.f;
    for (int j = 0; j < C; j++) se += expf(lp[j] - m);
    const float lse = m + logf(se);
    s += lse - lp[tg];
    c += 1.f;
    if (dlogits)
      for (int j = 0; j < C; j++) dlogits[n * C + j] = expf(lp[j] - lse) - (j == tg ? 1.f : 0.f);
  }
  s = block_sum_256(s, sh);
  c = block_sum_256(c, sh);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = c; }
}

// The same for an even number of classes <= 16 on contiguous rows (the 14 speakers of the classifier and the adversarial net):
// the row is read ONCE, as 8-byte pieces into registers (ce_partial walks it three times with 4-byte loads 56 bytes apart
// from lane to lane, and writes the gradient the same way: 42 scattered loads and 14 stores per frame), the arithmetic on
// the registers is ce_partial's - same loss, count and gradient, bit for bit.
template <int CH>  // C / 2
__global__ __launch_bounds__(256) void ce_partial_regs(const float* __restrict__ logits, const long long* __restrict__ target,
                                                       long N, int ignore, float* __restrict__ dlogits, float* __restrict__ part) {
  __shared__ float sh[4];
  constexpr int C = 2 * CH;
  float s = 0.f, c = 0.f;
  for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < N; n += (long)gridDim.x * 256) {
    const long long tg = target[n];
    float2* dp = dlogits ? reinterpret_cast<float2*>(dlogits + n * C) : nullptr;
    if (tg == ignore) {
      if (dp) {
#pragma unroll
        for (int k = 0; k < CH; k++) dp[k] = make_float2(0.f, 0.f);
      }
      continue;
    }
    const float2* lp2 = reinterpret_cast<const float2*>(logits + n * C);
    float lp[C];
#pragma unroll
    for (int k = 0; k < CH; k++) { const float2 t = lp2[k]; lp[2 * k] = t.x; lp[2 * k + 1] = t.y; }
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < C; j++) m = fmaxf(m, lp[j]);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < C; j++) se += expf(lp[j] - m);
    const float lse = m + logf(se);
    float lt = 0.f;
#pragma unroll
    for (int j = 0; j < C; j++) lt = (j == (int)tg) ? lp[j] : lt;
    s += lse - lt;
    c += 1.f;
    if (dp) {
#pragma unroll
      for (int k = 0; k < CH; k++)
        dp[k] = make_float2(expf(lp[2 * k] - lse) - ((2 * k) == (int)tg ? 1.f : 0.f), expf(lp[2 * k + 1] - lse) - ((2 * k + 1) == (int)tg ? 1.f : 0.f));
    }
  }
  s = block_sum_256(s, sh);
  c = block_sum_256(c, sh);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = c; }
}

__global__ __launch_bounds__(256) void scale_by_kernel(float* __restrict__ v, long total,
                                                       const float* __restrict__ gout,
                                                       const float* __restrict__ stat, float* __restrict__ out) {
  const float g = gout[0] / stat[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) out[i] = v[i] * g;
}

extern "C" int crk_ce_fwd(const float* logits, int ldl, const long long* target, long long N, int C, int ignore_index,
                          float* out2, float* dlogits_unscaled, float* scratch, void* stream) {
  if (!logits || !target || !out2 || !scratch || C <= 0) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nb = loss_blocks(N);
  // (measured and dropped: the rows of a workgroup's 256 frames through LDS - one contiguous run in, one out - 18.8 us
  // against 7.9: the per-row passes over LDS cost more than the strided loads they replace)
  const bool rows8 = ldl == C && !(((uintptr_t)logits | (uintptr_t)dlogits_unscaled) & 7);
  if (rows8 && C == 14)
    hipLaunchKernelGGL(ce_partial_regs<7>, dim3(nb), dim3(256), 0, s, logits, target, (long)N, ignore_index, dlogits_unscaled, scratch);
  else if (rows8 && C == 12)
    hipLaunchKernelGGL(ce_partial_regs<6>, dim3(nb), dim3(256), 0, s, logits, target, (long)N, ignore_index, dlogits_unscaled, scratch);
  else
    hipLaunchKernelGGL(ce_partial, dim3(nb), dim3(256), 0, s, logits, ldl, target, (long)N, C, ignore_index,
                       dlogits_unscaled, scratch);
  hipLaunchKernelGGL(masked_loss_final, dim3(1), dim3(256), 0, s, scratch, nb, out2);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

extern "C" int crk_ce_bwd(float* dlogits_unscaled, long long N, int C, const float* stat2, const float* gout,
                          float* dlogits, void* stream) {
  const int nb = loss_blocks(N * C);
  hipLaunchKernelGGL(scale_by_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dlogits_unscaled, (long)N * C, gout,
                     stat2, dlogits);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// ------------------------------------------------------------------------------
// STFT-magnitude L1 loss along the FRAME axis of a feature matrix (loss.py:50-114).
// A "signal" is one feature dimension of one utterance: rows = B*D signals of length T.
// torch.stft(center=True, pad_mode="reflect", onesided) with a window of win_length
// taps zero-padded (centred) to n_fft: only win_length samples per frame are non-zero,
// so each (signal, frame, bin) is a direct win_length-point DFT from LDS twiddles.
//   mag = sqrt(max(re^2+im^2, 1e-7));
//   loss = (1-r) * mean|mag_x-mag_y| + r * mean|log mag_x - log mag_y|
// forward writes per-block partial sums; the backward kernel re-evaluates the DFT and
// scatters d loss / d x with atomics (a sample is touched by at most a few frames).
// ------------------------------------------------------------------------------
struct StftP {
  const float* x; const float* y; int ldx, ldy;  // [B*T, D]
  int B, T, D;
  int n_fft, hop, win;        // as torch.stft receives them
  int n_frames, n_bins;
  float logratio;
  const float* window;        // [win]
  float* part;                // partial sums
  // backward
  const float* gout; float scale; float* dx; int lddx;
};

__device__ __forceinline__ int reflect_idx(int s, int T) {
  if (s < 0) s = -s;
  if (s >= T) s = 2 * (T - 1) - s;
  return s;
}

template <bool BWD>
__global__ __launch_bounds__(256) void stft_loss_kernel(const StftP p) {
  extern __shared__ float tw[];  // cos [n_bins][win], sin [n_bins][win]
  __shared__ float sh[4];
  const int nb = p.n_bins, W = p.win;
  const int lpad = (p.n_fft - W) / 2;
  for (int i = threadIdx.x; i < nb * W; i += 256) {
    const int f = i / W, j = i - f * W;
    // exact phase reduction: angle = 2*pi*((f*(j+lpad)) mod n_fft)/n_fft
    const int ph = (int)(((long)f * (j + lpad)) % p.n_fft);
    const float a = 6.283185307179586476925f * (float)ph / (float)p.n_fft;
    tw[i] = cosf(a) * p.window[j];
    tw[nb * W + i] = sinf(a) * p.window[j];
  }
  __syncthreads();
  const long total = (long)p.B * p.D * p.n_frames * nb;
  float s = 0.f;
  const float g = BWD ? p.gout[0] * p.scale : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int f = (int)(i % nb);
    long r = i / nb;
    const int fr = (int)(r % p.n_frames);
    r /= p.n_frames;
    const int d = (int)(r % p.D);
    const int b = (int)(r / p.D);
    const int s0 = fr * p.hop - p.n_fft / 2 + lpad;  // first windowed sample (unpadded coords)
    const float* cw = tw + f * W;
    const float* sw = tw + nb * W + f * W;
    float rx = 0.f, ix = 0.f, ry = 0.f, iy = 0.f;
    for (int j = 0; j < W; j++) {
      const int t = reflect_idx(s0 + j, p.T);
      const long n = (long)b * p.T + t;
      const float xv = p.x[n * p.ldx + d], yv = p.y[n * p.ldy + d];
      rx += xv * cw[j]; ix -= xv * sw[j];
      ry += yv * cw[j]; iy -= yv * sw[j];
    }
    const float px = rx * rx + ix * ix, py = ry * ry + iy * iy;
    const float mx = sqrtf(fmaxf(px, 1e-7f)), my = sqrtf(fmaxf(py, 1e-7f));
    if (!BWD) {
      float v = (1.f - p.logratio) * fabsf(mx - my);
      if (p.logratio != 0.f) v += p.logratio * fabsf(logf(mx) - logf(my));
      s += v;
    } else {
      if (px > 1e-7f) {
        const float dm = mx - my;
        float c = (1.f - p.logratio) * (dm > 0.f ? 1.f : (dm < 0.f ? -1.f : 0.f));
        if (p.logratio != 0.f) {
          const float dl = logf(mx) - logf(my);
          c += p.logratio * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f)) / mx;
        }
        const float k = g * c / mx;  // d loss / d (re,im) = k * (re, im)
        if (k != 0.f) {
          for (int j = 0; j < W; j++) {
            const int t = reflect_idx(s0 + j, p.T);
            const long n = (long)b * p.T + t;
            atomicAdd(p.dx + n * p.lddx + d, k * (rx * cw[j] - ix * sw[j]));
          }
        }
      }
    }
  }
  if (!BWD) {
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) { p.part[2 * blockIdx.x] = s; p.part[2 * blockIdx.x + 1] = 0.f; }
  }
}

// Short-window variant (win_length <= 64, the step's configuration: 16 / 32 taps): one
// thread owns one (utterance, feature dim, frame).  Its W windowed samples of x and y sit
// in registers, every bin's DFT is 4*W FMAs against LDS twiddles (wave-uniform reads ->
// broadcast), and in the backward pass the W sample gradients are accumulated in
// registers over all bins and leave with ONE atomic each (frames only overlap through the
// reflect padding or when hop < win), instead of W atomics per (frame, bin).
// Consecutive threads take consecutive feature dims: global loads are coalesced rows.
#define STFT_BG 8
// workgroup `bid` of `nblk` working on resolution p; returns the workgroup's partial loss sum (forward)
// MODE 0: loss partial sums; 1: gradient for the upstream gradient gout[0]; 2: both at once, the gradient for an upstream
// gradient of 1 (the forward of a loss that will be differentiated: its backward is then a multiply, not a second DFT)
template <int W, int MODE>
__device__ __forceinline__ float stft_frame_body(const StftP& p, int bid, int nblk, float* tw, float* sh) {
  const int nb = p.n_bins;
  const int lpad = (p.n_fft - p.win) / 2;
  for (int i = threadIdx.x; i < nb * W; i += 256) {
    const int f = i / W, j = i - f * W;
    float c = 0.f, s = 0.f;
    if (j < p.win) {
      const int ph = (int)(((long)f * (j + lpad)) % p.n_fft);
      const float a = 6.283185307179586476925f * (float)ph / (float)p.n_fft;
      c = cosf(a) * p.window[j];
      s = sinf(a) * p.window[j];
    }
    tw[i] = c;
    tw[nb * W + i] = s;
  }
  __syncthreads();
  // a (batch, frame, feature dim) item is shared by STFT_BG lanes, each taking every STFT_BG-th
  // bin: 8x the parallelism of one thread per item (B * n_frames * D is only ~20 k items); lane =
  // bin group * 8 + item, so the partial window gradients meet through three xor shuffles
  const long total = (long)p.B * p.n_frames * p.D;
  constexpr bool BWD = MODE != 0, FWD = MODE != 1;
  const float g = MODE == 1 ? p.gout[0] * p.scale : (MODE == 2 ? p.scale : 0.f);
  float lsum = 0.f;
  const int lane = threadIdx.x & 63, bg = lane >> 3;
  const long wave0 = ((long)bid * 256 + threadIdx.x) >> 6, nwaves = (long)nblk * 4;
  for (long iw = wave0; iw * 8 < total; iw += nwaves) {
    const long i = iw * 8 + (lane & 7);
    const bool on = i < total;
    const int d = on ? (int)(i % p.D) : 0;
    long r = on ? i / p.D : 0;
    const int fr = (int)(r % p.n_frames);
    const int b = (int)(r / p.n_frames);
    const int s0 = fr * p.hop - p.n_fft / 2 + lpad;
    float xs[W], ys[W], gr[W];
#pragma unroll
    for (int j = 0; j < W; j++) {
      int t = reflect_idx(s0 + j, p.T);
      t = t < 0 ? 0 : (t >= p.T ? p.T - 1 : t);  // only reachable where the window weight is 0
      const long n = (long)b * p.T + t;
      xs[j] = p.x[n * p.ldx + d];
      ys[j] = p.y[n * p.ldy + d];
      if (BWD) gr[j] = 0.f;
    }
    for (int f = bg; f < nb; f += STFT_BG) {
      const float* cw = tw + f * W;
      const float* sw = tw + nb * W + f * W;
      float rx = 0.f, ix = 0.f, ry = 0.f, iy = 0.f;
#pragma unroll
      for (int j = 0; j < W; j++) {
        const float c = cw[j], s = sw[j];
        rx += xs[j] * c; ix -= xs[j] * s;
        ry += ys[j] * c; iy -= ys[j] * s;
      }
      const float px = rx * rx + ix * ix, py = ry * ry + iy * iy;
      const float mx = sqrtf(fmaxf(px, 1e-7f)), my = sqrtf(fmaxf(py, 1e-7f));
      if (FWD) {
        float v = (1.f - p.logratio) * fabsf(mx - my);
        if (p.logratio != 0.f) v += p.logratio * fabsf(logf(mx) - logf(my));
        if (on) lsum += v;
      }
      if (BWD && px > 1e-7f) {
        const float dm = mx - my;
        float c = (1.f - p.logratio) * (dm > 0.f ? 1.f : (dm < 0.f ? -1.f : 0.f));
        if (p.logratio != 0.f) {
          const float dl = logf(mx) - logf(my);
          c += p.logratio * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f)) / mx;
        }
        const float k = g * c / mx;
        const float kr = k * rx, ki = k * ix;
#pragma unroll
        for (int j = 0; j < W; j++) gr[j] += kr * cw[j] - ki * sw[j];
      }
    }
    if (BWD) {
#pragma unroll
      for (int j = 0; j < W; j++) {
        float v = gr[j];
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (j < p.win && bg == 0 && on) {
          const int t = reflect_idx(s0 + j, p.T);
          atomicAdd(p.dx + ((long)b * p.T + t) * p.lddx + d, v);
        }
      }
    }
  }
  if (FWD) lsum = block_sum_256(lsum, sh);
  return lsum;
}

template <int W, bool BWD>
__global__ __launch_bounds__(256) void stft_frame_kernel(const StftP p) {
  extern __shared__ float tw[];  // cos [n_bins][W], sin [n_bins][W] (zero beyond win)
  __shared__ float sh[4];
  const float lsum = stft_frame_body<W, BWD ? 1 : 0>(p, blockIdx.x, gridDim.x, tw, sh);
  if (!BWD && threadIdx.x == 0) { p.part[2 * blockIdx.x] = lsum; p.part[2 * blockIdx.x + 1] = 0.f; }
}

// Every resolution of the multi-resolution loss in ONE launch per direction: workgroups [bstart[r], bstart[r+1]) work on
// resolution r; the forward's last workgroup sums each resolution's partials and writes the loss.
struct StftMP {
  StftP r[LOSS_MAX_RES];
  int nres;
  int bstart[LOSS_MAX_RES + 1];
  float inv_count[LOSS_MAX_RES];
  float weight;
  float* out;
};

// WMAX: the longest window tile any resolution of the launch needs (the register allocation of the kernel is that of
// its widest body, so a launch of 16- and 32-tap windows must not carry the 64-tap one)
template <int MODE, int WMAX>
__global__ __launch_bounds__(256, 2) void stft_multi_kernel(const StftMP m) {
  extern __shared__ float tw[];
  __shared__ float sh[4];
  int r = 0;
  while (r + 1 < m.nres && (int)blockIdx.x >= m.bstart[r + 1]) r++;
  const int bid = blockIdx.x - m.bstart[r], nblk = m.bstart[r + 1] - m.bstart[r];
  const StftP& p = m.r[r];
  float lsum;
  if (WMAX == 16 || p.win <= 16) lsum = stft_frame_body<16, MODE>(p, bid, nblk, tw, sh);
  else if (WMAX == 32 || p.win <= 32) lsum = stft_frame_body<32, MODE>(p, bid, nblk, tw, sh);
  else lsum = stft_frame_body<64, MODE>(p, bid, nblk, tw, sh);
  if (MODE != 1 && threadIdx.x == 0) p.part[2 * bid] = lsum;
}

// one workgroup: every resolution's partials -> the loss (same arithmetic as one stft_final per resolution, accumulating)
__global__ __launch_bounds__(256) void stft_multi_final(const StftMP m) {
  __shared__ float sh[4];
  float total = 0.f;
  for (int q = 0; q < m.nres; q++) {
    float s = 0.f;
    const int nq = m.bstart[q + 1] - m.bstart[q];
    for (int i = threadIdx.x; i < nq; i += 256) s += m.r[q].part[2 * i];
    s = block_sum_256(s, sh);
    const float v = s * m.inv_count[q] * m.weight;
    total = q ? total + v : v;
  }
  if (threadIdx.x == 0) m.out[0] = total;
}

template <bool BWD>
static bool launch_stft_frames(const StftP& p, int* nblocks, hipStream_t s) {
  if (p.win > 64) return false;
  const int W = p.win <= 16 ? 16 : (p.win <= 32 ? 32 : 64);
  const size_t lds = (size_t)2 * p.n_bins * W * sizeof(float);
  if (lds > 60 * 1024) return false;
  const int nb = loss_blocks((long)p.B * p.n_frames * p.D * STFT_BG);
  *nblocks = nb;
  if (W == 16) hipLaunchKernelGGL((stft_frame_kernel<16, BWD>), dim3(nb), dim3(256), lds, s, p);
  else if (W == 32) hipLaunchKernelGGL((stft_frame_kernel<32, BWD>), dim3(nb), dim3(256), lds, s, p);
  else hipLaunchKernelGGL((stft_frame_kernel<64, BWD>), dim3(nb), dim3(256), lds, s, p);
  return true;
}

__global__ __launch_bounds__(256) void stft_final(const float* __restrict__ part, int nblocks, float inv_count,
                                                  float weight, float* __restrict__ out, int accumulate) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += part[2 * i];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) {
    const float v = s * inv_count * weight;
    out[0] = accumulate ? out[0] + v : v;
  }
}

// one resolution; the caller loops over resolutions with weight = 1/n_resolutions.
extern "C" int crk_stft_loss_fwd(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int n_fft,
                                 int hop_length, int win_length, const float* window, float logratio, float weight,
                                 int accumulate, float* out1, float* scratch, void* stream) {
  if (!x || !y || !window || !out1 || !scratch || win_length > n_fft || n_fft / 2 >= T) return CRK_ERR_ARG;
  StftP p{};
  p.x = x; p.y = y; p.ldx = ldx; p.ldy = ldy; p.B = B; p.T = T; p.D = D;
  p.n_fft = n_fft; p.hop = hop_length; p.win = win_length;
  p.n_frames = 1 + T / hop_length;  // center=True: (T + 2*(n_fft/2) - n_fft)/hop + 1
  p.n_bins = n_fft / 2 + 1;
  p.logratio = logratio; p.window = window; p.part = scratch;
  const long total = (long)B * D * p.n_frames * p.n_bins;
  int nb = loss_blocks(total);
  hipStream_t s = (hipStream_t)stream;
  if (!launch_stft_frames<false>(p, &nb, s)) {
    const size_t lds = (size_t)2 * p.n_bins * win_length * sizeof(float);
    if (lds > 60 * 1024) return CRK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(stft_loss_kernel<false>, dim3(nb), dim3(256), lds, s, p);
  }
  hipLaunchKernelGGL(stft_final, dim3(1), dim3(256), 0, s, scratch, nb, 1.0f / (float)total, weight, out1, accumulate);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// dx must be zero-initialised by the caller before the first resolution.
extern "C" int crk_stft_loss_bwd(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int n_fft,
                                 int hop_length, int win_length, const float* window, float logratio, float weight,
                                 const float* gout, float* dx, int lddx, void* stream) {
  if (!x || !y || !window || !gout || !dx || win_length > n_fft) return CRK_ERR_ARG;
  StftP p{};
  p.x = x; p.y = y; p.ldx = ldx; p.ldy = ldy; p.B = B; p.T = T; p.D = D;
  p.n_fft = n_fft; p.hop = hop_length; p.win = win_length;
  p.n_frames = 1 + T / hop_length;
  p.n_bins = n_fft / 2 + 1;
  p.logratio = logratio; p.window = window;
  const long total = (long)B * D * p.n_frames * p.n_bins;
  p.gout = gout; p.scale = weight / (float)total; p.dx = dx; p.lddx = lddx;
  int nb = loss_blocks(total);
  if (!launch_stft_frames<true>(p, &nb, (hipStream_t)stream)) {
    const size_t lds = (size_t)2 * p.n_bins * win_length * sizeof(float);
    if (lds > 60 * 1024) return CRK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(stft_loss_kernel<true>, dim3(nb), dim3(256), lds, (hipStream_t)stream, p);
  }
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

static int stft_fill(StftP& p, const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int n_fft, int hop,
                     int win, const float* window, float logratio) {
  if (!window || win > n_fft || n_fft / 2 >= T || win > 64) return CRK_ERR_UNSUPPORTED;
  p.x = x; p.y = y; p.ldx = ldx; p.ldy = ldy; p.B = B; p.T = T; p.D = D;
  p.n_fft = n_fft; p.hop = hop; p.win = win;
  p.n_frames = 1 + T / hop;
  p.n_bins = n_fft / 2 + 1;
  p.logratio = logratio; p.window = window;
  return CRK_OK;
}

template <int MODE>
static int stft_multi_launch(StftMP& m, hipStream_t s) {
  size_t lds = 0;
  int wmax = 16;
  m.bstart[0] = 0;
  for (int r = 0; r < m.nres; r++) {
    const StftP& p = m.r[r];
    const int W = p.win <= 16 ? 16 : (p.win <= 32 ? 32 : 64);
    if (W > wmax) wmax = W;
    const size_t l = (size_t)2 * p.n_bins * W * sizeof(float);
    if (l > lds) lds = l;
    m.bstart[r + 1] = m.bstart[r] + loss_blocks((long)p.B * p.n_frames * p.D * STFT_BG);
  }
  if (lds > 60 * 1024) return CRK_ERR_UNSUPPORTED;
  if (wmax == 16) hipLaunchKernelGGL((stft_multi_kernel<MODE, 16>), dim3(m.bstart[m.nres]), dim3(256), lds, s, m);
  else if (wmax == 32) hipLaunchKernelGGL((stft_multi_kernel<MODE, 32>), dim3(m.bstart[m.nres]), dim3(256), lds, s, m);
  else hipLaunchKernelGGL((stft_multi_kernel<MODE, 64>), dim3(m.bstart[m.nres]), dim3(256), lds, s, m);
  if (MODE != 1) hipLaunchKernelGGL(stft_multi_final, dim3(1), dim3(256), 0, s, m);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// All resolutions at once (every win_length <= 64, at most LOSS_MAX_RES of them); CRK_ERR_UNSUPPORTED otherwise: the
// caller then loops over crk_stft_loss_fwd / _bwd.  out1[0] = mean over resolutions of the per-resolution loss.
extern "C" int crk_stft_loss_multi_fwd(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int nres,
                                       const int* n_fft, const int* hop_length, const int* win_length,
                                       const float* const* windows, float logratio, float* out1, float* scratch,
                                       void* stream) {
  if (!x || !y || !out1 || !scratch || !n_fft || !hop_length || !win_length || !windows || nres < 1) return CRK_ERR_ARG;
  if (nres > LOSS_MAX_RES) return CRK_ERR_UNSUPPORTED;
  StftMP m{};
  m.nres = nres; m.weight = 1.0f / (float)nres; m.out = out1;
  for (int r = 0; r < nres; r++) {
    const int rc = stft_fill(m.r[r], x, ldx, y, ldy, B, T, D, n_fft[r], hop_length[r], win_length[r], windows[r], logratio);
    if (rc != CRK_OK) return rc;
    m.r[r].part = scratch + (size_t)r * 2 * LOSS_MAX_BLOCKS;
    m.inv_count[r] = 1.0f / (float)((long)B * D * m.r[r].n_frames * m.r[r].n_bins);
  }
  return stft_multi_launch<0>(m, (hipStream_t)stream);
}

// Loss AND its gradient in one pass (the forward of a loss that is going to be differentiated): out1[0] as above,
// dx_unit += d out1 / d x (zero-initialised by the caller).  The backward is then dx = upstream gradient * dx_unit
// instead of a second evaluation of every DFT.
extern "C" int crk_stft_loss_multi_fwd_grad(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int nres,
                                            const int* n_fft, const int* hop_length, const int* win_length,
                                            const float* const* windows, float logratio, float* out1, float* dx_unit,
                                            int lddx, float* scratch, void* stream) {
  if (!x || !y || !out1 || !dx_unit || !scratch || !n_fft || !hop_length || !win_length || !windows || nres < 1) return CRK_ERR_ARG;
  if (nres > LOSS_MAX_RES) return CRK_ERR_UNSUPPORTED;
  StftMP m{};
  m.nres = nres; m.weight = 1.0f / (float)nres; m.out = out1;
  for (int r = 0; r < nres; r++) {
    const int rc = stft_fill(m.r[r], x, ldx, y, ldy, B, T, D, n_fft[r], hop_length[r], win_length[r], windows[r], logratio);
    if (rc != CRK_OK) return rc;
    const long total = (long)B * D * m.r[r].n_frames * m.r[r].n_bins;
    m.r[r].part = scratch + (size_t)r * 2 * LOSS_MAX_BLOCKS;
    m.inv_count[r] = 1.0f / (float)total;
    m.r[r].scale = (1.0f / (float)nres) / (float)total; m.r[r].dx = dx_unit; m.r[r].lddx = lddx;
  }
  return stft_multi_launch<2>(m, (hipStream_t)stream);
}

// dx must be zero-initialised by the caller (or hold a gradient this one is to be added to).
extern "C" int crk_stft_loss_multi_bwd(const float* x, int ldx, const float* y, int ldy, int B, int T, int D, int nres,
                                       const int* n_fft, const int* hop_length, const int* win_length,
                                       const float* const* windows, float logratio, const float* gout, float* dx,
                                       int lddx, void* stream) {
  if (!x || !y || !gout || !dx || !n_fft || !hop_length || !win_length || !windows || nres < 1) return CRK_ERR_ARG;
  if (nres > LOSS_MAX_RES) return CRK_ERR_UNSUPPORTED;
  StftMP m{};
  m.nres = nres;
  for (int r = 0; r < nres; r++) {
    const int rc = stft_fill(m.r[r], x, ldx, y, ldy, B, T, D, n_fft[r], hop_length[r], win_length[r], windows[r], logratio);
    if (rc != CRK_OK) return rc;
    const long total = (long)B * D * m.r[r].n_frames * m.r[r].n_bins;
    m.r[r].gout = gout; m.r[r].scale = (1.0f / (float)nres) / (float)total; m.r[r].dx = dx; m.r[r].lddx = lddx;
  }
  return stft_multi_launch<1>(m, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------
// Reconstruction losses of the decoded features as ONE launch per direction (trainer_vqvae.py:215-225: L1, MSE and
// the multi-resolution STFT loss are taken on the same pair): workgroups [r.blk0, r.blk0 + r.nblk) evaluate the DFTs of
// resolution r, the workgroups behind them sum |x-y| and (x-y)^2 over the masked frames.
//
// STFT part.  An item is one (utterance, frame, feature dim); its W window samples of x and y sit in the registers of
// one lane (consecutive lanes = consecutive feature dims: coalesced rows), the four waves of a workgroup share the 64
// items and take every fourth bin each.  Twiddles (cos * window, -sin * window) come from a table built once per
// (n_fft, window) by crk_stft_twiddles, copied to LDS and read with wave-uniform 16-byte reads; x and y ride in the
// two halves of packed fp32 FMAs.  The gradient of a frame for an upstream gradient of 1 is summed over a wave's bins
// in registers, over the four waves through LDS in a fixed order, and written to a COMPACT buffer
// [utterance][frame][span][feature dim], span = win + 3 samples starting at frame * hop - lo (one sample either side
// of the window and one more for odd windows: the reflect padding of the first / last frame lands there).  Spans of
// different frames are disjoint when hop >= win + 3 - the only geometry this path takes (quirk Q1 makes the step's
// resolutions (64, 64, 16) and (128, 128, 32)) - so there are no atomics and no zero-filled dense buffer, and the
// result is deterministic.  recon_bwd_kernel gathers: sample t of resolution r belongs to frame (t + lo) / hop.
// ------------------------------------------------------------------------------
#define RC_SPAN_EXTRA 3
#define RC_NW 4  // waves per workgroup: the bins of a frame are dealt to them (8 - half the run of FMAs per wave - measured
                 // slower at the step's shape, 35 against 32 us: twice the redundant window loads)
__device__ __forceinline__ float block_sum_nw(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  float t = sh[0];
#pragma unroll
  for (int w = 1; w < RC_NW; w++) t += sh[w];
  return t;
}
struct ReconRes {
  const float* tw;   // [n_bins][2][W]
  float* gc;         // compact gradient or nullptr
  int n_fft, hop, win, nf, nb, W, lo;
  int blk0, nblk, ngroups;
  float inv_count, scale;
};
struct ReconP {
  const float* x; const float* y; const unsigned char* mask;
  int ldx, ldy, B, T, D, nres;
  float logratio, weight;
  ReconRes r[LOSS_MAX_RES];
  int el_blk0, el_nblk, el_vec4;
  float* part;       // [LOSS_MAX_RES][LOSS_MAX_BLOCKS] STFT partials, then [el_nblk][3]
  float* out;        // {L1 mean, count, MSE mean, count, STFT loss}
};
typedef float rc_f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void stft_twiddle_kernel(int n_fft, int win, int W, int nb, const float* __restrict__ window,
                                                           float* __restrict__ tw) {
  const int lpad = (n_fft - win) / 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nb * W; i += gridDim.x * 256) {
    const int f = i / W, j = i - f * W;
    float c = 0.f, s = 0.f;
    if (j < win) {
      const int ph = (int)(((long)f * (j + lpad)) % n_fft);  // exact phase reduction
      const double a = 6.283185307179586476925 * (double)ph / (double)n_fft;
      c = (float)(cos(a) * (double)window[j]);
      s = (float)(-sin(a) * (double)window[j]);
    }
    tw[(f * 2) * W + j] = c;
    tw[(f * 2 + 1) * W + j] = s;
  }
}

// the RC_NW waves' partial gradients of tap j, summed in wave order
template <int W>
__device__ __forceinline__ float rc_psum(const float* pg, int j, int lane) {
  float t = pg[j * 64 + lane];
#pragma unroll
  for (int w = 1; w < RC_NW; w++) t += pg[(w * W + j) * 64 + lane];
  return t;
}
template <int W, bool GRAD>
__device__ __forceinline__ void recon_stft_body(const ReconP& p, const ReconRes& r, int ri, int bid, float* lds, float* sh) {
  // Twiddles: scalar loads from the table (uniform per wave) straight into the packed FMAs' scalar operand.  (Read from an
  // LDS copy with wave-uniform 16-byte reads - the first version - every read still moves 1 KB to the lanes: 16 reads per
  // bin x 8 waves saturated the CU's LDS return path, 20 us for the 32-tap resolution against 6 us of FMAs.)
  typedef const float __attribute__((address_space(4))) rc_cf;
  rc_cf* tw = (rc_cf*)r.tw;            // [nb][2][W]
  float* pg = lds;                      // [RC_NW][W][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long total = (long)p.B * r.nf * p.D;
  const int span = r.win + RC_SPAN_EXTRA;
  const float lr = p.logratio, olr = 1.f - p.logratio;
  float lsum = 0.f;
  for (int g = bid; g < r.ngroups; g += r.nblk) {
    const long i = (long)g * 64 + lane;
    const bool on = i < total;
    const int d = on ? (int)(i % p.D) : 0;
    const long q = on ? i / p.D : 0;
    const int fr = (int)(q % r.nf), b = (int)(q / r.nf);
    const int s0 = fr * r.hop - r.lo + 1;  // first windowed sample (unpadded coordinates)
    rc_f2 xy[W];
#pragma unroll
    for (int j = 0; j < W; j++) {
      int t = reflect_idx(s0 + j, p.T);
      t = t < 0 ? 0 : (t >= p.T ? p.T - 1 : t);  // only reachable where the window weight is 0
      const long n = (long)b * p.T + t;
      xy[j].x = p.x[n * p.ldx + d];
      xy[j].y = p.y[n * p.ldy + d];
    }
    rc_f2 gr[W / 2];
    if (GRAD) {
#pragma unroll
      for (int j = 0; j < W / 2; j++) gr[j] = (rc_f2){0.f, 0.f};
    }
    for (int f = wave; f < r.nb; f += RC_NW) {
      rc_cf* cw = tw + f * 2 * W;
      rc_cf* nw = cw + W;
      float cs[W], ns[W];
#pragma unroll
      for (int j = 0; j < W; j++) { cs[j] = cw[j]; ns[j] = nw[j]; }
      rc_f2 re = {0.f, 0.f}, im = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < W; j++) { re += xy[j] * cs[j]; im += xy[j] * ns[j]; }
      const rc_f2 pw = re * re + im * im;
      const float mx = sqrtf(fmaxf(pw.x, 1e-7f)), my = sqrtf(fmaxf(pw.y, 1e-7f));
      float v = olr * fabsf(mx - my);
      float dl = 0.f;
      if (lr != 0.f) { dl = logf(mx) - logf(my); v += lr * fabsf(dl); }
      if (on) lsum += v;
      if (GRAD) {
        const float dm = mx - my;
        float c = olr * (dm > 0.f ? 1.f : (dm < 0.f ? -1.f : 0.f));
        if (lr != 0.f) c += lr * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f)) / mx;
        const float k = pw.x > 1e-7f ? r.scale * c / mx : 0.f;  // d loss / d (re, im) = k * (re, im)
        const float kr = k * re.x, ki = k * im.x;
#pragma unroll
        for (int j = 0; j < W; j += 2) gr[j / 2] += (rc_f2){cs[j], cs[j + 1]} * kr + (rc_f2){ns[j], ns[j + 1]} * ki;
      }
    }
    if (GRAD) {
      __syncthreads();  // the previous group's sums have been read
#pragma unroll
      for (int j = 0; j < W / 2; j++) {
        pg[(wave * W + 2 * j) * 64 + lane] = gr[j].x;
        pg[(wave * W + 2 * j + 1) * 64 + lane] = gr[j].y;
      }
      __syncthreads();
      float* go = r.gc + ((size_t)((long)b * r.nf + fr) * span) * p.D + d;
      for (int off = wave; off < span; off += RC_NW) {
        const int t = s0 - 1 + off;
        float v = 0.f;
        int j = off - 1;                         // the sample itself
        if (j >= 0 && j < r.win && t >= 0 && t < p.T)
          v += rc_psum<W>(pg, j, lane);
        j = -t - s0;                             // reflected at the start: window sample -t
        if (t > 0 && j >= 0 && j < r.win)
          v += rc_psum<W>(pg, j, lane);
        j = 2 * (p.T - 1) - t - s0;              // reflected at the end: window sample 2 (T - 1) - t >= T
        if (t >= 0 && t <= p.T - 2 && j >= 0 && j < r.win)
          v += rc_psum<W>(pg, j, lane);
        if (on) go[(size_t)off * p.D] = v;
      }
    }
  }
  lsum = block_sum_nw(lsum, sh);
  if (tid == 0) p.part[ri * LOSS_MAX_BLOCKS + bid] = lsum;
}

template <int WMAX, bool GRAD>
__global__ __launch_bounds__(RC_NW * 64) void recon_fwd_kernel(const ReconP p) {
  extern __shared__ float lds[];
  __shared__ float sh[RC_NW];
  const int bx = blockIdx.x;
  if (bx >= p.el_blk0) {  // |x-y| and (x-y)^2 over the masked frames (the loops of masked_loss_partial4<2> / _both_partial)
    const int bid = bx - p.el_blk0;
    float s1 = 0.f, s2 = 0.f, c = 0.f;
    if (p.el_vec4) {
      const int D4 = p.D / 4;
      const long total = (long)p.B * p.T * D4;
      for (long i = (long)bid * (RC_NW * 64) + threadIdx.x; i < total; i += (long)p.el_nblk * (RC_NW * 64)) {
        const long n = i / D4;
        const int d = (int)(i - n * D4) * 4;
        if (p.mask && !p.mask[n]) continue;
        const float4 xv = *reinterpret_cast<const float4*>(p.x + n * p.ldx + d);
        const float4 yv = *reinterpret_cast<const float4*>(p.y + n * p.ldy + d);
        const float df[4] = {xv.x - yv.x, xv.y - yv.y, xv.z - yv.z, xv.w - yv.w};
#pragma unroll
        for (int j = 0; j < 4; j++) { s1 += fabsf(df[j]); s2 += df[j] * df[j]; }
        c += 4.f;
      }
    } else {
      const long total = (long)p.B * p.T * p.D;
      for (long i = (long)bid * (RC_NW * 64) + threadIdx.x; i < total; i += (long)p.el_nblk * (RC_NW * 64)) {
        const long n = i / p.D;
        const int d = (int)(i - n * p.D);
        if (p.mask && !p.mask[n]) continue;
        const float df = p.x[n * p.ldx + d] - p.y[n * p.ldy + d];
        s1 += fabsf(df); s2 += df * df; c += 1.f;
      }
    }
    s1 = block_sum_nw(s1, sh);
    s2 = block_sum_nw(s2, sh);
    c = block_sum_nw(c, sh);
    if (threadIdx.x == 0) {
      float* q = p.part + LOSS_MAX_RES * LOSS_MAX_BLOCKS + 3 * bid;
      q[0] = s1; q[1] = s2; q[2] = c;
    }
    return;
  }
  // (statically indexed copies: a dynamic index into the kernel argument would move the whole struct to scratch)
  int ri = 0;
  ReconRes r = p.r[0];
#pragma unroll
  for (int k = 1; k < LOSS_MAX_RES; k++)
    if (k < p.nres && bx >= p.r[k].blk0) { ri = k; r = p.r[k]; }
  const int bid = bx - r.blk0;
  if (WMAX == 16 || r.W == 16) recon_stft_body<16, GRAD>(p, r, ri, bid, lds, sh);
  else if (WMAX == 32 || r.W == 32) recon_stft_body<32, GRAD>(p, r, ri, bid, lds, sh);
  else recon_stft_body<64, GRAD>(p, r, ri, bid, lds, sh);
}

__global__ __launch_bounds__(256) void recon_final_kernel(const ReconP p) {
  __shared__ float sh[4];
  float s1 = 0.f, s2 = 0.f, c = 0.f;
  const float* q = p.part + LOSS_MAX_RES * LOSS_MAX_BLOCKS;
  for (int i = threadIdx.x; i < p.el_nblk; i += 256) { s1 += q[3 * i]; s2 += q[3 * i + 1]; c += q[3 * i + 2]; }
  s1 = block_sum_256(s1, sh);
  s2 = block_sum_256(s2, sh);
  c = block_sum_256(c, sh);
  float total = 0.f;
#pragma unroll
  for (int k = 0; k < LOSS_MAX_RES; k++) {
    if (k < p.nres) {  // (uniform)
      float s = 0.f;
      for (int i = threadIdx.x; i < p.r[k].nblk; i += 256) s += p.part[k * LOSS_MAX_BLOCKS + i];
      s = block_sum_256(s, sh);
      const float v = s * p.r[k].inv_count * p.weight;
      total = k ? total + v : v;
    }
  }
  if (threadIdx.x == 0) { p.out[0] = s1 / c; p.out[1] = c; p.out[2] = s2 / c; p.out[3] = c; p.out[4] = total; }
}

static int rc_tile(int win) { return win <= 16 ? 16 : (win <= 32 ? 32 : 64); }
// the geometry the fused path takes; everything else goes through crk_masked_loss_both_fwd + crk_stft_loss_multi_*
static bool recon_res_ok(int T, int n_fft, int hop, int win) {
  if (win < 1 || win > 64 || win > n_fft || n_fft / 2 >= T || hop < win + RC_SPAN_EXTRA) return false;
  return (long)(n_fft / 2 + 1) * rc_tile(win) <= 8192;
}
extern "C" int crk_recon_supported(int T, int nres, const int* n_fft, const int* hop_length, const int* win_length) {
  if (nres < 1 || nres > LOSS_MAX_RES || !n_fft || !hop_length || !win_length) return 0;
  for (int r = 0; r < nres; r++)
    if (!recon_res_ok(T, n_fft[r], hop_length[r], win_length[r])) return 0;
  return 1;
}
extern "C" long long crk_stft_twiddle_floats(int n_fft, int win_length) {
  return (long long)(n_fft / 2 + 1) * 2 * rc_tile(win_length);
}
extern "C" int crk_stft_twiddles(int n_fft, int win_length, const float* window, float* table, void* stream) {
  if (!window || !table || win_length < 1 || win_length > 64 || win_length > n_fft) return CRK_ERR_ARG;
  const int W = rc_tile(win_length), nb = n_fft / 2 + 1;
  hipLaunchKernelGGL(stft_twiddle_kernel, dim3((nb * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_fft, win_length, W,
                     nb, window, table);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
extern "C" long long crk_recon_grad_floats(int B, int T, int D, int nres, const int* hop_length, const int* win_length) {
  long long n = 0;
  for (int r = 0; r < nres; r++) n += (long long)B * (1 + T / hop_length[r]) * (win_length[r] + RC_SPAN_EXTRA) * D;
  return n;
}

// out5 = {L1 mean, count, MSE mean, count, STFT loss}; grad (crk_recon_grad_floats floats, or NULL when nothing will be
// differentiated) receives the STFT loss's gradient for an upstream gradient of 1 in the compact layout above;
// tables[r] from crk_stft_twiddles.  CRK_ERR_UNSUPPORTED outside crk_recon_supported.
extern "C" int crk_recon_loss_fwd(const float* x, int ldx, const float* y, int ldy, const unsigned char* mask, int B, int T,
                                  int D, int nres, const int* n_fft, const int* hop_length, const int* win_length,
                                  const float* const* tables, float logratio, float* out5, float* grad, float* scratch,
                                  void* stream) {
  if (!x || !y || !out5 || !scratch || !tables || B < 1 || T < 1 || D < 1) return CRK_ERR_ARG;
  if (!crk_recon_supported(T, nres, n_fft, hop_length, win_length)) return CRK_ERR_UNSUPPORTED;
  ReconP p;
  p.x = x; p.y = y; p.mask = mask; p.ldx = ldx; p.ldy = ldy; p.B = B; p.T = T; p.D = D; p.nres = nres;
  p.logratio = logratio; p.weight = 1.0f / (float)nres; p.part = scratch; p.out = out5;
  int blk = 0, wmax = 16;
  size_t lds = 0;
  float* g = grad;
  for (int r = 0; r < nres; r++) {
    ReconRes& q = p.r[r];
    if (!tables[r]) return CRK_ERR_ARG;
    q.tw = tables[r]; q.n_fft = n_fft[r]; q.hop = hop_length[r]; q.win = win_length[r];
    q.nf = 1 + T / q.hop; q.nb = q.n_fft / 2 + 1; q.W = rc_tile(q.win);
    q.lo = q.n_fft / 2 - (q.n_fft - q.win) / 2 + 1;
    const long total = (long)B * D * q.nf * q.nb;
    q.inv_count = 1.0f / (float)total;
    q.scale = p.weight / (float)total;
    const long groups = ((long)B * q.nf * D + 63) / 64;
    q.ngroups = (int)groups;
    q.nblk = (int)(groups > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : groups);
    q.blk0 = blk; blk += q.nblk;
    q.gc = g;
    if (g) g += (size_t)B * q.nf * (q.win + RC_SPAN_EXTRA) * D;
    if (q.W > wmax) wmax = q.W;
    const size_t need = (size_t)RC_NW * q.W * 64 * sizeof(float);
    if (need > lds) lds = need;
  }
  p.el_blk0 = blk;
  p.el_vec4 = loss_vec4_ok(x, y, nullptr, D, ldx, ldy, 0) ? 1 : 0;
  {
    const long el = p.el_vec4 ? (long)B * T * (D / 4) : (long)B * T * D;
    const long nb = (el + RC_NW * 64 - 1) / (RC_NW * 64);
    p.el_nblk = (int)(nb > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : (nb < 1 ? 1 : nb));
  }
  blk += p.el_nblk;
  hipStream_t s = (hipStream_t)stream;
#define RC_LAUNCH(WM, GR)                                                                                           \
  do {                                                                                                               \
    if (lds > 48 * 1024)                                                                                             \
      (void)hipFuncSetAttribute((const void*)recon_fwd_kernel<WM, GR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((recon_fwd_kernel<WM, GR>), dim3(blk), dim3(RC_NW * 64), lds, s, p);                                 \
  } while (0)
  if (grad) {
    if (wmax == 16) RC_LAUNCH(16, true); else if (wmax == 32) RC_LAUNCH(32, true); else RC_LAUNCH(64, true);
  } else {
    if (wmax == 16) RC_LAUNCH(16, false); else if (wmax == 32) RC_LAUNCH(32, false); else RC_LAUNCH(64, false);
  }
#undef RC_LAUNCH
  hipLaunchKernelGGL(recon_final_kernel, dim3(1), dim3(256), 0, s, p);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

struct ReconBRes { const float* gc; int hop, lo, span, nf; };
struct ReconBP {
  const float* x; const float* y; const unsigned char* mask;
  int ldx, ldy, T, D, nres;
  long N;
  const float* stat; const float* g1; const float* g2; const float* g3;
  float* dx; int lddx;
  ReconBRes r[LOSS_MAX_RES];
};
// dx = g3 * (gathered STFT gradient) + g1 * d L1 + g2 * d MSE, V consecutive feature dims per thread
template <int V>
__global__ __launch_bounds__(256) void recon_bwd_kernel(const ReconBP p) {
  const float g1 = p.g1 ? p.g1[0] / p.stat[1] : 0.f;
  const float g2 = p.g2 ? p.g2[0] / p.stat[3] : 0.f;
  const float g3 = p.g3 ? p.g3[0] : 0.f;
  const int DV = p.D / V;
  const long total = p.N * DV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / DV;
    const int d = (int)(i - n * DV) * V;
    const int b = (int)(n / p.T), t = (int)(n - (long)b * p.T);
    float v[V];
#pragma unroll
    for (int j = 0; j < V; j++) v[j] = 0.f;
    if (p.g3) {
#pragma unroll
      for (int k = 0; k < LOSS_MAX_RES; k++) {
        if (k >= p.nres) break;
        const ReconBRes& r = p.r[k];
        const int u = t + r.lo, fr = u / r.hop, off = u - fr * r.hop;
        if (off < r.span && fr < r.nf) {
          const float* gp = r.gc + ((size_t)((long)b * r.nf + fr) * r.span + off) * p.D + d;
          if (V == 4) {
            const float4 a = *reinterpret_cast<const float4*>(gp);
            v[0] += a.x * g3; v[1 % V] += a.y * g3; v[2 % V] += a.z * g3; v[3 % V] += a.w * g3;
          } else {
            v[0] += gp[0] * g3;
          }
        }
      }
    }
    if (!p.mask || p.mask[n]) {
      float xv[V], yv[V];
      if (V == 4) {
        const float4 a = *reinterpret_cast<const float4*>(p.x + n * p.ldx + d);
        const float4 c = *reinterpret_cast<const float4*>(p.y + n * p.ldy + d);
        xv[0] = a.x; xv[1 % V] = a.y; xv[2 % V] = a.z; xv[3 % V] = a.w;
        yv[0] = c.x; yv[1 % V] = c.y; yv[2 % V] = c.z; yv[3 % V] = c.w;
      } else {
        xv[0] = p.x[n * p.ldx + d]; yv[0] = p.y[n * p.ldy + d];
      }
#pragma unroll
      for (int j = 0; j < V; j++) {
        const float df = xv[j] - yv[j];
        if (p.g1) v[j] += df > 0.f ? g1 : (df < 0.f ? -g1 : 0.f);
        if (p.g2) v[j] += 2.f * df * g2;
      }
    }
    if (V == 4) *reinterpret_cast<float4*>(p.dx + n * p.lddx + d) = make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]);
    else p.dx[n * p.lddx + d] = v[0];
  }
}

// dx = g1[0] * d L1 + g2[0] * d MSE + g3[0] * d STFT (each gradient pointer may be NULL: term not differentiated);
// out5 and grad as crk_recon_loss_fwd left them
extern "C" int crk_recon_loss_bwd(const float* x, int ldx, const float* y, int ldy, const unsigned char* mask, int B, int T,
                                  int D, int nres, const int* n_fft, const int* hop_length, const int* win_length,
                                  const float* out5, const float* grad, const float* g1, const float* g2, const float* g3,
                                  float* dx, int lddx, void* stream) {
  if (!x || !y || !out5 || !dx || B < 1 || T < 1 || D < 1) return CRK_ERR_ARG;
  if (g3 && (!grad || !crk_recon_supported(T, nres, n_fft, hop_length, win_length))) return CRK_ERR_ARG;
  ReconBP p;
  p.x = x; p.y = y; p.mask = mask; p.ldx = ldx; p.ldy = ldy; p.T = T; p.D = D; p.nres = g3 ? nres : 0;
  p.N = (long)B * T; p.stat = out5; p.g1 = g1; p.g2 = g2; p.g3 = g3; p.dx = dx; p.lddx = lddx;
  const float* g = grad;
  for (int r = 0; r < p.nres; r++) {
    ReconBRes& q = p.r[r];
    q.hop = hop_length[r]; q.span = win_length[r] + RC_SPAN_EXTRA; q.nf = 1 + T / q.hop;
    q.lo = n_fft[r] / 2 - (n_fft[r] - win_length[r]) / 2 + 1;
    q.gc = g;
    g += (size_t)B * q.nf * q.span * D;
  }
  const bool v4 = !((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dx) | ((uintptr_t)grad)) & 15) && !((D | ldx | ldy | lddx) & 3);
  if (v4) hipLaunchKernelGGL(recon_bwd_kernel<4>, dim3(loss_blocks(p.N * (D / 4))), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(recon_bwd_kernel<1>, dim3(loss_blocks(p.N * D)), dim3(256), 0, (hipStream_t)stream, p);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// ------------------------------------------------------------------------------
// Weighted sum of up to 16 device scalars, and its backward (the trainers' loss totals: "loss[G] += alpha * term",
// basetrainer.py:200-206 / trainer_vqvae.py:210-239, as one launch per direction instead of stack + multiply + sum).
// ------------------------------------------------------------------------------
#define WS_MAX 16
struct WsumP { const float* t[WS_MAX]; float w[WS_MAX]; int n; float c; float* out; const float* g; };
__global__ void weighted_sum_kernel(const WsumP p) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < WS_MAX; i++)
    if (i < p.n) s += p.w[i] * p.t[i][0];
  p.out[0] = s + p.c;
}
__global__ void weighted_sum_bwd_kernel(const WsumP p) {
  const float g = p.g[0];
#pragma unroll
  for (int i = 0; i < WS_MAX; i++)
    if (i < p.n) p.out[i] = p.w[i] * g;
}
// out[0] = sum_i weights[i] * terms[i][0] + constant (terms: device pointers, weights: host values); n <= 16
extern "C" int crk_weighted_sum(int n, const float* const* terms, const float* weights, float constant, float* out, void* stream) {
  if (n < 1 || n > WS_MAX || !terms || !weights || !out) return CRK_ERR_ARG;
  WsumP p{};
  for (int i = 0; i < n; i++) { if (!terms[i]) return CRK_ERR_ARG; p.t[i] = terms[i]; p.w[i] = weights[i]; }
  p.n = n; p.c = constant; p.out = out;
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
// grads[i] = weights[i] * gout[0], i < n
extern "C" int crk_weighted_sum_bwd(int n, const float* weights, const float* gout, float* grads, void* stream) {
  if (n < 1 || n > WS_MAX || !weights || !gout || !grads) return CRK_ERR_ARG;
  WsumP p{};
  for (int i = 0; i < n; i++) p.w[i] = weights[i];
  p.n = n; p.g = gout; p.out = grads;
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// ------------------------------------------------------------------------------
// Adam over one flat fp32 parameter block (torch.optim.Adam defaults: betas (0.9,
// 0.999), eps 1e-8, no weight decay, no amsgrad; crank/net/trainer/utils.py:40-58).
// lr and the step counter live in device memory (graph-capturable, no host sync):
// step_dev[0] = step count (float, advanced by a one-thread launch behind the update), lr_dev[0] = lr.
// ------------------------------------------------------------------------------
template <bool CLEAR>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   const float* __restrict__ lr_dev, float* __restrict__ step_dev,
                                                   float beta1, float beta2, float eps) {
  const AdamCoef c = adam_coef(lr_dev, step_dev, beta1, beta2, eps);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) adam_elem<CLEAR>(p, g, m, v, i, c);
}

__global__ void adam_bump_kernel(float* step_dev) { step_dev[0] += 1.f; }

extern "C" int crk_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                             const float* lr_dev, float* step_dev, float beta1, float beta2, float eps, int clear_grads,
                             void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (clear_grads & 1)
    hipLaunchKernelGGL(adam_kernel<true>, dim3((int)b), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (long)n, lr_dev,
                       step_dev, beta1, beta2, eps);
  else
    hipLaunchKernelGGL(adam_kernel<false>, dim3((int)b), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (long)n, lr_dev,
                       step_dev, beta1, beta2, eps);
  if (!(clear_grads & 2)) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, s, step_dev);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// ------------------------------------------------------------------------------
// RAdam (crank/net/trainer/utils.py:44-45: torch_optimizer.RAdam(lr), a third-party package absent from the reference
// tree; restated from Liu et al., "On the Variance of the Adaptive Learning Rate and Beyond", Alg. 2, in the form that
// package publishes: betas (0.9, 0.999), eps 1e-8, no weight decay; the length of the approximated SMA decides per
// STEP between the rectified adaptive update and a momentum-only one, threshold N_sma >= 5).  The step's scalars come
// out of a cancellation (N_sma = N_max - 2 t b2^t / (1 - b2^t) ~ t while N_max = 1999), so they are formed in double -
// once per thread; the element arithmetic is fp32 like the package's.
// ------------------------------------------------------------------------------
struct RAdamCoef { float step_size; bool rect; };
__device__ __forceinline__ RAdamCoef radam_coef(const float* lr_dev, const float* step_dev, double b1, double b2) {
  const double t = (double)step_dev[0] + 1.0;
  const double b2t = pow(b2, t), bc1 = 1.0 - pow(b1, t);
  const double nmax = 2.0 / (1.0 - b2) - 1.0;
  const double nsma = nmax - 2.0 * t * b2t / (1.0 - b2t);
  RAdamCoef c;
  c.rect = nsma >= 5.0;
  const double lr = (double)lr_dev[0];
  c.step_size = (float)(c.rect ? lr * sqrt((1.0 - b2t) * (nsma - 4.0) / (nmax - 4.0) * (nsma - 2.0) / nsma * nmax / (nmax - 2.0)) / bc1
                               : lr / bc1);
  return c;
}
// b1 / b2 and their complements arrive as the fp32 roundings of the host's doubles (1 - 0.999f is 1.3e-5 off 0.001)
struct MomentCoef { float b1, omb1, b2, omb2, eps; };
static MomentCoef moment_coef(double beta1, double beta2, double eps) {
  MomentCoef k;
  k.b1 = (float)beta1; k.omb1 = (float)(1.0 - beta1); k.b2 = (float)beta2; k.omb2 = (float)(1.0 - beta2); k.eps = (float)eps;
  return k;
}
template <bool CLEAR>
__global__ __launch_bounds__(256) void radam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, const float* __restrict__ lr_dev,
                                                    float* __restrict__ step_dev, double beta1, double beta2, MomentCoef k) {
#pragma clang fp contract(off)  // mul_ / add_ / addcmul_ / addcdiv_ round one by one
  const RAdamCoef c = radam_coef(lr_dev, step_dev, beta1, beta2);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i];
    if (CLEAR) g[i] = 0.f;
    const float vi = v[i] * k.b2 + k.omb2 * gi * gi;
    const float mi = m[i] * k.b1 + k.omb1 * gi;
    v[i] = vi; m[i] = mi;
    if (c.rect) p[i] -= c.step_size * (mi / (sqrtf(vi) + k.eps));
    else p[i] -= c.step_size * mi;
  }
}

extern "C" int crk_radam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                              const float* lr_dev, float* step_dev, double beta1, double beta2, double eps, int clear_grads,
                              void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  const MomentCoef k = moment_coef(beta1, beta2, eps);
  if (clear_grads & 1)
    hipLaunchKernelGGL(radam_kernel<true>, dim3((int)b), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (long)n, lr_dev,
                       step_dev, beta1, beta2, k);
  else
    hipLaunchKernelGGL(radam_kernel<false>, dim3((int)b), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (long)n, lr_dev,
                       step_dev, beta1, beta2, k);
  if (!(clear_grads & 2)) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, s, step_dev);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// ------------------------------------------------------------------------------
// LAMB (crank/net/trainer/utils.py:46-47: pytorch_lamb.Lamb(lr), third party, absent; restated from You et al., "Large
// Batch Optimization for Deep Learning", Alg. 2, in the form that package publishes: betas (0.9, 0.999), eps 1e-6, no
// weight decay, no bias correction; PER PARAMETER TENSOR r = clamp(||w||, 0, 10) / ||u|| with u = m / (sqrt(v) + eps),
// r = 1 where either norm is 0; w -= lr r u).  The flat block is cut into tiles that never cross a tensor
// (tiles[t] = {offset, length <= LAMB_TILE, tensor, -}; tensors[s] = {first tile, tiles}); two launches:
//   lamb_moments_kernel  tile -> m, v, u (kept in `upd`), {sum w^2, sum u^2} of the tile
//   lamb_apply_kernel    tile -> its tensor's sums (the tiles' pairs in tile order, double), r, the update
// Fixed summation order: bit-reproducible from run to run.
// ------------------------------------------------------------------------------
#define LAMB_TILE 2048
__global__ __launch_bounds__(256) void lamb_moments_kernel(const float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ upd, const int4* __restrict__ tiles,
                                                           float* __restrict__ part, MomentCoef k, int clear) {
#pragma clang fp contract(off)
  __shared__ float sh[4];
  const int4 tl = tiles[blockIdx.x];
  float sw = 0.f, su = 0.f;
  for (int j = threadIdx.x; j < tl.y; j += 256) {
    const long i = (long)tl.x + j;
    const float gi = g[i];
    if (clear) g[i] = 0.f;
    const float mi = m[i] * k.b1 + k.omb1 * gi;
    const float vi = v[i] * k.b2 + k.omb2 * gi * gi;
    m[i] = mi; v[i] = vi;
    const float u = mi / (sqrtf(vi) + k.eps);
    upd[i] = u;
    const float w = p[i];
    sw += w * w; su += u * u;
  }
  sw = block_sum_256(sw, sh);
  su = block_sum_256(su, sh);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = sw; part[2 * blockIdx.x + 1] = su; }
}

__global__ __launch_bounds__(256) void lamb_apply_kernel(float* __restrict__ p, const float* __restrict__ upd,
                                                         const int4* __restrict__ tiles, const int2* __restrict__ tensors,
                                                         const float* __restrict__ part, const float* __restrict__ lr_dev,
                                                         float* __restrict__ ratio_out) {
#pragma clang fp contract(off)
  __shared__ float sr;
  const int4 tl = tiles[blockIdx.x];
  if (threadIdx.x == 0) {
    const int2 ts = tensors[tl.z];
    double sw = 0.0, su = 0.0;
    for (int t = ts.x; t < ts.x + ts.y; t++) { sw += (double)part[2 * t]; su += (double)part[2 * t + 1]; }
    const float wn = fminf((float)sqrt(sw), 10.f), un = (float)sqrt(su);
    const float r = (wn == 0.f || un == 0.f) ? 1.f : wn / un;
    sr = r;
    if (ratio_out && blockIdx.x == ts.x) ratio_out[tl.z] = r;
  }
  __syncthreads();
  const float sc = lr_dev[0] * sr;
  for (int j = threadIdx.x; j < tl.y; j += 256) {
    const long i = (long)tl.x + j;
    p[i] -= sc * upd[i];
  }
}

extern "C" int crk_lamb_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* upd, const int* tiles,
                             int n_tiles, const int* tensors, int n_tensors, float* part, float* ratio_out,
                             const float* lr_dev, float* step_dev, double beta1, double beta2, double eps, int clear_grads,
                             void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !upd || !tiles || !tensors || !part || !lr_dev || !step_dev ||
      n_tiles < 0 || n_tensors < 0)
    return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (n_tiles > 0) {
    hipLaunchKernelGGL(lamb_moments_kernel, dim3(n_tiles), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, upd,
                       reinterpret_cast<const int4*>(tiles), part, moment_coef(beta1, beta2, eps), clear_grads & 1);
    hipLaunchKernelGGL(lamb_apply_kernel, dim3(n_tiles), dim3(256), 0, s, params, upd, reinterpret_cast<const int4*>(tiles),
                       reinterpret_cast<const int2*>(tensors), part, lr_dev, ratio_out);
  }
  if (!(clear_grads & 2)) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, s, step_dev);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
extern "C" int crk_lamb_tile() { return LAMB_TILE; }

// ------------------------------------------------------------------------------
// speaker-embedding gather + conditioning concat (vqvae2.py:154-158,
// trainer_lsgan.py:194-206): out[n, :] = [src0[n, :c0] | table[idx[n], :E] | ...]
// and its backward (scatter-add of the embedding slice into the table gradient).
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void concat_embed_kernel(const float* __restrict__ a, int lda, int ca,
                                                           const float* __restrict__ b2, int ldb, int cb2,
                                                           const float* __restrict__ table, int E,
                                                           const long long* __restrict__ idx, long run, long N,
                                                           float* __restrict__ out, int ldo) {
  const int C = ca + cb2 + E;
  const long total = N * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / C;
    const int c = (int)(i - n * C);
    float v;
    if (c < ca) v = a[n * lda + c];
    else if (c < ca + cb2) v = b2[n * ldb + (c - ca)];
    else v = table[idx[run > 1 ? n - n % run : n] * E + (c - ca - cb2)];  // (run frames share their first frame's label)
    out[n * ldo + c] = v;
  }
}

extern "C" int crk_concat_embed_run(const float* a, int lda, int ca, const float* b, int ldb, int cb, const float* table,
                                    int E, const long long* idx, long long run, long long N, float* out, int ldo,
                                    void* stream) {
  if (!out || run < 1 || (ca > 0 && !a) || (cb > 0 && !b) || (E > 0 && (!table || !idx))) return CRK_ERR_ARG;
  const long total = N * (ca + cb + E);
  hipLaunchKernelGGL(concat_embed_kernel, dim3(loss_blocks(total)), dim3(256), 0, (hipStream_t)stream, a, lda, ca, b, ldb,
                     cb, table, E, idx, (long)run, (long)N, out, ldo);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
extern "C" int crk_concat_embed(const float* a, int lda, int ca, const float* b, int ldb, int cb, const float* table,
                                int E, const long long* idx, long long N, float* out, int ldo, void* stream) {
  return crk_concat_embed_run(a, lda, ca, b, ldb, cb, table, E, idx, 1, N, out, ldo, stream);
}

// embedding-table gradient: dtable[r][e] += sum over frames n with idx[n] == r of dcat[n][c0 + e].
// Two stages, no atomics, fixed summation order (bit-reproducible): a workgroup reduces a run of
// 256 frames into [rows][E] (each (frame-lane, column) thread owns private LDS accumulators,
// combined in lane order), then one pass adds the per-run tables in run order.
#define EMB_FRAMES 256
__global__ __launch_bounds__(256) void embed_bwd_partial_kernel(const float* __restrict__ dcat, int ld, int c0, int E,
                                                                const long long* __restrict__ idx, long run, long N,
                                                                int n_rows, float* __restrict__ part) {
  extern __shared__ float acc[];  // [nsub][n_rows][E]
#define EMB_LABEL(n) idx[run > 1 ? (n) - (n) % run : (n)]
  const int tid = threadIdx.x;
  const int nsub = 256 / E, tab = n_rows * E;
  for (int i = tid; i < nsub * tab; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int e = tid % E, sub = tid / E;
  if (sub < nsub) {
    const long beg = (long)blockIdx.x * EMB_FRAMES;
    const long end = min(N, beg + EMB_FRAMES);
    // eight frames' loads in flight, then the accumulation in frame order (one frame at a time was a chain of 256 / nsub
    // dependent HBM round trips per thread: 28 us for a 4 MB read)
    long n = beg + sub;
    for (; n + 7 * (long)nsub < end; n += 8 * (long)nsub) {
      long r[8]; float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { r[u] = EMB_LABEL(n + u * (long)nsub); v[u] = dcat[(n + u * (long)nsub) * ld + c0 + e]; }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (r[u] >= 0 && r[u] < n_rows) acc[(sub * n_rows + (int)r[u]) * E + e] += v[u];
    }
    for (; n < end; n += nsub) {
      const long r = EMB_LABEL(n);
      if (r >= 0 && r < n_rows) acc[(sub * n_rows + (int)r) * E + e] += dcat[n * ld + c0 + e];
    }
  }
  __syncthreads();
  for (int i = tid; i < tab; i += 256) {
    float s = 0.f;
    for (int u = 0; u < nsub; u++) s += acc[u * tab + i];
    part[(long)blockIdx.x * tab + i] = s;
  }
#undef EMB_LABEL
}
__global__ __launch_bounds__(256) void embed_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int tab,
                                                               float* __restrict__ dtable) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= tab) return;
  // (32 tables in flight: at the benchmark shape 125 tables are four memory round trips instead of sixteen; same order)
  float s = 0.f;
  int b = 0;
  for (; b + 32 <= nblk; b += 32) {
    float t[32];
#pragma unroll
    for (int u = 0; u < 32; u++) t[u] = part[(long)(b + u) * tab + i];
#pragma unroll
    for (int u = 0; u < 32; u++) s += t[u];
  }
  for (; b + 8 <= nblk; b += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) t[u] = part[(long)(b + u) * tab + i];
#pragma unroll
    for (int u = 0; u < 8; u++) s += t[u];
  }
  for (; b < nblk; b++) s += part[(long)b * tab + i];
  dtable[i] += s;
}

extern "C" long long crk_embed_bwd_scratch_floats(long long N, int E, int n_rows) {
  return ((N + EMB_FRAMES - 1) / EMB_FRAMES) * (long long)n_rows * E;
}

extern "C" int crk_embed_bwd_run(const float* dcat, int ld, int c0, int E, const long long* idx, long long run, long long N,
                                 int n_rows, float* dtable, float* scratch, void* stream) {
  if (!dcat || !idx || !dtable || !scratch || E <= 0 || E > 256 || run < 1) return CRK_ERR_ARG;
  const int nsub = 256 / E;
  if ((long long)nsub * n_rows * E * 4 > 60 * 1024) return CRK_ERR_UNSUPPORTED;
  const int nb = (int)((N + EMB_FRAMES - 1) / EMB_FRAMES), tab = n_rows * E;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_bwd_partial_kernel, dim3(nb), dim3(256), (size_t)nsub * tab * sizeof(float), s, dcat, ld, c0, E,
                     idx, (long)run, (long)N, n_rows, scratch);
  hipLaunchKernelGGL(embed_bwd_reduce_kernel, dim3((tab + 255) / 256), dim3(256), 0, s, scratch, nb, tab, dtable);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
extern "C" int crk_embed_bwd(const float* dcat, int ld, int c0, int E, const long long* idx, long long N, int n_rows,
                             float* dtable, float* scratch, void* stream) {
  return crk_embed_bwd_run(dcat, ld, c0, E, idx, 1, N, n_rows, dtable, scratch, stream);
}
