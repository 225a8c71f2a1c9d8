// Shared device/host helpers for the crank_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CRK_OK 0
#define CRK_ERR_ARG 1
#define CRK_ERR_HIP 2
#define CRK_ERR_UNSUPPORTED 3
// flags of crk_net_forward / crk_net_backward: the values of include/crank_hip.h (that header is the C ABI's; the
// library's sources share this one)
#ifndef CRK_FLAG_PRECISE
#define CRK_FLAG_PRECISE 1
#define CRK_FLAG_NO_PARAM_GRAD 2
#define CRK_FLAG_NO_SAVE 4
#define CRK_FLAG_DEFER_WNORM 8
#define CRK_FLAG_SEED_ON_DEVICE 16
#define CRK_FLAG_FWD_PRECISE 32
#define CRK_FLAG_BWD_PLAIN 64
#endif

#define CRK_CHECK_LAUNCH()                                                        \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "[crank_hip] %s:%d launch failed: %s\n", __FILE__, __LINE__, \
              hipGetErrorString(e_));                                             \
      return CRK_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

// activation codes used by prologues / epilogues
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };

// fp32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {  // {bf16(a), bf16(b)} in one dword, one instruction
  typedef __bf16 crk_bf2 __attribute__((ext_vector_type(2)));
  typedef float crk_f2 __attribute__((ext_vector_type(2)));
  const crk_f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, crk_bf2));
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// split v into hi + lo bf16 (lo = rounding residual); hi alone is the fast path
__device__ __forceinline__ void split_bf(float v, uint16_t& hi, uint16_t& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
  return v;
}
// derivative selector: value>0 works on pre- or post-activation tensors (slope > 0)
__device__ __forceinline__ float act_grad(float side, int act, float slope) {
  if (act == ACT_RELU) return side > 0.f ? 1.f : 0.f;
  if (act == ACT_LRELU) return side > 0.f ? 1.f : slope;
  return 1.f;
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  return *reinterpret_cast<bf16x8*>(&v);
}
__device__ __forceinline__ bf16x8 lds_frag2(const unsigned char* p) {  // 8-byte aligned only
  uint2 a = *reinterpret_cast<const uint2*>(p);
  uint2 b = *reinterpret_cast<const uint2*>(p + 8);
  uint4 v = make_uint4(a.x, a.y, b.x, b.y);
  return *reinterpret_cast<bf16x8*>(&v);
}

// counter-based dropout RNG (keep mask reproducible in backward): one 32-bit hash
// per (seed, element index); keep iff hash >= p * 2^32.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// seed of a launch: a host value, or (ptr non-null) a device-resident base the host value is added to - the form a
// captured HIP graph needs: the base is advanced on the device by crk_seed_next, nothing per-call lives in kernel arguments
__device__ __forceinline__ unsigned long long crk_seed(unsigned long long s, const unsigned long long* ptr) {
  return ptr ? *ptr + s : s;
}
// Four consecutive elements (idx >> 2) share one pair of hashes and take 16 bits each: a kernel that walks a lane's
// 4-channel quads (every fused stack kernel does) pays the hashing once per quad - the compiler merges the identical
// computations - instead of twice per element.  Keep iff the element's 16 bits >= round(p * 2^16): the keep probability
// is p to within 2^-17, the same mask in every kernel that regenerates it (forward, data gradient, weight gradient).
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p) {
  const uint64_t quad = idx >> 2;
  const unsigned ln = (unsigned)idx & 3u;
  uint32_t h = hash32((uint32_t)(quad ^ (quad >> 32)) * 0x9e3779b9u + (uint32_t)seed);
  h = hash32(h ^ (uint32_t)(seed >> 32));
  const uint32_t h2 = hash32(h + 0x68bc21ebu);
  const uint32_t w = (ln & 2u) ? h2 : h;
  const uint32_t bits = (ln & 1u) ? (w >> 16) : (w & 0xffffu);
  const uint32_t thr = (uint32_t)(p * 65536.f + 0.5f);
  return bits < thr ? 0.f : 1.f / (1.f - p);
}

// Adam (torch.optim.Adam defaults, crank/net/trainer/utils.py:40-58) on ONE element (adam_kernel, loss_kernels.hip).
struct AdamCoef { float beta1, beta2, eps, step_size, bc2s; };
__device__ __forceinline__ AdamCoef adam_coef(const float* lr_dev, const float* step_dev, float beta1, float beta2, float eps) {
  const float step = step_dev[0] + 1.f;  // every thread reads the pre-update value
  const float bc1 = 1.f - powf(beta1, step);
  const float bc2 = 1.f - powf(beta2, step);
  AdamCoef c;
  c.beta1 = beta1; c.beta2 = beta2; c.eps = eps;
  c.step_size = lr_dev[0] / bc1;
  c.bc2s = sqrtf(bc2);
  return c;
}
template <bool CLEAR>
__device__ __forceinline__ void adam_elem(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                          long i, const AdamCoef& c) {
  const float gi = g[i];
  if (CLEAR) g[i] = 0.f;  // the gradient block is zero again when the next step starts: no memset launch
  const float mi = m[i] + (gi - m[i]) * (1.f - c.beta1);  // torch: exp_avg.lerp_(grad, 1-beta1)
  const float vi = c.beta2 * v[i] + (1.f - c.beta2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / c.bc2s + c.eps;
  p[i] -= c.step_size * (mi / denom);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
