// Device helpers shared by the fused stack kernels (stack_kernels.hip, pstack_kernels.hip).
#ifndef CRK_STACK_COMMON_H
#define CRK_STACK_COMMON_H
#include "common.h"

#define SK_GUARD 16  // zero guard rows above and below the operand tile (>= largest tap offset)
#define SK_XS 144    // operand row stride: 64 bf16 + 16 B pad (conflict-free ds_read_b128)

typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int sk_u32x2 __attribute__((ext_vector_type(2)));
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));

// Plane accesses go through buffer descriptors: scalar base + one 32-bit VGPR offset +
// an immediate (the constant part of the offset expression is folded into the instruction's
// offset field; soffset stays 0.  Do NOT pass the constant as soffset: it lands in an SGPR, and
// hipcc's hazard recogniser then treats a 16-byte buffer store whose data registers are
// rewritten by the very next VALU instruction as safe - on gfx950 it is not, the first dword
// of the store was observed corrupted), hardware bounds check (an out-of-range offset drops the store / loads 0), so
// invalid frames need no branch and no 64-bit per-element addresses are kept alive.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc(const float* base, long n_floats) {
  const long bytes = n_floats * 4;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}
#define SK_OOB 0x7ffffff0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc16(const uint16_t* base, long n_elems) {
  const long bytes = n_elems * 2;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}

// NOTE: __builtin_bit_cast applied directly to a vector-element lvalue (q[j]) reads element 0
// of the vector with this compiler; going through by-value helpers is required, not style.
__device__ __forceinline__ float sk_u2f(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ unsigned sk_f2u(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float sk_bf_lo(float v) { return v - bf2f(f2bf(v)); }  // rounding residual

// gate nonlinearities: the fast path uses the hardware exp2 / rcp (1 ulp-class), the precise
// path libm expf and an IEEE division
__device__ __forceinline__ float sk_tanh(float x, bool precise) {
#if defined(S2_ABL) && (S2_ABL & 1)
  return x * 0.5f;  // ablation build (timing only): no transcendentals
#endif
  if (precise) return 1.f - 2.f / (1.f + expf(2.f * x));
  // exp(2x) = exp2(x * 2 log2(e)): the constant is twice the fp32 log2(e) that __expf multiplies by, so the product is
  // exactly twice __expf's (bit-identical results) without the v_add that formed 2x
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * sk_u2f(0x4038aa3bu)));
}
__device__ __forceinline__ float sk_sigmoid(float x, bool precise) {
#if defined(S2_ABL) && (S2_ABL & 1)
  return x * 0.25f + 0.5f;
#endif
  if (precise) return 1.f / (1.f + expf(-x));
  return __builtin_amdgcn_rcpf(1.f + __expf(-x));
}

// Gate backward and residual update of the data-gradient chains with the floating-point contraction spelled out: both
// chains (stack_kernels.hip, stack2b_kernels.hip; every template instantiation of each) then round identically - left to
// the compiler, one instantiation formed v_pk_fma where another formed v_pk_mul + v_pk_add and a handful of dG values per
// million landed on the other side of a bf16 rounding boundary.
__device__ __forceinline__ void sk_gate_bwd(float dz, float ta, float sb, float& da, float& db) {
#pragma clang fp contract(off)
  const float omt = __builtin_fmaf(-ta, ta, 1.f);  // 1 - tanh^2
  const float oms = 1.f - sb;
  da = dz * sb * omt;
  db = dz * ta * sb * oms;
}
__device__ __forceinline__ float sk_res_bwd(float dxo, float rs, float cv) { return __builtin_fmaf(dxo, rs, cv); }
__device__ __forceinline__ float sk_mul_nc(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

struct SkRegs {  // one weight chunk (128 rows x 64 k bf16 = 1024 16-byte pieces) per plane, spread over the workgroup
  sk_u32x4 h0, h1, h2, h3, l0, l1, l2, l3;
};
#define SK_ALL(X) X(0) X(1) X(2) X(3)

template <bool PRECISE, int NT>
__device__ __forceinline__ void sk_fetch(SkRegs& w, const uint16_t* shi, const uint16_t* slo, int total, int tid) {
#define SK_F(u)                                                                   \
  if (u * NT < 1024) {                                                            \
    const int idx = tid + u * NT;                                                 \
    const long off = idx < total ? (long)idx * 8 : 0;                             \
    w.h##u = *reinterpret_cast<const sk_u32x4*>(shi + off);                       \
    if (PRECISE) w.l##u = *reinterpret_cast<const sk_u32x4*>(slo + off);          \
  }
  SK_ALL(SK_F)
#undef SK_F
}

// 4 consecutive channels (one register quad) as bf16: hi plane and, for bf16x3, the residual plane
template <bool PRECISE>
__device__ __forceinline__ void sk_quad(float a, float b, float c, float d, sk_u32x2& hi, sk_u32x2& lo) {
  hi[0] = pack_bf2(a, b);
  hi[1] = pack_bf2(c, d);
  if (PRECISE) {
    lo[0] = pack_bf2(sk_bf_lo(a), sk_bf_lo(b));
    lo[1] = pack_bf2(sk_bf_lo(c), sk_bf_lo(d));
  }
}

// accumulator-layout quads g0 (P) and g0+1 (Q) of one 16-channel group -> this lane's B fragment:
// the half-0 lane of a frame keeps P and receives the half-1 lane's P (channels 0..7), the half-1
// lane keeps Q and receives the half-0 lane's Q (channels 8..15).
__device__ __forceinline__ bf16x8 sk_swap_frag(sk_u32x2 P, sk_u32x2 Q) {
  const sk_u32x2 s0 = __builtin_amdgcn_permlane32_swap(P[0], Q[0], false, false);
  const sk_u32x2 s1 = __builtin_amdgcn_permlane32_swap(P[1], Q[1], false, false);
  const sk_u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ sk_u32x4 sk_frag_bits(bf16x8 f) { return __builtin_bit_cast(sk_u32x4, f); }


// ---- transpose reads (reduction axis = frame axis) for the weight-gradient kernels ----
#define SW_LDS __attribute__((address_space(3)))
typedef short sw_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 sw_tr_frag(const unsigned char* p0, int rs) {
  const sw_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SW_LDS sw_v4s*)(p0));
  const sw_v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SW_LDS sw_v4s*)(p0 + 4 * rs));
  typedef short sw_v8s __attribute__((ext_vector_type(8)));
  const sw_v8s r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ float sw_sum8(bf16x8 f) {
  const sk_u32x4 u = __builtin_bit_cast(sk_u32x4, f);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned w = u[j];
    s += sk_u2f(w << 16) + sk_u2f(w & 0xffff0000u);
  }
  return s;
}


#endif
