// VQ codebook kernels:
//   vq_forward   - nearest code (fp32 L2 search, first index wins ties), gather and
//                  straight-through value, replaces Quantizer.vq + the one-hot GEMM
//                  lookup of crank/net/module/vqvae2.py:306-313,333,338-347
//   vq_ema_stats - per-code frame counts and feature sums (vqvae2.py:316-321)
//   vq_ema_apply - EMA blend, Laplace smoothing, codebook refresh (vqvae2.py:316-330)
//
// The distance is evaluated with the reference's own fp32 expression
//   dist = (sum_d W^2 - 2 * x.w) + sum_d x^2
// (two roundings after the dot product, same association as torch evaluates it).
// The codebook (K x D fp32 <= 128 KiB) lives in LDS; a 512-thread workgroup scores
// 128 frames: lane = frame, the 8 waves split {2 frame groups} x {4 code groups} and
// the per-frame minimum is reduced across the code groups through LDS.
//
// EMA sums are accumulated in 64-bit fixed point (2^-28 resolution): integer adds are
// associative, so the statistics - and the codebook derived from them - are bitwise
// reproducible run to run and identical on every data-parallel rank after an
// integer all-reduce, whatever order the adds land in.
#include "common.h"
#include "switches.h"
// (conv_kernels.hip: HIP-event timing per kernel class for bench.py's roofline leg; class 7 = the codebook search)
void conv_prof_begin(int cls, double flops, hipStream_t s);
void conv_prof_end(int cls, hipStream_t s);
void conv_prof_bytes(int cls, double bytes);

#define VQ_FRAMES 128
#define VQ_FIX_SCALE 268435456.0f        // 2^28
#define VQ_FIX_INV 3.7252902984619140625e-9f  // 2^-28

template <int D>
__global__ __launch_bounds__(512) void vq_forward_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ cb, int N, int K, int kchunk,
                                                         long long* __restrict__ idx_out, float* __restrict__ e_out,
                                                         int lde, float* __restrict__ qx_out, int ldq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* cbs = reinterpret_cast<float*>(smem);            // [kchunk][D]
  float* w2s = cbs + (size_t)kchunk * D;                    // [kchunk]
  float* red_d = w2s + kchunk;                              // [4][128]
  int* red_i = reinterpret_cast<int*>(red_d + 4 * VQ_FRAMES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fg = wave & 1, cg = wave >> 1;
  const int fl = fg * 64 + lane;  // frame within block
  const long n = (long)blockIdx.x * VQ_FRAMES + fl;
  const bool valid = n < N;

  float xr[D];
  float x2 = 0.f;
  if (valid) {
    const float* xp = x + n * ldx;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float4 f = *reinterpret_cast<const float4*>(xp + d);
      xr[d] = f.x; xr[d + 1] = f.y; xr[d + 2] = f.z; xr[d + 3] = f.w;
    }
#pragma unroll
    for (int d = 0; d < D; d++) x2 += xr[d] * xr[d];
  } else {
#pragma unroll
    for (int d = 0; d < D; d++) xr[d] = 0.f;
  }

  float best = INFINITY;
  int besti = 0x7fffffff;
  for (int k0 = 0; k0 < K; k0 += kchunk) {
    const int kc = min(kchunk, K - k0);
    __syncthreads();
    for (int i = tid; i < kc * (D / 4); i += 512)
      reinterpret_cast<float4*>(cbs)[i] = reinterpret_cast<const float4*>(cb + (size_t)k0 * D)[i];
    __syncthreads();
    for (int k = tid; k < kc; k += 512) {
      float s = 0.f;
      const float4* wp = reinterpret_cast<const float4*>(cbs + k * D);
#pragma unroll
      for (int d4 = 0; d4 < D / 4; d4++) {  // same d order as a scalar loop, 4x fewer LDS reads
        const float4 w = wp[d4];
        s += w.x * w.x; s += w.y * w.y; s += w.z * w.z; s += w.w * w.w;
      }
      w2s[k] = s;
    }
    __syncthreads();
    const int per = (kc + 3) >> 2;
    const int kb = cg * per, ke = min(kc, kb + per);
    // four codes in flight: each code's dot product is still one d-ordered fmaf chain
    // (the numerics per code are unchanged), the four chains hide each other's latency
    int k = kb;
    for (; k + 4 <= ke; k += 4) {
      const float4* wp = reinterpret_cast<const float4*>(cbs + k * D);
      float dot0 = 0.f, dot1 = 0.f, dot2 = 0.f, dot3 = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < D / 4; d4++) {
        const float4 w0 = wp[d4], w1 = wp[D / 4 + d4], w2 = wp[2 * (D / 4) + d4], w3 = wp[3 * (D / 4) + d4];
        const float a = xr[4 * d4], b = xr[4 * d4 + 1], c = xr[4 * d4 + 2], e = xr[4 * d4 + 3];
        dot0 = fmaf(a, w0.x, dot0); dot1 = fmaf(a, w1.x, dot1); dot2 = fmaf(a, w2.x, dot2); dot3 = fmaf(a, w3.x, dot3);
        dot0 = fmaf(b, w0.y, dot0); dot1 = fmaf(b, w1.y, dot1); dot2 = fmaf(b, w2.y, dot2); dot3 = fmaf(b, w3.y, dot3);
        dot0 = fmaf(c, w0.z, dot0); dot1 = fmaf(c, w1.z, dot1); dot2 = fmaf(c, w2.z, dot2); dot3 = fmaf(c, w3.z, dot3);
        dot0 = fmaf(e, w0.w, dot0); dot1 = fmaf(e, w1.w, dot1); dot2 = fmaf(e, w2.w, dot2); dot3 = fmaf(e, w3.w, dot3);
      }
      const float d0 = (w2s[k] - 2.f * dot0) + x2, d1 = (w2s[k + 1] - 2.f * dot1) + x2;
      const float d2 = (w2s[k + 2] - 2.f * dot2) + x2, d3 = (w2s[k + 3] - 2.f * dot3) + x2;
      if (d0 < best) { best = d0; besti = k0 + k; }
      if (d1 < best) { best = d1; besti = k0 + k + 1; }
      if (d2 < best) { best = d2; besti = k0 + k + 2; }
      if (d3 < best) { best = d3; besti = k0 + k + 3; }
    }
    for (; k < ke; k++) {
      const float4* wp = reinterpret_cast<const float4*>(cbs + k * D);
      float dot = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < D / 4; d4++) {
        float4 w = wp[d4];
        dot = fmaf(xr[4 * d4], w.x, dot);
        dot = fmaf(xr[4 * d4 + 1], w.y, dot);
        dot = fmaf(xr[4 * d4 + 2], w.z, dot);
        dot = fmaf(xr[4 * d4 + 3], w.w, dot);
      }
      const float t = w2s[k] - 2.f * dot;
      const float dist = t + x2;
      if (dist < best) { best = dist; besti = k0 + k; }
    }
  }
  red_d[cg * VQ_FRAMES + fl] = best;
  red_i[cg * VQ_FRAMES + fl] = besti;
  __syncthreads();
  if (cg == 0 && valid) {
    float bd = red_d[fl];
    int bi = red_i[fl];
#pragma unroll
    for (int g = 1; g < 4; g++) {
      const float d2 = red_d[g * VQ_FRAMES + fl];
      const int i2 = red_i[g * VQ_FRAMES + fl];
      if (d2 < bd || (d2 == bd && i2 < bi)) { bd = d2; bi = i2; }
    }
    if (bi == 0x7fffffff) bi = 0;  // all-NaN row: torch.argmin would pick a NaN slot; pin to 0
    idx_out[n] = (long long)bi;
    const float* ep = cb + (size_t)bi * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float4 e = *reinterpret_cast<const float4*>(ep + d);
      if (e_out) *reinterpret_cast<float4*>(e_out + n * lde + d) = e;
      if (qx_out) {
        // straight-through value x + (e - x), two roundings like the reference
        float4 q;
        q.x = xr[d] + (e.x - xr[d]);
        q.y = xr[d + 1] + (e.y - xr[d + 1]);
        q.z = xr[d + 2] + (e.z - xr[d + 2]);
        q.w = xr[d + 3] + (e.w - xr[d + 3]);
        *reinterpret_cast<float4*>(qx_out + n * ldq + d) = q;
      }
    }
  }
}

// ------------------------------------------------------------------------------
// Same search with the roles swapped (K <= 512): a LANE owns a CODE (its D weights in registers), the
// eight waves of a workgroup cover 512 codes, and the frames of the block are visited one after the
// other with the frame's row in SGPRs (uniform address -> scalar loads), so a distance costs D
// v_fmac with a scalar operand and no LDS traffic at all (the frame-per-lane kernel above reads the
// codebook from LDS by broadcast and is bound by that).  Per frame a DPP reduction gives the wave's
// minimum, a ballot the lowest lane that attains it (= lowest code index: ties go to the first
// index like torch.argmin); the eight wave results meet in LDS once per block.
// Arithmetic per (frame, code) is the same d-ordered chain as above: identical indices.
// ------------------------------------------------------------------------------
#define VQL_FB 64  // frames per workgroup

__device__ __forceinline__ float vq_dpp_min(float v) {
  // rows of 16 lanes: quad swaps, half mirror, mirror; then row broadcasts 15 -> rows 1,3 and 31 -> rows 2,3:
  // lane 63 ends up with the minimum of the wave
  // v = min(v, v permuted) in place, one instruction per step (the builtin route costs a move, a
  // canonicalising max and the min); "s_nop 1": a DPP operand written by the previous VALU instruction
  // needs two wait states, which the compiler cannot insert inside inline assembly.
  // v_min_f32 in IEEE mode returns the other operand when one is NaN: NaN distances never win.
#define VQ_DPP(ctrl) asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " ctrl : "+v"(v));
  VQ_DPP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
  VQ_DPP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
  VQ_DPP("row_half_mirror row_mask:0xf bank_mask:0xf")
  VQ_DPP("row_mirror row_mask:0xf bank_mask:0xf")
  VQ_DPP("row_bcast:15 row_mask:0xa bank_mask:0xf")
  VQ_DPP("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef VQ_DPP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int D>
__global__ __launch_bounds__(512) void vq_forward_lc_kernel(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ cb, int N, int K,
                                                            long long* __restrict__ idx_out, float* __restrict__ e_out,
                                                            int lde, float* __restrict__ qx_out, int ldq) {
  __shared__ float x2s[VQL_FB];
  __shared__ float red_d[8][VQL_FB];
  __shared__ int red_i[8][VQL_FB];
  __shared__ int best_s[VQL_FB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long n0 = (long)blockIdx.x * VQL_FB;
  const int nf = (int)min((long)VQL_FB, (long)N - n0);
  const int k = wave * 64 + lane;
  const bool kv = k < K;

  float w[D];
  {
    const float* wp = cb + (size_t)(kv ? k : 0) * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const float4 f = *reinterpret_cast<const float4*>(wp + d);
      w[d] = f.x; w[d + 1] = f.y; w[d + 2] = f.z; w[d + 3] = f.w;
    }
  }
  float w2 = 0.f;
#pragma unroll
  for (int d = 0; d < D; d++) w2 += w[d] * w[d];
  for (int f = tid; f < nf; f += 512) {
    const float* xp = x + (n0 + f) * ldx;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const float4 v = *reinterpret_cast<const float4*>(xp + d);
      s += v.x * v.x; s += v.y * v.y; s += v.z * v.z; s += v.w * v.w;
    }
    x2s[f] = s;
  }
  __syncthreads();

  for (int f = 0; f < nf; f++) {
    const float* xp = x + (n0 + f) * ldx;  // uniform: the row is fetched with scalar loads
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < D; d++) dot = fmaf(xp[d], w[d], dot);
    float dist = (w2 - 2.f * dot) + x2s[f];
    dist = kv ? dist : INFINITY;
    const float m = vq_dpp_min(dist);
    const unsigned long long hit = __ballot(dist == m);
    const int first = hit ? (int)__builtin_ctzll(hit) : 0x7fffffff - wave * 64;  // all NaN: no lane attains the minimum
    if (lane == 0) {
      red_d[wave][f] = m;
      red_i[wave][f] = wave * 64 + first;
    }
  }
  __syncthreads();
  for (int f = tid; f < nf; f += 512) {
    float bd = red_d[0][f];
    int bi = red_i[0][f];
#pragma unroll
    for (int g = 1; g < 8; g++) {
      const float d2 = red_d[g][f];
      const int i2 = red_i[g][f];
      if (d2 < bd || (d2 == bd && i2 < bi)) { bd = d2; bi = i2; }
    }
    if (bi >= K) bi = 0;  // all-NaN row (see above): pin to 0
    best_s[f] = bi;
    idx_out[n0 + f] = (long long)bi;
  }
  __syncthreads();
  // gathered code vectors and the straight-through value x + (e - x), two roundings like the reference
  for (int i = tid; i < nf * (D / 4); i += 512) {
    const int f = i / (D / 4), c = (i - f * (D / 4)) * 4;
    const long n = n0 + f;
    const float4 e = *reinterpret_cast<const float4*>(cb + (size_t)best_s[f] * D + c);
    if (e_out) *reinterpret_cast<float4*>(e_out + n * lde + c) = e;
    if (qx_out) {
      const float4 xv = *reinterpret_cast<const float4*>(x + n * ldx + c);
      float4 q;
      q.x = xv.x + (e.x - xv.x);
      q.y = xv.y + (e.y - xv.y);
      q.z = xv.z + (e.z - xv.z);
      q.w = xv.w + (e.w - xv.w);
      *reinterpret_cast<float4*>(qx_out + n * ldq + c) = q;
    }
  }
}

// ------------------------------------------------------------------------------
// The same search on the matrix cores (D = 64, K <= 512): v_mfma_f32_32x32x2_f32 takes fp32 operands and
// accumulates each output element as a k-ordered chain of fmaf - bit for bit the d-ordered chain
// `dot = fmaf(x[d], w[d], dot)` of the two kernels above (MI355X_MICROARCH.md, "FP32-input MFMA") - at the full
// fp32 rate from one wave per SIMD, where the code-per-lane kernel reaches a quarter of it (its frame rows arrive
// through 64 SGPRs that cannot be double buffered).  A = a tile of 32 codes (rows), B = the wave's 32 frames
// (columns), 32 MFMAs walk d = 0..63 two at a time; the accumulator then holds, per lane, 16 codes of ONE frame in
// ascending order, so the running (min, first index) is lane-local and the two lanes of a frame meet once at the
// end.  The codebook sits in LDS in fragment order ([tile][d/8][lane][4 consecutive d pairs]: one ds_read_b128
// feeds four MFMAs), the frame rows in registers.  Workgroup = 4 waves = 128 frames, one wave per SIMD: 250
// workgroups for the 32 000 frames of the benchmark batch, 512 MFMAs x 64 cycles each.
// dist = (sum_d w^2 - 2 dot) + sum_d x^2 with both sums formed exactly as above: identical indices.
#define VQM_FB 128
typedef float vq_f32x4 __attribute__((ext_vector_type(4)));
#ifdef VQ_PROF
__device__ unsigned long long vq_prof_buf[256 * 4];
extern "C" int crk_debug_vq_prof(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vq_prof_buf), sizeof(unsigned long long) * 256 * 4) == hipSuccess ? 0 : 2;
}
#define VQ_T(i) if (tid == 0 && blockIdx.x < 256) vq_prof_buf[blockIdx.x * 4 + (i)] = __builtin_readcyclecounter() - vq_t0_;
#else
#define VQ_T(i)
#endif

// Optional fusions (crk_vq_forward_fused): `add` - the quantizer's input is x + add (vqvae2.py:177, "enc[n] + dec"), formed
// on the fly and written to `xsum`; `cpart` - per-workgroup partials {sum (x - e)^2, count} over the frames `mask`
// selects: the commitment loss (trainer_vqvae.py:227-237), which the gather epilogue has both operands of.
struct VqFuse {
  const float* add; int ldadd;
  float* xsum; int ldsum;
  const unsigned char* mask;
  float* cpart;
  const unsigned char* img;  // prepared codebook image (crk_vq_image_build_multi; vq_forward_f16_kernel only), or null
  int xsum_early;            // vq_forward_f16_kernel, image path: xsum is stored where x + add is formed (see there)
};
// One term of a code's squared norm, w2 + e * e with the product rounded on its own (NOT an fma): the chain every search kernel
// of this file forms and the one the exact re-scoring compares against.  Spelled out because the compiler's contraction
// depends on the surroundings - the same source line became v_pk_mul + v_add in one kernel and v_fmac in another, one ulp
// apart, which is enough to send a near-tie to the other code.
__device__ __forceinline__ float vq_sq_acc(float w2, float e) {
#pragma clang fp contract(off)
  const float sq = e * e;
  return w2 + sq;
}
// TP = 2: eight waves - two per SIMD - share the workgroup's 128 frames: wave (fg, tp) takes the code tiles tp, tp + 2, ... of
// frame group fg, the two candidates of a frame meet through LDS under the same (distance, then index) rule.  The argmin's
// VALU instructions do not hide under the fp32 MFMAs of their own wave (tools/probe/mfma_f32_chain.hip): with a second wave
// on the SIMD about half of them run under the other wave's MFMAs, and staging has twice the threads.  Every distance is the
// same chain of operations as with TP = 1: identical indices.  (Tile count a multiple of 4; otherwise TP = 1.)
template <int TP>
__global__ __launch_bounds__(256 * TP, 1) void vq_forward_mfma_kernel(const float* __restrict__ x, int ldx,
                                                                      const float* __restrict__ cb, int N, int K,
                                                                      long long* __restrict__ idx_out, float* __restrict__ e_out,
                                                                      int lde, float* __restrict__ qx_out, int ldq, const VqFuse fz) {
  constexpr int D = 64, NT = 256 * TP;
  __shared__ float vq_red[8 * TP];
  __shared__ float vq_xb[TP][VQM_FB];
  __shared__ int vq_xi[TP][VQM_FB];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wf = reinterpret_cast<float*>(smem);       // [KT tiles][8][64 lanes][4]
  const int KT = ((K + 63) >> 6) * 2;               // 32-code tiles, an even number of them
  float* w2s = wf + (size_t)KT * 32 * D;            // [KT * 32]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), fg = wave & 3, tp = wave >> 2;
#ifdef VQ_PROF
  const unsigned long long vq_t0_ = __builtin_readcyclecounter();
#endif

  // ---- this lane's frame: the whole row once (x2 in d order), its half of every d pair kept as the B operand ----
  const long n = (long)blockIdx.x * VQM_FB + fg * 32 + l31;
  const bool valid = n < N;
  float xb[32];
  float x2 = 0.f;
  vq_f32x4 row[16];  // requested now, consumed behind the codebook staging (one memory round trip, not two)
  {
    const float* xp = x + (valid ? n : 0) * (long)ldx;
#pragma unroll
    for (int q = 0; q < 16; q++) row[q] = *reinterpret_cast<const vq_f32x4*>(xp + 4 * q);
    if (fz.add) {
      const float* ap = fz.add + (valid ? n : 0) * (long)fz.ldadd;
#pragma unroll
      for (int q = 0; q < 16; q++) row[q] += *reinterpret_cast<const vq_f32x4*>(ap + 4 * q);
    }
  }
  // ---- codebook -> LDS in fragment order.  Pass 1: coalesced 16-byte pieces (a code's row = 16 lanes, a wave-load =
  // 1 KB of consecutive codebook; a thread walking its own row touches 64 cache lines per instruction).  Piece
  // (code k, d0 = 4c .. 4c+3) holds d pairs s0 = 2c, 2c+1 for both halves h: element (row i, d = 2 s + h) lives at
  // [tile][s / 4][lane = i + 32 h][s % 4], so the piece is two 8-byte stores.  Pass 2: a thread owns a code and forms
  // its squared norm in d order from the LDS image.
  for (int j0 = 0; j0 < KT * 2 / TP; j0 += 16) {  // 16 loads in flight per thread: two memory round trips for K = 512 (TP = 1)
    vq_f32x4 pv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int pi = tid + NT * (j0 + j), k = pi >> 4;
      pv[j] = (j0 + j < KT * 2 / TP && k < K) ? *reinterpret_cast<const vq_f32x4*>(cb + (size_t)pi * 4) : vq_f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (j0 + j < KT * 2 / TP) {
        const int pi = tid + NT * (j0 + j), k = pi >> 4, c = pi & 15;
        const int ct = k >> 5, i = k & 31, s0 = 2 * c;
        float* dst = wf + (((size_t)ct * 8 + (s0 >> 2)) * 64 + i) * 4 + (s0 & 3);
        typedef float vq_f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<vq_f32x2*>(dst) = vq_f32x2{pv[j][0], pv[j][2]};            // h = 0: d = 4c, 4c + 2
        *reinterpret_cast<vq_f32x2*>(dst + 32 * 4) = vq_f32x2{pv[j][1], pv[j][3]};   // h = 1: d = 4c + 1, 4c + 3
      }
    }
  }
  __syncthreads();
  for (int k = tid; k < KT * 32; k += NT) {
    const int ct = k >> 5, i = k & 31;
    float w2 = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
      const vq_f32x4 e0 = *reinterpret_cast<const vq_f32x4*>(wf + (((size_t)ct * 8 + s4) * 64 + i) * 4);       // d = 8 s4 + 0, 2, 4, 6
      const vq_f32x4 e1 = *reinterpret_cast<const vq_f32x4*>(wf + (((size_t)ct * 8 + s4) * 64 + i + 32) * 4);  // d = 8 s4 + 1, 3, 5, 7
#pragma unroll
      for (int j = 0; j < 4; j++) { w2 = vq_sq_acc(w2, e0[j]); w2 = vq_sq_acc(w2, e1[j]); }
    }
    w2s[k] = k < K ? w2 : INFINITY;
  }
#pragma unroll
  for (int q = 0; q < 16; q++) {
    x2 = vq_sq_acc(vq_sq_acc(vq_sq_acc(vq_sq_acc(x2, row[q][0]), row[q][1]), row[q][2]), row[q][3]);
    xb[2 * q] = half ? row[q][1] : row[q][0];
    xb[2 * q + 1] = half ? row[q][3] : row[q][2];
  }
  __syncthreads();
  VQ_T(0)

  float best = INFINITY;
  int besti = 0x7fffffff;
  const float* wl = wf + lane * 4;
// the 32 MFMAs of code tile ct into accumulator `accv`
// (A fragments of the tile come from `acur`, read one tile ahead; the reads of the following tile go into `anxt`)
#define VQM_TILE(accv, ct, acur, anxt)                                                                       \
  {                                                                                                          \
    _Pragma("unroll") for (int r = 0; r < 16; r++) accv[r] = 0.f;                                            \
    const float* wt = wl + (size_t)((ct) + TP < KT ? (ct) + TP : (ct)) * 8 * 64 * 4;                         \
    _Pragma("unroll") for (int s4 = 0; s4 < 8; s4++) anxt[s4] = *reinterpret_cast<const vq_f32x4*>(wt + s4 * 64 * 4); \
    _Pragma("unroll") for (int s4 = 0; s4 < 8; s4++) {                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; j++) accv = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[s4][j], xb[4 * s4 + j], accv, 0, 0, 0); \
    }                                                                                                        \
  }
// this lane's 16 codes of tile ct, ascending: strict < keeps the first index
#define VQM_PICK(accv, ct)                                                                                   \
  _Pragma("unroll") for (int r = 0; r < 16; r++) {                                                           \
    const int kk = (ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;                                            \
    const float dist = (w2s[kk] - 2.f * accv[r]) + x2;                                                       \
    if (dist < best) { best = dist; besti = kk; }                                                            \
  }
  // Two accumulators: the selection over tile ct (80 VALU instructions) is issued between the MFMAs of tile ct + 1
  // (three per MFMA; on its own the compiler emits the 32 MFMAs, then the selection behind a pipeline drain).
  // KT2 = tile count rounded up to even (codes beyond K carry an infinite norm), last pair peeled: no branch
  // inside a pipelined region.
#define VQM_PAIR(accn, ctn, acur, anxt, accp, ctp)                                         \
  VQM_TILE(accn, ctn, acur, anxt)                                                          \
  VQM_PICK(accp, ctp)                                                                      \
  _Pragma("unroll") for (int m_ = 0; m_ < 32; m_++) {                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
    if (m_ < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                         \
    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                     \
  }                                                                                        \
  __builtin_amdgcn_sched_barrier(0);
  // this wave's tiles: tp, tp + TP, ... (KT / TP of them, an even count)
  f32x16 acc0, acc1;
  vq_f32x4 aA[8], aB[8];
#pragma unroll
  for (int s4 = 0; s4 < 8; s4++) aA[s4] = *reinterpret_cast<const vq_f32x4*>(wl + (size_t)tp * 8 * 64 * 4 + s4 * 64 * 4);
  VQM_TILE(acc0, tp, aA, aB)
  __builtin_amdgcn_sched_barrier(0);
  for (int ct = tp; ct < KT - 2 * TP; ct += 2 * TP) {
    VQM_PAIR(acc1, ct + TP, aB, aA, acc0, ct)
    VQM_PAIR(acc0, ct + 2 * TP, aA, aB, acc1, ct + TP)
  }
  VQM_PAIR(acc1, KT - TP + tp, aB, aA, acc0, KT - 2 * TP + tp)
  VQM_PICK(acc1, KT - TP + tp)
#undef VQM_PAIR
#undef VQM_TILE
#undef VQM_PICK
  // the other half-wave holds the other 16 codes per tile of the same frame
  {
    const float ob = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(besti, 32, 64);
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (TP > 1) {  // the other wave of this frame group holds the other tiles' candidate
    if (half == 0) { vq_xb[tp][fg * 32 + l31] = best; vq_xi[tp][fg * 32 + l31] = besti; }
    __syncthreads();
    const float ob = vq_xb[tp ^ (TP - 1)][fg * 32 + l31];
    const int oi = vq_xi[tp ^ (TP - 1)][fg * 32 + l31];
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (besti >= K) besti = 0;  // all-NaN row: torch.argmin would pick a NaN slot; pin to 0 like the other kernels
  VQ_T(1)
  if (valid && half == 0 && tp == 0) idx_out[n] = (long long)besti;
  // gathered code vectors and the straight-through value x + (e - x), two roundings like the reference.  16 lanes per
  // row (16 bytes each), four frames per instruction: whole cache lines per access (a lane walking its own row
  // touches 64 lines per instruction, and this phase took a third of the kernel).
  float csum = 0.f, ccnt = 0.f;
  {
    const int sub = lane >> 4, c4 = (lane & 15) * 4;
    const long nw = (long)blockIdx.x * VQM_FB + fg * 32;
#pragma unroll
    for (int g = tp * (8 / TP); g < (tp + 1) * (8 / TP); g++) {  // (the two waves of a frame group split its rows)
      const int f = 4 * g + sub;
      const int bi = __shfl(besti, f, 64);
      const long nf = nw + f;
      if (nf < N) {
        const float4 e = *reinterpret_cast<const float4*>(cb + (size_t)bi * D + c4);
        if (e_out) *reinterpret_cast<float4*>(e_out + nf * (long)lde + c4) = e;
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qx_out || fz.xsum || fz.cpart) {
          xv = *reinterpret_cast<const float4*>(x + nf * (long)ldx + c4);
          if (fz.add) {  // the same sum as the search took, element for element
            const float4 av = *reinterpret_cast<const float4*>(fz.add + nf * (long)fz.ldadd + c4);
            xv.x += av.x; xv.y += av.y; xv.z += av.z; xv.w += av.w;
          }
          if (fz.xsum) *reinterpret_cast<float4*>(fz.xsum + nf * (long)fz.ldsum + c4) = xv;
        }
        if (fz.cpart && (!fz.mask || fz.mask[nf])) {
          const float d0 = xv.x - e.x, d1 = xv.y - e.y, d2 = xv.z - e.z, d3 = xv.w - e.w;
          csum += d0 * d0; csum += d1 * d1; csum += d2 * d2; csum += d3 * d3;
          ccnt += 4.f;
        }
        if (qx_out) {
          float4 o;
          o.x = xv.x + (e.x - xv.x);
          o.y = xv.y + (e.y - xv.y);
          o.z = xv.z + (e.z - xv.z);
          o.w = xv.w + (e.w - xv.w);
          *reinterpret_cast<float4*>(qx_out + nf * (long)ldq + c4) = o;
        }
      }
    }
  }
  if (fz.cpart) {  // (uniform)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { csum += __shfl_xor(csum, o, 64); ccnt += __shfl_xor(ccnt, o, 64); }
    if (lane == 0) { vq_red[wave] = csum; vq_red[4 * TP + wave] = ccnt; }
    __syncthreads();
    if (tid == 0) {
      float a = vq_red[0], c = vq_red[4 * TP];
#pragma unroll
      for (int w = 1; w < 4 * TP; w++) { a += vq_red[w]; c += vq_red[4 * TP + w]; }
      fz.cpart[2 * blockIdx.x] = a; fz.cpart[2 * blockIdx.x + 1] = c;
    }
  }
  VQ_T(2)
}

// =====================================================================================================================
// Round 4: the same search with the fp32 matrix pipe (64 cycles per 32x32x2 MFMA, 36 us per call) replaced by a
// SPLIT-f16 search on the f16 matrix pipe plus an exact re-scoring of the few frames it cannot decide.
//
//   approximate distance  d~_k = w2_k - 2 (xh.wh + xh.wl + xl.wh),   x = xh + xl + O(2^-22 |x|) (two f16 roundings of
//   2^ex x, a per-frame power of two that puts the row's largest element into [2^10, 2^11); w likewise with one power of
//   two for the codebook): 12 MFMAs 32x32x16 f16 per (32 codes x 32 frames) instead of 32 fp32 MFMAs, 408 cycles against
//   2048.  The hi.hi products go to one accumulator (a 64-term fp32 sum, like the exact chain), the cross terms (2^-11
//   smaller) to a second one.
//   (every CODE has its own power of two: a codebook that went through the reference's EMA update holds never-used codes
//   of magnitude 1e5 next to codes of magnitude 1 - quirk Q2 - and one common scale would push the small ones into f16's
//   subnormals.)
//   bound  |d~_k - d_k| <= delta_k for the value d_k the exact kernel forms (fp32 fma chain over d, then (w2 - 2 dot) + x2):
//   truncation of the splits 3 x 2^-22 S_k + the two 64-term fp32 sums 2 x 64 x 2^-24 S_k + cross-term and final roundings,
//   S_k = sum_d |x_d w_kd| <= sqrt(x2 w2_k)  ->  delta_k <= 2e-5 sqrt(x2 w2_k) + 5e-7 (x2 + w2_k).
//   A code with ||w_k|| > R := 2.001 ||x|| + sqrt(max(m1 + delta_i1, 0)) cannot win whatever its rounding errors
//   (d_k >= w2_k - 2.001 ||x|| ||w_k|| > m1 + delta_i1), so thr := delta_i1 + delta(||w|| = min(||w||max, R)) separates
//   code i1 from every code that can.
//   Per frame the three smallest d~ are tracked as KEYS: the value with its 7 low mantissa bits replaced by the candidate's
//   place in the lane (round 5; the codes of the first two are read back from the keys after the loop).  A key is within
//   2^-16 |d~| of its value (127 subnormal steps of it where d~ is subnormal), and the comparisons below carry that on top
//   of thr (VQH_KEY_EPS (|m1| + |m_j|) + 2 VQH_KEY_ABS):
//     m2 - m1 > thr               : code i1 IS the exact kernel's argmin (every other code is at least thr above it);
//     else, m3 - m1 > thr         : the answer is i1 or i2 - both get the exact fp32 chain (one lane each), (distance,
//                                   index) order decides: ~0.2 % of random frames, every exact tie;
//     else (3 codes within thr, a non-finite row, exponents out of range): the frame takes the full exact scan, lane = code.
// Indices are therefore those of vq_forward_mfma_kernel / the VALU chain bit for bit, by construction and by test
// (tests/test_gpu_ops.py: both kernels on the same inputs, ties and near-ties included).
typedef _Float16 vq_h8 __attribute__((ext_vector_type(8)));
typedef float vq_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 vq_h4 __attribute__((ext_vector_type(4)));
#define VQH_KEY_EPS 1.6e-5f  // > 2^-16: what 7 replaced mantissa bits can move a value by, relative to it
#ifndef VQH_KEY_ABS  // (-DVQH_KEY_ABS=0.f: the negative control of the zero_rows_zero_codes test case)
#define VQH_KEY_ABS 1.2e-38f  // ... and absolutely, for subnormal values (127 x 2^-149 = 1.8e-43)
#endif
#define VQH_WS 68  // row stride (floats) of the transient fp32 codebook image: conflict-free 16-byte row reads

__device__ unsigned long long vq_f16_flag_counts[3];  // frames decided by [1] the two-candidate re-scoring [2] the full scan ([0] unused)
extern "C" int crk_debug_vq_flags(unsigned long long* host_out3, int reset) {
  if (hipMemcpyFromSymbol(host_out3, HIP_SYMBOL(vq_f16_flag_counts), 3 * sizeof(unsigned long long)) != hipSuccess) return 2;
  if (reset) {
    const unsigned long long z[3] = {0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(vq_f16_flag_counts), z, sizeof(z)) != hipSuccess) return 2;
  }
  return 0;
}

// v_min_f32 as it is: fminf() comes with a canonicalising v_max_f32 per operand the compiler cannot prove quiet (a key is
// made with integer operations), one more VALU operation per candidate
__device__ __forceinline__ float vq_min_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
struct VqTop3 { float m1, m2, m3; int i1, i2; };
// insert (v, k) under the lexicographic (value, index) order (merges across lanes / waves: no scan order to rely on)
__device__ __forceinline__ void vq_top3_insert(VqTop3& t, float v, int k) {
  const bool c1 = v < t.m1 || (v == t.m1 && k < t.i1);
  const bool c2 = v < t.m2 || (v == t.m2 && k < t.i2);
  t.m3 = c2 ? t.m2 : fminf(t.m3, v);
  t.i2 = c1 ? t.i1 : (c2 ? k : t.i2);
  t.m2 = c1 ? t.m1 : (c2 ? v : t.m2);
  t.i1 = c1 ? k : t.i1;
  t.m1 = c1 ? v : t.m1;
}
__device__ __forceinline__ void vq_top3_merge(VqTop3& t, float b1, float b2, float b3, int j1, int j2) {
  vq_top3_insert(t, b1, j1);
  vq_top3_insert(t, b2, j2);
  t.m3 = fminf(t.m3, b3);
}
// power of two 2^e as a float (|e| <= 126)
__device__ __forceinline__ float vq_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }
// unbiased exponent of a finite non-zero float (denormals: -127)
__device__ __forceinline__ int vq_expo(float v) { return (int)((__builtin_bit_cast(unsigned, v) >> 23) & 0xffu) - 127; }

// squared norm of the padding codes k >= K of the last tile (their planes are zero): finite, because the search packs the
// candidate's position into the low mantissa bits of its value and an infinity would turn into a NaN there; larger than any
// value a frame the search trusts can see for a real code short of overflow (and a padding code that does get picked is
// caught by the k < K tests of the decision)
#define VQ_PAD_W2 1.0e37f
// ---- what the split-f16 search derives from the codebook alone (the "codebook image") ----
// per code k: w2 = sum of squares in d order (the exact kernels' chain), the power-of-two scale that puts the largest element
// into [2^10, 2^11), -2 / scale, and the code's contribution to the workgroup-wide largest squared norm (INFINITY for a
// non-finite or out-of-range code: every frame then takes the exact scan; 0 for the padding codes k >= K).
// ONE function for the in-kernel staging and for vq_image_kernel: the two produce the same bits.
__device__ __forceinline__ void vq_code_stats(const float* wrow, int k, int K, float& w2o, float& uswo, float& swso, float& wmxo) {
  float w2 = 0.f, am = 0.f;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const vq_f32x4 e = *reinterpret_cast<const vq_f32x4*>(wrow + 4 * q);
#pragma unroll
    for (int j = 0; j < 4; j++) { w2 = vq_sq_acc(w2, e[j]); am = fmaxf(am, fabsf(e[j])); }
  }
  w2o = k < K ? w2 : VQ_PAD_W2;
  const int er = 10 - vq_expo(am);
  const bool ok = am == 0.f || (w2 < INFINITY && er >= -60 && er <= 60);
  const int ewk = (ok && am > 0.f) ? er : 0;
  uswo = -2.f * vq_pow2(-ewk);
  swso = vq_pow2(ewk);
  wmxo = k < K ? (ok ? (w2 == w2 ? w2 : INFINITY) : INFINITY) : 0.f;
}
// one 16-byte piece (4 consecutive d of code k, piece c of its row) scaled and split into f16 hi + lo at its place in the
// A-fragment planes [ct][k step][64 lanes][8 halves]
__device__ __forceinline__ void vq_plane_piece(vq_f32x4 pv, float sw, int k, int c, unsigned char* wh, unsigned char* wlo) {
  const int ct = k >> 5, i = k & 31, kc = c >> 2, h = (c & 3) >> 1;
  vq_h4 hi, lo;
#pragma unroll
  for (int jj = 0; jj < 4; jj++) {
    const float v = pv[jj] * sw;
    hi[jj] = (_Float16)v;
    lo[jj] = (_Float16)(v - (float)hi[jj]);
  }
  const size_t off = ((((size_t)ct * 4 + kc) * 64 + i + 32 * h) * 8 + 4 * (c & 1)) * 2;
  *reinterpret_cast<vq_h4*>(wh + off) = hi;
  *reinterpret_cast<vq_h4*>(wlo + off) = lo;
}
// Image layout (bytes; KT = 32-code tiles, an even number): hi plane [KT * 4096] | lo plane [KT * 4096] | w2s | usw | sws | wmx
// (KT * 32 floats each).  Built ONCE per codebook update (the EMA blend, load_state_dict, any write to the codebook) instead of
// by every one of the search's 250 workgroups on every call (11 - 15 us of its 30).
struct VqImgQ { const float* cb; unsigned char* img; int K; };
struct VqImgM { VqImgQ q[4]; };
__global__ __launch_bounds__(256) void vq_image_kernel(const VqImgM m) {
  __shared__ float wimg[32 * VQH_WS];
  __shared__ float sw_s[32];
  const VqImgQ e = m.q[blockIdx.y];
  const int K = e.K, KT = ((K + 63) >> 6) * 2, ct = blockIdx.x, tid = threadIdx.x;
  if (ct >= KT) return;  // (the grid spans the largest codebook of the call)
  vq_f32x4 pv[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int pi = tid + 256 * j, k = ct * 32 + (pi >> 4);
    pv[j] = k < K ? *reinterpret_cast<const vq_f32x4*>(e.cb + (size_t)k * 64 + 4 * (pi & 15)) : vq_f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<vq_f32x4*>(wimg + (size_t)(pi >> 4) * VQH_WS + 4 * (pi & 15)) = pv[j];
  }
  __syncthreads();
  float* tabs = reinterpret_cast<float*>(e.img + (size_t)KT * 8192);
  if (tid < 32) {
    const int k = ct * 32 + tid;
    float w2, us, sw, wm;
    vq_code_stats(wimg + (size_t)tid * VQH_WS, k, K, w2, us, sw, wm);
    tabs[k] = w2; tabs[KT * 32 + k] = us; tabs[2 * KT * 32 + k] = sw; tabs[3 * KT * 32 + k] = wm;
    sw_s[tid] = sw;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int pi = tid + 256 * j;
    vq_plane_piece(pv[j], sw_s[pi >> 4], ct * 32 + (pi >> 4), pi & 15, e.img, e.img + (size_t)KT * 4096);
  }
}

template <int TP>
__global__ __launch_bounds__(256 * TP, 1) void vq_forward_f16_kernel(const float* __restrict__ x, int ldx,
                                                                     const float* __restrict__ cb, int N, int K,
                                                                     long long* __restrict__ idx_out, float* __restrict__ e_out,
                                                                     int lde, float* __restrict__ qx_out, int ldq, const VqFuse fz) {
  constexpr int D = 64, NT = 256 * TP;
  __shared__ float vq_red[8 * TP];
  __shared__ float vq_wmax[4 * TP];
  __shared__ float vq_m[TP][3][VQM_FB];
  __shared__ int vq_i[TP][2][VQM_FB];
  __shared__ float vq_xrow[4 * TP][D];  // full-scan fallback: the frame's row, one slot per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int KT = ((K + 63) >> 6) * 2;               // 32-code tiles, an even number of them
  unsigned char* wh = smem;                          // [KT][4 k steps][64 lanes][8 halves]  hi plane (A fragments)
  unsigned char* wlo = smem + (size_t)KT * 4096;     // lo plane
  float* wimg = reinterpret_cast<float*>(smem);      // transient: [KT * 32][VQH_WS] fp32 image (overlaps the planes)
  float* w2s = reinterpret_cast<float*>(smem + (size_t)KT * 32 * VQH_WS * 4);  // [KT * 32] behind the image
  float* usw = w2s + KT * 32;                                                     // [KT * 32] -2 * 2^-ew_k
  float* sws = usw + KT * 32;                                                     // [KT * 32] 2^ew_k
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), fg = wave & 3, tp = wave >> 2;
#ifdef VQ_PROF
  const unsigned long long vq_t0_ = __builtin_readcyclecounter();
#endif

  // ---- this lane's frame: the whole row once (x2 in d order, largest magnitude) ----
  const long n = (long)blockIdx.x * VQM_FB + fg * 32 + l31;
  const bool valid = n < N;
  vq_f32x4 row[16];
  float w2max;
  if (fz.img) {
    // ---- the prepared image: planes and per-code tables are copied, nothing is derived ----
    constexpr int NPI = 16 / TP * 2;  // 16-byte pieces per thread for K = 512 (KT * 512 / NT)
    const vq_f32x4* src = reinterpret_cast<const vq_f32x4*>(fz.img);
    const float* tabs = reinterpret_cast<const float*>(fz.img + (size_t)KT * 8192);
    // ---- the workgroup's 128 frames (x, + add) as ONE coalesced pass through LDS (the planes' area, before the image lands
    // there): a lane that walks its own 256-byte row touches 64 cache lines per load instruction, and the row of a frame is
    // wanted by four lanes (two half-lanes x two code-tile waves) - 16 + 16 such instructions per lane were 9 - 18 k of the
    // call's 40 - 56 k cycles (profiles/round5_vq_phase_cycles.txt).  Here: 4 (+ 4) fully coalesced pieces per thread, requested
    // FIRST (the memory counter retires in order: whatever is consumed first must be asked for first), the image behind them.
    float* xt = reinterpret_cast<float*>(smem);  // [VQM_FB][VQH_WS]
    constexpr int NXP = VQM_FB * 16 / NT;
    vq_f32x4 xv[NXP], av[NXP];
#pragma unroll
    for (int i = 0; i < NXP; i++) {
      const int pidx = tid + NT * i, f = pidx >> 4, c = pidx & 15;
      const long nf = (long)blockIdx.x * VQM_FB + f;
      xv[i] = nf < N ? *reinterpret_cast<const vq_f32x4*>(x + nf * (long)ldx + 4 * c) : vq_f32x4{0.f, 0.f, 0.f, 0.f};
      av[i] = (fz.add && nf < N) ? *reinterpret_cast<const vq_f32x4*>(fz.add + nf * (long)fz.ldadd + 4 * c) : vq_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    vq_f32x4 pc[NPI];
#pragma unroll
    for (int j = 0; j < NPI; j++) { const int i = tid + NT * j; pc[j] = i < KT * 512 ? src[i] : vq_f32x4{0.f, 0.f, 0.f, 0.f}; }
    float tb[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { const int i = tid + NT * j; tb[j] = i < 3 * KT * 32 ? tabs[i] : 0.f; }
    constexpr int NWM = (16 * 32 + NT - 1) / NT;  // KT <= 16
    float wm[NWM];
#pragma unroll
    for (int j = 0; j < NWM; j++) { const int k = tid + NT * j; wm[j] = k < KT * 32 ? tabs[3 * KT * 32 + k] : 0.f; }
#pragma unroll
    for (int i = 0; i < NXP; i++) {
      const int pidx = tid + NT * i;
      if (fz.add) xv[i] += av[i];  // (the sum the search takes, element for element: x + add)
      // xsum leaves HERE when it may (it aliases neither input): the store drains under the search, which touches no memory,
      // instead of doubling the gather epilogue's 32 KB of stores per workgroup (the epilogue is store-issue bound)
      if (fz.xsum_early) {
        const long nf = (long)blockIdx.x * VQM_FB + (pidx >> 4);
        if (nf < N) *reinterpret_cast<vq_f32x4*>(fz.xsum + nf * (long)fz.ldsum + 4 * (pidx & 15)) = xv[i];
      }
      *reinterpret_cast<vq_f32x4*>(xt + (size_t)(pidx >> 4) * VQH_WS + 4 * (pidx & 15)) = xv[i];
    }
    __syncthreads();
    {
      const float* xr = xt + (size_t)(fg * 32 + l31) * VQH_WS;
#pragma unroll
      for (int q = 0; q < 16; q++) row[q] = *reinterpret_cast<const vq_f32x4*>(xr + 4 * q);
    }
    __syncthreads();  // every row is in registers: the image may overwrite the tile
    float wmx = 0.f;
#pragma unroll
    for (int j = 0; j < NWM; j++) wmx = fmaxf(wmx, wm[j]);
#pragma unroll
    for (int j = 0; j < NPI; j++) { const int i = tid + NT * j; if (i < KT * 512) reinterpret_cast<vq_f32x4*>(wh)[i] = pc[j]; }
#pragma unroll
    for (int j = 0; j < 3; j++) { const int i = tid + NT * j; if (i < 3 * KT * 32) w2s[i] = tb[j]; }  // (w2s | usw | sws are contiguous)
    for (int i = 3 * NT + tid; i < 3 * KT * 32; i += NT) w2s[i] = tabs[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmx = fmaxf(wmx, __shfl_xor(wmx, o, 64));
    if (lane == 0) vq_wmax[wave] = wmx;
    __syncthreads();
    w2max = vq_wmax[0];
#pragma unroll
    for (int w = 1; w < 4 * TP; w++) w2max = fmaxf(w2max, vq_wmax[w]);
  } else {
  {
    const float* xp = x + (valid ? n : 0) * (long)ldx;
#pragma unroll
    for (int q = 0; q < 16; q++) row[q] = *reinterpret_cast<const vq_f32x4*>(xp + 4 * q);
    if (fz.add) {
      const float* ap = fz.add + (valid ? n : 0) * (long)fz.ldadd;
#pragma unroll
      for (int q = 0; q < 16; q++) row[q] += *reinterpret_cast<const vq_f32x4*>(ap + 4 * q);
    }
  }
  // ---- codebook pieces (coalesced, 16 bytes each) -> registers; fp32 image -> LDS for the squared norms ----
  constexpr int NPV = 16 / TP * 2;  // pieces per thread for K = 512 (KT * 32 * 16 / NT)
  vq_f32x4 pv[NPV];
#pragma unroll
  for (int j = 0; j < NPV; j++) {
    const int pi = tid + NT * j, k = pi >> 4;
    pv[j] = (pi < KT * 32 * 16 && k < K) ? *reinterpret_cast<const vq_f32x4*>(cb + (size_t)pi * 4) : vq_f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < NPV; j++) {
    const int pi = tid + NT * j, k = pi >> 4, c = pi & 15;
    if (pi < KT * 32 * 16) *reinterpret_cast<vq_f32x4*>(wimg + (size_t)k * VQH_WS + 4 * c) = pv[j];
  }
  __syncthreads();
  float wmx = 0.f;
  for (int k = tid; k < KT * 32; k += NT) {  // squared norm in d order: the exact kernels' chain; the code's scale
    float wm;
    vq_code_stats(wimg + (size_t)k * VQH_WS, k, K, w2s[k], usw[k], sws[k], wm);
    wmx = fmaxf(wmx, wm);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmx = fmaxf(wmx, __shfl_xor(wmx, o, 64));
  if (lane == 0) vq_wmax[wave] = wmx;
  __syncthreads();  // every read of the fp32 image done; the workgroup's largest squared norm visible
  w2max = vq_wmax[0];
#pragma unroll
  for (int w = 1; w < 4 * TP; w++) w2max = fmaxf(w2max, vq_wmax[w]);
  {
    float swk[NPV];  // (read before the planes overwrite nothing they need: sws lives behind the image)
#pragma unroll
    for (int j = 0; j < NPV; j++) { const int pi = tid + NT * j; swk[j] = pi < KT * 32 * 16 ? sws[pi >> 4] : 1.f; }
#pragma unroll
    for (int j = 0; j < NPV; j++) {
      const int pi = tid + NT * j;
      if (pi < KT * 32 * 16) vq_plane_piece(pv[j], swk[j], pi >> 4, pi & 15, wh, wlo);
    }
  }
  }
  const bool w_ok = w2max < INFINITY;
  float x2 = 0.f, xmx = 0.f;  // the row's squared norm in d order (the pinned chain of the exact kernels), its largest magnitude
#pragma unroll
  for (int q = 0; q < 16; q++) {
#pragma unroll
    for (int j = 0; j < 4; j++) { x2 = vq_sq_acc(x2, row[q][j]); xmx = fmaxf(xmx, fabsf(row[q][j])); }
  }
  // ---- the frame's B fragments: k step kc holds d = 16 kc + 8 half .. + 7 ----
  const int ex_raw = 10 - vq_expo(xmx);
  const bool x_ok = x2 < INFINITY && (xmx == 0.f || (ex_raw >= -60 && ex_raw <= 60));  // (NaN / Inf rows fail x2 < inf)
  const int ex = (x_ok && xmx > 0.f) ? ex_raw : 0;
  vq_h8 xh[4], xl[4];
  {
    const float sx = vq_pow2(ex);
#pragma unroll
    for (int kc = 0; kc < 4; kc++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int d = 16 * kc + 8 * half + j;  // (half is per lane: select between the two candidate registers)
        const float v0 = row[(16 * kc + j) >> 2][(16 * kc + j) & 3], v1 = row[(16 * kc + 8 + j) >> 2][(16 * kc + 8 + j) & 3];
        const float v = (half ? v1 : v0) * sx;
        (void)d;
        const _Float16 hi = (_Float16)v;
        xh[kc][j] = hi;
        xl[kc][j] = (_Float16)(v - (float)hi);
      }
  }
  const float sxinv = vq_pow2(-ex);  // d~_k = w2_k + (usw_k * 2^-ex) (accA + accB)
  __syncthreads();  // planes complete
  VQ_T(0)

  VqTop3 t;
  t.m1 = t.m2 = t.m3 = INFINITY;
  t.i1 = t.i2 = 0x7fffffff;
  const unsigned char* whl = wh + lane * 16;
  const unsigned char* wll = wlo + lane * 16;
// the 12 MFMAs of code tile ct; the A fragments of the wave's next tile are read into (hn, ln) meanwhile
#define VQH_TILE(aA, aB, ct, hc, lc, hn, ln)                                                                   \
  {                                                                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; r++) { aA[r] = 0.f; aB[r] = 0.f; }                               \
    const size_t to_ = (size_t)((ct) + TP < KT ? (ct) + TP : (ct)) * 4096;                                     \
    _Pragma("unroll") for (int kc = 0; kc < 4; kc++) {                                                         \
      hn[kc] = *reinterpret_cast<const vq_h8*>(whl + to_ + kc * 1024);                                         \
      ln[kc] = *reinterpret_cast<const vq_h8*>(wll + to_ + kc * 1024);                                         \
    }                                                                                                          \
    _Pragma("unroll") for (int kc = 0; kc < 4; kc++) {                                                         \
      aA = __builtin_amdgcn_mfma_f32_32x32x16_f16(hc[kc], xh[kc], aA, 0, 0, 0);                                \
      aB = __builtin_amdgcn_mfma_f32_32x32x16_f16(lc[kc], xh[kc], aB, 0, 0, 0);                                \
      aB = __builtin_amdgcn_mfma_f32_32x32x16_f16(hc[kc], xl[kc], aB, 0, 0, 0);                                \
    }                                                                                                          \
  }
// this lane's 16 codes of tile ct.  The candidate's place (the wave's tile ordinal, 3 bits | its slot in the lane, 4 bits)
// replaces the 7 low mantissa bits of its value: the three smallest KEYS are tracked with two med3 and a min per value (5 VALU
// operations per value with the packed add / fma, against 11.5 for values and indices kept apart - the compare / select chain
// was twice the MFMA time of the loop), and the key moves the value by less than 2^-16 of its magnitude, which the decision
// adds to its threshold (VQH_KEY_EPS)
#define VQH_PICK(aA, aB, ct, tord) { const unsigned tagb_ = (unsigned)(tord) << 4; VQH_PICK_(aA, aB, ct) }
#define VQH_PICK_(aA, aB, ct)                                                                                  \
  _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                              \
    const vq_f32x4 w2q = *reinterpret_cast<const vq_f32x4*>(w2s + (ct) * 32 + 8 * q + 4 * half);               \
    const vq_f32x4 usq = *reinterpret_cast<const vq_f32x4*>(usw + (ct) * 32 + 8 * q + 4 * half) * sxinv;       \
    _Pragma("unroll") for (int j2 = 0; j2 < 2; j2++) {                                                         \
      const vq_f32x2 acc = vq_f32x2{aA[4 * q + 2 * j2], aA[4 * q + 2 * j2 + 1]} +                              \
                           vq_f32x2{aB[4 * q + 2 * j2], aB[4 * q + 2 * j2 + 1]};                               \
      const vq_f32x2 v2 = __builtin_elementwise_fma(acc, vq_f32x2{usq[2 * j2], usq[2 * j2 + 1]},               \
                                                    vq_f32x2{w2q[2 * j2], w2q[2 * j2 + 1]});                   \
      const float vs_[2] = {v2.x, v2.y};  /* (scalars first: a bit_cast of v2[jj] reads element 0 for both) */    \
      _Pragma("unroll") for (int jj = 0; jj < 2; jj++) {                                                       \
        unsigned tag = tagb_ | (unsigned)(4 * q + 2 * j2 + jj);                                                \
        asm volatile("" : "+s"(tag));  /* the whole tag in one SGPR: (v & mask) | tag is ONE v_and_or_b32 */    \
        const float key = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, vs_[jj]) & keymask) | tag);  \
        t.m3 = __builtin_amdgcn_fmed3f(key, t.m2, t.m3);                                                       \
        t.m2 = __builtin_amdgcn_fmed3f(key, t.m1, t.m2);                                                       \
        t.m1 = vq_min_raw(key, t.m1);                                                                          \
      }                                                                                                        \
    }                                                                                                          \
  }
// (how the MFMAs and the candidate chain interleave is left to the compiler: hand-placed groups of one MFMA + n VALU
// operations, MFMAs first, three accumulator chains and scalar instead of packed arithmetic all measured within 4 % of each
// other - profiles/round5_vq_packed_keys.txt)
#define VQH_PAIR(nA, nB, ctn, hc, lc, hn, ln, pA, pB, ctp, tordp)                                              \
  VQH_TILE(nA, nB, ctn, hc, lc, hn, ln)                                                                        \
  VQH_PICK(pA, pB, ctp, tordp)                                                                                 \
  __builtin_amdgcn_sched_barrier(0);
  unsigned keymask = ~0x7fu;  // in a VGPR: v_and_or_b32 takes ONE scalar operand, and that is the tag
  asm volatile("" : "+v"(keymask));
  f32x16 a0A, a0B, a1A, a1B;
  vq_h8 hA[4], lA[4], hB[4], lB[4];
#pragma unroll
  for (int kc = 0; kc < 4; kc++) {
    hA[kc] = *reinterpret_cast<const vq_h8*>(whl + (size_t)tp * 4096 + kc * 1024);
    lA[kc] = *reinterpret_cast<const vq_h8*>(wll + (size_t)tp * 4096 + kc * 1024);
  }
  VQH_TILE(a0A, a0B, tp, hA, lA, hB, lB)
  __builtin_amdgcn_sched_barrier(0);
  int tord = 0;  // ordinal of the wave's tile: ct = tord * TP + tp (at most 8: KT <= 16, TP = 2)
  for (int ct = tp; ct < KT - 2 * TP; ct += 2 * TP, tord += 2) {
    VQH_PAIR(a1A, a1B, ct + TP, hB, lB, hA, lA, a0A, a0B, ct, tord)
    VQH_PAIR(a0A, a0B, ct + 2 * TP, hA, lA, hB, lB, a1A, a1B, ct + TP, tord + 1)
  }
  VQH_PAIR(a1A, a1B, KT - TP + tp, hB, lB, hA, lA, a0A, a0B, KT - 2 * TP + tp, tord)
  VQH_PICK(a1A, a1B, KT - TP + tp, tord + 1)
#undef VQH_PAIR
#undef VQH_TILE
#undef VQH_PICK
  // ---- the codes of the two smallest keys (a lane sees 16 x KT / TP >= 32 values: all three are keys) ----
  {
    const unsigned k1 = __builtin_bit_cast(unsigned, t.m1) & 0x7fu, k2 = __builtin_bit_cast(unsigned, t.m2) & 0x7fu;
    t.i1 = (int)((k1 >> 4) * TP + tp) * 32 + (int)(k1 & 3) + 8 * (int)((k1 >> 2) & 3) + 4 * half;
    t.i2 = (int)((k2 >> 4) * TP + tp) * 32 + (int)(k2 & 3) + 8 * (int)((k2 >> 2) & 3) + 4 * half;
  }
  // ---- the other half-wave holds the other 16 codes per tile of the same frame; the other wave(s) the other tiles ----
  {
    const float b1 = __shfl_xor(t.m1, 32, 64), b2 = __shfl_xor(t.m2, 32, 64), b3 = __shfl_xor(t.m3, 32, 64);
    const int j1 = __shfl_xor(t.i1, 32, 64), j2 = __shfl_xor(t.i2, 32, 64);
    vq_top3_merge(t, b1, b2, b3, j1, j2);
  }
  if (TP > 1) {
    if (half == 0) {
      const int f = fg * 32 + l31;
      vq_m[tp][0][f] = t.m1; vq_m[tp][1][f] = t.m2; vq_m[tp][2][f] = t.m3;
      vq_i[tp][0][f] = t.i1; vq_i[tp][1][f] = t.i2;
    }
    __syncthreads();
    const int f = fg * 32 + l31, o = tp ^ (TP - 1);
    vq_top3_merge(t, vq_m[o][0][f], vq_m[o][1][f], vq_m[o][2][f], vq_i[o][0][f], vq_i[o][1][f]);
  }
  VQ_T(3)
  // ---- decide ----
  // thr = delta_i1 + the largest delta of a code that can still win (header comment)
  const float xn = sqrtf(x2);
  const float w2a = w2s[(t.i1 >= 0 && t.i1 < KT * 32) ? t.i1 : 0];
  const float d1 = 2.0e-5f * xn * sqrtf(w2a) + 5.0e-7f * (x2 + w2a);
  // (m1, m2, m3 are keys: each within VQH_KEY_EPS of its magnitude of the value d~ the bounds are written for - or, for a
  // value in the subnormal range, within 127 subnormal steps = 1.8e-43: VQH_KEY_ABS covers that with the smallest normal number.
  // It matters for an all-zero frame in front of two all-zero codes: every d~ is 0, the keys are the bare tags, thr underflows
  // to 0, and without the absolute term the smaller TAG would win "for sure" instead of the lower index after the exact tie)
  const float p1 = VQH_KEY_EPS * fabsf(t.m1) + VQH_KEY_ABS;
  const float Rr = 2.001f * xn + sqrtf(fmaxf(t.m1 + p1 + d1, 0.f));
  const float wn = fminf(sqrtf(w2max), Rr * 1.001f);
  const float thr = d1 + 2.0e-5f * xn * wn + 5.0e-7f * (x2 + wn * wn);
  const bool trust = x_ok && w_ok && t.i1 < K;
  const bool sure = trust && (t.m2 - t.m1 > thr + p1 + (VQH_KEY_EPS * fabsf(t.m2) + VQH_KEY_ABS));
  const bool two = trust && !sure && (t.m3 - t.m1 > thr + p1 + (VQH_KEY_EPS * fabsf(t.m3) + VQH_KEY_ABS));
  const bool full = !sure && !two;
  int besti = t.i1;
  if (__builtin_amdgcn_ballot_w64(two)) {
    // exact chain for code i1 (half-0 lane of the frame) and i2 (half-1 lane): fma over d ascending from 0, then
    // (w2 - 2 dot) + x2 - the expression of vq_forward_mfma_kernel / the VALU kernels
    float dist = INFINITY;
    const int kc_ = half ? t.i2 : t.i1;
    if (two && kc_ < K) {
      const float* wp = cb + (size_t)kc_ * D;
      const float* xp = x + (valid ? n : 0) * (long)ldx;  // (the row again: 64 registers are not kept across the search)
      const float* ap = fz.add ? fz.add + (valid ? n : 0) * (long)fz.ldadd : nullptr;
      float dot = 0.f;
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const vq_f32x4 wv = *reinterpret_cast<const vq_f32x4*>(wp + 4 * q);
        vq_f32x4 xv = *reinterpret_cast<const vq_f32x4*>(xp + 4 * q);
        if (ap) xv += *reinterpret_cast<const vq_f32x4*>(ap + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; j++) dot = __builtin_fmaf(wv[j], xv[j], dot);
      }
      dist = (w2s[kc_] - 2.f * dot) + x2;
    }
    const float od = __shfl_xor(dist, 32, 64);
    const int oi = __shfl_xor(kc_, 32, 64);
    if (two) {
      const bool other = od < dist || (od == dist && oi < kc_);
      besti = other ? oi : kc_;
      if (!(dist < INFINITY) && !(od < INFINITY)) besti = 0x7fffffff;  // (cannot happen for finite rows; keeps the NaN rule)
    }
  }
  unsigned long long fm = __builtin_amdgcn_ballot_w64(full && half == 0);
  while (fm) {  // full exact scan, one frame at a time, lane = code (rare)
    const int f = __builtin_ctzll(fm);
    fm &= fm - 1;
    // the frame's row -> this wave's LDS slot (lanes 0-15 fetch one 16-byte piece each)
    {
      const long nf = (long)blockIdx.x * VQM_FB + fg * 32 + f;
      if (lane < 16) {
        vq_f32x4 xv = *reinterpret_cast<const vq_f32x4*>(x + (nf < N ? nf : 0) * (long)ldx + 4 * lane);
        if (fz.add) xv += *reinterpret_cast<const vq_f32x4*>(fz.add + (nf < N ? nf : 0) * (long)fz.ldadd + 4 * lane);
        *reinterpret_cast<vq_f32x4*>(&vq_xrow[wave][4 * lane]) = xv;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    const float fx2 = __shfl(x2, f, 64);
    float bd = INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
      const float* wp = cb + (size_t)k * D;
      float dot = 0.f;
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const vq_f32x4 wv = *reinterpret_cast<const vq_f32x4*>(wp + 4 * q);
        const vq_f32x4 xv = *reinterpret_cast<const vq_f32x4*>(&vq_xrow[wave][4 * q]);
#pragma unroll
        for (int j = 0; j < 4; j++) dot = __builtin_fmaf(wv[j], xv[j], dot);
      }
      const float dist = (w2s[k] - 2.f * dot) + fx2;
      if (dist < bd) { bd = dist; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float od = __shfl_xor(bd, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (l31 == f) besti = bi;  // (both half lanes of the frame)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the slot is rewritten by the next frame
  }
  if (besti >= K) besti = 0;  // all-NaN row: torch.argmin would pick a NaN slot; pin to 0 like the other kernels
  VQ_T(1)
  if (valid && half == 0 && tp == 0) {
    idx_out[n] = (long long)besti;
    if (!sure) atomicAdd(&vq_f16_flag_counts[two ? 1 : 2], 1ull);
  }
  // ---- gathered code vectors and the straight-through value: as vq_forward_mfma_kernel, with every load of the wave's
  // 8 / TP row groups (code vector, x, add, mask byte) requested before the first is used - a load behind a per-group branch on
  // the frame's validity and its mask byte was one memory round trip per group, in series ----
  float csum = 0.f, ccnt = 0.f;
  {
    constexpr int NG = 8 / TP;  // (the two waves of a frame group split its rows)
    const int sub = lane >> 4, c4 = (lane & 15) * 4;
    const long nw = (long)blockIdx.x * VQM_FB + fg * 32;
    const bool want_x = qx_out || (fz.xsum && !fz.xsum_early) || fz.cpart;
    long nfs[NG];
    float4 ev[NG], xs[NG], as[NG];
    unsigned char mk[NG];
#pragma unroll
    for (int u = 0; u < NG; u++) {
      const int f = 4 * (tp * NG + u) + sub;
      const int bi = __shfl(besti, f, 64);
      nfs[u] = nw + f;
      const long nc = nfs[u] < N ? nfs[u] : 0;  // (a frame past the end reads frame 0 and stores nothing)
      ev[u] = *reinterpret_cast<const float4*>(cb + (size_t)bi * D + c4);
      xs[u] = want_x ? *reinterpret_cast<const float4*>(x + nc * (long)ldx + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      as[u] = (want_x && fz.add) ? *reinterpret_cast<const float4*>(fz.add + nc * (long)fz.ldadd + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      mk[u] = (fz.cpart && fz.mask) ? fz.mask[nc] : (unsigned char)1;
    }
#pragma unroll
    for (int u = 0; u < NG; u++) {
      const long nf = nfs[u];
      if (nf < N) {
        const float4 e = ev[u];
        if (e_out) *reinterpret_cast<float4*>(e_out + nf * (long)lde + c4) = e;
        float4 xv = xs[u];
        if (want_x && fz.add) {  // the same sum as the search took, element for element
          xv.x += as[u].x; xv.y += as[u].y; xv.z += as[u].z; xv.w += as[u].w;
        }
        if (fz.xsum && !fz.xsum_early) *reinterpret_cast<float4*>(fz.xsum + nf * (long)fz.ldsum + c4) = xv;
        if (fz.cpart && mk[u]) {
          const float d0 = xv.x - e.x, d1 = xv.y - e.y, d2 = xv.z - e.z, d3 = xv.w - e.w;
          csum += d0 * d0; csum += d1 * d1; csum += d2 * d2; csum += d3 * d3;
          ccnt += 4.f;
        }
        if (qx_out) {
          float4 o;
          o.x = xv.x + (e.x - xv.x);
          o.y = xv.y + (e.y - xv.y);
          o.z = xv.z + (e.z - xv.z);
          o.w = xv.w + (e.w - xv.w);
          *reinterpret_cast<float4*>(qx_out + nf * (long)ldq + c4) = o;
        }
      }
    }
  }
  if (fz.cpart) {  // (uniform)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { csum += __shfl_xor(csum, o, 64); ccnt += __shfl_xor(ccnt, o, 64); }
    if (lane == 0) { vq_red[wave] = csum; vq_red[4 * TP + wave] = ccnt; }
    __syncthreads();
    if (tid == 0) {
      float a = vq_red[0], c = vq_red[4 * TP];
#pragma unroll
      for (int w = 1; w < 4 * TP; w++) { a += vq_red[w]; c += vq_red[4 * TP + w]; }
      fz.cpart[2 * blockIdx.x] = a; fz.cpart[2 * blockIdx.x + 1] = c;
    }
  }
  VQ_T(2)
}

static int vq_mfma_attrs() {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vq_forward_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096) != hipSuccess ||
        hipFuncSetAttribute((const void*)vq_forward_mfma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096) != hipSuccess ||
        hipFuncSetAttribute((const void*)vq_forward_f16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 12288) != hipSuccess)
      return CRK_ERR_HIP;
    attr_set = true;
  }
  return CRK_OK;
}
// the split-f16 search (default; CRK_VQ_F16=0: the fp32-MFMA search, for A/B runs and the equality test)
static int vq_f16_mode = -1;
extern "C" int crk_debug_vq_set_f16(int on) { vq_f16_mode = on ? 1 : 0; return 0; }  // (tests: both searches in one process)
static bool vq_use_f16(int kt) {
  if (vq_f16_mode < 0) vq_f16_mode = crk_sw().vq_f16;
  return vq_f16_mode != 0 && kt % 4 == 0;
}
// eight waves where the tile count splits evenly between two waves per frame group
static void vq_mfma_launch(int nblk, int kt, size_t lds, hipStream_t s, const float* x, int ldx, const float* cb, int N, int K,
                           long long* idx, float* e, int lde, float* qx, int ldq, const VqFuse& fz) {
  const int tp_env = 2;
  // algorithmic bytes of a quantizer call = N x (64 x 4 read + 8 index + 64 x 4 gathered code) = 520 B per frame
  conv_prof_bytes(7, 520.0 * N);
  conv_prof_begin(7, 2.0 * N * (double)K * 64.0, s);
  if (vq_use_f16(kt)) {
    const size_t lds16 = (size_t)kt * 32 * VQH_WS * 4 + (size_t)kt * 32 * 4 * 3;  // fp32 image (the f16 planes reuse it) + 3 per-code tables
    hipLaunchKernelGGL(vq_forward_f16_kernel<2>, dim3(nblk), dim3(512), lds16, s, x, ldx, cb, N, K, idx, e, lde, qx, ldq, fz);
    conv_prof_end(7, s);
    return;
  }
  if (tp_env == 2 && kt % 4 == 0)
    hipLaunchKernelGGL(vq_forward_mfma_kernel<2>, dim3(nblk), dim3(512), lds, s, x, ldx, cb, N, K, idx, e, lde, qx, ldq, fz);
  else
    hipLaunchKernelGGL(vq_forward_mfma_kernel<1>, dim3(nblk), dim3(256), lds, s, x, ldx, cb, N, K, idx, e, lde, qx, ldq, fz);
  conv_prof_end(7, s);
}

__global__ __launch_bounds__(256) void vq_commit_final_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ out) {
  __shared__ float sh[8];
  float s = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) { s += part[2 * i]; c += part[2 * i + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; sh[4 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float st = ((sh[0] + sh[1]) + sh[2]) + sh[3], ct = ((sh[4] + sh[5]) + sh[6]) + sh[7];
    out[0] = st / ct; out[1] = ct;  // 0/0 -> NaN like the mean of an empty selection
  }
}

// crk_vq_forward with the quantizer's surroundings in the same launch (D = 64, K <= 512 only - CRK_ERR_UNSUPPORTED
// otherwise, the caller then composes the separate entry points):
//   add (NULL: none): the input is x + add, written to xsum (may be NULL);
//   commit_out2 (NULL: none): {mean over the frames mask selects (NULL: all) of (input - e)^2, element count} - what
//   crk_masked_loss_fwd(input, e, mask, mode 1) returns; scratch: crk_loss_scratch_floats() floats.
// Codebook image of the split-f16 search (D = 64, K <= 512): bytes, and the build - one launch for up to 4 codebooks (the
// quantizers of one generator forward).  The caller owns the buffer and its validity: rebuild after ANY write to the codebook
// (the EMA blend entry points do not know about images).  crk_vq_forward_fused(image != NULL) then copies the image instead of
// deriving it in every workgroup; the indices are the same bit for bit (the same device functions produce both).
extern "C" long long crk_vq_image_bytes(int K, int D) {
  if (D != 64 || K <= 0 || K > 512) return 0;
  const long long kt = ((K + 63) / 64) * 2;
  return kt * 8192 + kt * 32 * 4 * 4;
}
extern "C" int crk_vq_image_build_multi(int nq, const float* const* codebooks, const int* K, int D, void* const* images, void* stream) {
  if (nq < 1 || nq > 4 || !codebooks || !K || !images || D != 64) return CRK_ERR_ARG;
  VqImgM m;
  int ktmax = 0;
  for (int i = 0; i < nq; i++) {
    if (!codebooks[i] || !images[i] || K[i] <= 0 || K[i] > 512) return CRK_ERR_ARG;
    m.q[i].cb = codebooks[i]; m.q[i].img = (unsigned char*)images[i]; m.q[i].K = K[i];
    const int kt = ((K[i] + 63) / 64) * 2;
    if (kt > ktmax) ktmax = kt;
  }
  hipLaunchKernelGGL(vq_image_kernel, dim3(ktmax, nq), dim3(256), 0, (hipStream_t)stream, m);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

extern "C" int crk_vq_forward_fused(const float* x, int ldx, const float* add, int ldadd, float* xsum, int ldsum,
                                    const float* codebook, int N, int D, int K, long long* idx, float* e, int lde, float* qx,
                                    int ldq, const unsigned char* mask, float* commit_out2, float* scratch, const void* image,
                                    void* stream) {
  if (!x || !codebook || !idx || N <= 0 || K <= 0) return CRK_ERR_ARG;
  if ((ldx & 3) || (lde & 3) || (ldq & 3) || (add && (ldadd & 3)) || (xsum && (ldsum & 3))) return CRK_ERR_ARG;
  if (commit_out2 && !scratch) return CRK_ERR_ARG;
  const int lc_env = crk_sw().vq_lc;  // (0: frame-per-lane kernel for every shape, 1: code-per-lane, 2: MFMA)
  const int nblk = (N + VQM_FB - 1) / VQM_FB;
  if (lc_env != 2 || D != 64 || K > 512 || nblk > 1024) return CRK_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (vq_mfma_attrs() != CRK_OK) return CRK_ERR_HIP;
  const int kt = ((K + 63) / 64) * 2;
  const size_t lds = (size_t)kt * 32 * (D + 1) * 4;
  VqFuse fz; fz.add = add; fz.ldadd = ldadd; fz.xsum = xsum; fz.ldsum = ldsum; fz.mask = mask; fz.cpart = commit_out2 ? scratch : nullptr;
  fz.img = (const unsigned char*)image;
  {
    // (a caller may sum in place, xsum == x or == add: then the epilogue's second read of the inputs must still see them)
    auto overlaps = [&](const float* p, int ld) {
      if (!p) return false;
      const char* a0 = (const char*)xsum; const char* a1 = a0 + (size_t)N * ldsum * 4;
      const char* b0 = (const char*)p; const char* b1 = b0 + (size_t)N * ld * 4;
      return a0 < b1 && b0 < a1;
    };
    fz.xsum_early = (fz.img && xsum && !overlaps(x, ldx) && !overlaps(add, ldadd)) ? 1 : 0;
  }
  vq_mfma_launch(nblk, kt, lds, s, x, ldx, codebook, N, K, idx, e, lde, qx, ldq, fz);
  if (commit_out2) hipLaunchKernelGGL(vq_commit_final_kernel, dim3(1), dim3(256), 0, s, scratch, nblk, commit_out2);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

extern "C" int crk_vq_forward(const float* x, int ldx, const float* codebook, int N, int D, int K, long long* idx,
                              float* e, int lde, float* qx, int ldq, void* stream) {
  if (!x || !codebook || !idx || N <= 0 || K <= 0) return CRK_ERR_ARG;
  if ((ldx & 3) || (lde & 3) || (ldq & 3)) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int lc_env = crk_sw().vq_lc;  // (0: frame-per-lane kernel for every shape, 1: code-per-lane, 2: MFMA)
  if (lc_env == 2 && D == 64 && K <= 512) {
    if (vq_mfma_attrs() != CRK_OK) return CRK_ERR_HIP;
    const int kt = ((K + 63) / 64) * 2;
    const size_t lds = (size_t)kt * 32 * (D + 1) * 4;
    VqFuse fz{};
    vq_mfma_launch((N + VQM_FB - 1) / VQM_FB, kt, lds, s, x, ldx, codebook, N, K, idx, e, lde, qx, ldq, fz);
    CRK_CHECK_LAUNCH();
    return CRK_OK;
  }
  if (lc_env && D == 64 && K <= 512) {
    hipLaunchKernelGGL(vq_forward_lc_kernel<64>, dim3((N + VQL_FB - 1) / VQL_FB), dim3(512), 0, s, x, ldx, codebook, N, K, idx,
                       e, lde, qx, ldq);
    CRK_CHECK_LAUNCH();
    return CRK_OK;
  }
  int kchunk = K;
  const int max_floats = (150 * 1024 - 4 * VQ_FRAMES * 8) / 4;
  while ((size_t)kchunk * (D + 1) > (size_t)max_floats) kchunk = (kchunk + 1) / 2;
  const size_t lds = ((size_t)kchunk * (D + 1) + 4 * VQ_FRAMES * 2) * 4;
  dim3 grid((N + VQ_FRAMES - 1) / VQ_FRAMES), block(512);
#define VQ_LAUNCH(DD)                                                                                          \
  {                                                                                                            \
    static bool attr_set = false;                                                                              \
    if (!attr_set) {                                                                                           \
      if (hipFuncSetAttribute((const void*)vq_forward_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                              160 * 1024) != hipSuccess) return CRK_ERR_HIP;                                   \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    hipLaunchKernelGGL(vq_forward_kernel<DD>, grid, block, lds, s, x, ldx, codebook, N, K, kchunk, idx, e,     \
                       lde, qx, ldq);                                                                          \
  }
  if (D == 64) VQ_LAUNCH(64)
  else if (D == 32) VQ_LAUNCH(32)
  else if (D == 128) VQ_LAUNCH(128)
  else if (D == 16) VQ_LAUNCH(16)
  else return CRK_ERR_UNSUPPORTED;
#undef VQ_LAUNCH
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// ---- EMA statistics (vqvae2.py:316-321): counts[k] = #frames on code k, sums[d][k] = sum of their
// x[:,d] in 2^-28 fixed point (integer sums: exact, order independent, so data-parallel ranks and
// reruns agree bit for bit).  No global atomics: workgroup (chunk, slice) owns a run of frames and
// a slice of SW dims, accumulates an [SW][K] int64 table in LDS, and writes it to its own slot of
// the scratch buffer; a second kernel adds the chunk slots up.
__device__ __forceinline__ void vq_ema_partial_body(const float* __restrict__ x, int ldx, const long long* __restrict__ idx,
                                                    int N, int D, int K, int SW, int frames_per_chunk,
                                                    unsigned long long* __restrict__ part_sums, int* __restrict__ part_counts,
                                                    int chunk, int slice, unsigned char* vq_smem) {
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(vq_smem);  // [SW][K]
  int* cnt = reinterpret_cast<int*>(acc + (size_t)SW * K);                     // [K] (slice 0 only)
  const int tid = threadIdx.x, d0 = slice * SW;
  const bool do_cnt = slice == 0;
  for (int i = tid; i < SW * K; i += 256) acc[i] = 0ull;
  if (do_cnt)
    for (int i = tid; i < K; i += 256) cnt[i] = 0;
  __syncthreads();
  const int qpf = SW >> 2, fpi = 256 / qpf;  // float4 pieces per frame, frames per iteration
  const int q = tid % qpf, fo = tid / qpf;
  const int n0 = chunk * frames_per_chunk, n1 = min(N, n0 + frames_per_chunk);
  const bool dok = d0 + 4 * q < D;
  for (int n = n0 + fo; n < n1; n += fpi) {
    const int k = (int)idx[n];
    if (dok) {
      const float4 f = *reinterpret_cast<const float4*>(x + (long)n * ldx + d0 + 4 * q);
      const float v[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
      for (int j = 0; j < 4; j++)
        atomicAdd(acc + (size_t)(4 * q + j) * K + k, (unsigned long long)__float2ll_rn(v[j] * VQ_FIX_SCALE));
    }
    if (do_cnt && q == 0) atomicAdd(cnt + k, 1);
  }
  __syncthreads();
  const int sw = min(SW, D - d0);
  unsigned long long* dst = part_sums + ((size_t)chunk * D + d0) * K;
  for (int i = tid; i < sw * K; i += 256) dst[i] = acc[i];
  if (do_cnt)
    for (int i = tid; i < K; i += 256) part_counts[(size_t)chunk * K + i] = cnt[i];
}

__global__ __launch_bounds__(256) void vq_ema_partial_kernel(const float* __restrict__ x, int ldx,
                                                             const long long* __restrict__ idx, int N, int D, int K,
                                                             int SW, int frames_per_chunk,
                                                             unsigned long long* __restrict__ part_sums,
                                                             int* __restrict__ part_counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vq_smem[];
  vq_ema_partial_body(x, ldx, idx, N, D, K, SW, frames_per_chunk, part_sums, part_counts, blockIdx.x, blockIdx.y, vq_smem);
}
// the per-chunk tables of several quantizer calls in one launch (grid z = call)
struct EmaPQ { const float* x; const long long* idx; unsigned long long* part_sums; int* part_counts; int ldx, N, D, K, SW, fpc, chunks, slices; };
struct EmaPM { EmaPQ q[4]; int nq; };
__global__ __launch_bounds__(256) void vq_ema_partial_multi_kernel(const EmaPM m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vq_smem[];
  EmaPQ e = m.q[0];  // (statically indexed copies: a dynamic index into the kernel argument would put it into scratch)
#pragma unroll
  for (int k = 1; k < 4; k++)
    if ((int)blockIdx.z == k) e = m.q[k];
  if ((int)blockIdx.x >= e.chunks || (int)blockIdx.y >= e.slices) return;
  vq_ema_partial_body(e.x, e.ldx, e.idx, e.N, e.D, e.K, e.SW, e.fpc, e.part_sums, e.part_counts, blockIdx.x, blockIdx.y, vq_smem);
}

__global__ __launch_bounds__(256) void vq_ema_reduce_kernel(const unsigned long long* __restrict__ part_sums,
                                                            const int* __restrict__ part_counts, int chunks, int DK,
                                                            int K, unsigned long long* __restrict__ sums,
                                                            int* __restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < DK) {
    unsigned long long a = 0ull;
    int c = 0;
    for (; c + 8 <= chunks; c += 8) {  // 8 independent loads in flight (integer sums: any order is exact)
      unsigned long long t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = part_sums[(size_t)(c + u) * DK + i];
#pragma unroll
      for (int u = 0; u < 8; u++) a += t[u];
    }
    for (; c < chunks; c++) a += part_sums[(size_t)c * DK + i];
    sums[i] = a;
  }
  if (i < K) {  // same shape as above: one load per iteration would be `chunks` dependent memory round trips
    int a = 0, c = 0;
    for (; c + 8 <= chunks; c += 8) {
      int t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = part_counts[(size_t)(c + u) * K + i];
#pragma unroll
      for (int u = 0; u < 8; u++) a += t[u];
    }
    for (; c < chunks; c++) a += part_counts[(size_t)c * K + i];
    counts[i] = a;
  }
}

static int vq_ema_plan(int N, int D, int K, int* SW, int* chunks, int* fpc) {
  int sw = 16;
  while (sw >= 4 && (size_t)sw * K * 8 + (size_t)K * 4 > 150 * 1024) sw >>= 1;
  if (sw < 4) return CRK_ERR_UNSUPPORTED;
  int c = (N + 127) / 128;
  if (c > 64) c = 64;
  if (c < 1) c = 1;
  *SW = sw; *chunks = c; *fpc = (N + c - 1) / c;
  return CRK_OK;
}

extern "C" long long crk_vq_ema_scratch_bytes(int N, int D, int K) {
  int sw, c, fpc;
  if (N < 0 || D <= 0 || K <= 0 || vq_ema_plan(N, D, K, &sw, &c, &fpc) != CRK_OK) return -1;
  return (long long)c * ((long long)D * K * 8 + (long long)K * 4);
}

extern "C" int crk_vq_ema_stats(const float* x, int ldx, const long long* idx, int N, int D, int K, int* counts,
                                long long* sums, void* scratch, void* stream) {
  if (!x || !idx || !counts || !sums || !scratch || (D & 3) || (ldx & 3)) return CRK_ERR_ARG;
  int sw, chunks, fpc;
  if (vq_ema_plan(N, D, K, &sw, &chunks, &fpc) != CRK_OK) return CRK_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vq_ema_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            152 * 1024) != hipSuccess)
      return CRK_ERR_HIP;
    attr_set = true;
  }
  unsigned long long* part_sums = reinterpret_cast<unsigned long long*>(scratch);
  int* part_counts = reinterpret_cast<int*>(part_sums + (size_t)chunks * D * K);
  const size_t lds = (size_t)sw * K * 8 + (size_t)K * 4;
  hipLaunchKernelGGL(vq_ema_partial_kernel, dim3(chunks, (D + sw - 1) / sw), dim3(256), lds, s, x, ldx, idx, N, D, K,
                     sw, fpc, part_sums, part_counts);
  const int DK = D * K;
  hipLaunchKernelGGL(vq_ema_reduce_kernel, dim3((DK + 255) / 256), dim3(256), 0, s, part_sums, part_counts, chunks, DK,
                     K, reinterpret_cast<unsigned long long*>(sums), counts);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// decay * a + (1 - decay) * b with ONE fixed rounding sequence (the compiler's choice of which product to fuse
// differs between kernels; the fused and the per-quantizer paths must agree bit for bit)
__device__ __forceinline__ float ema_mix(float decay, float a, float omd, float b) {
  return __fmaf_rn(decay, a, __fmul_rn(omd, b));
}

// ---- EMA apply: cluster sizes in one workgroup (K <= 4096), then the D x K blend ----
__global__ __launch_bounds__(1024) void vq_ema_size_kernel(const int* __restrict__ counts, float* __restrict__ ema_size,
                                                           int K, float decay, float omd, float eps, float keps) {
  __shared__ float red[1024];
  __shared__ float sz[4096];
  const int tid = threadIdx.x;
  float part = 0.f;
  for (int k = tid; k < K; k += 1024) {
    const float v = ema_mix(decay, ema_size[k], omd, (float)counts[k]);
    sz[k] = v;
    part += v;
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float n = red[0];
  const float den = n + keps;
  for (int k = tid; k < K; k += 1024) ema_size[k] = (sz[k] + eps) / den * n;
}

// one thread per (k, d): reads ema_w / sums along k (their fast axis), writes the codebook
// through an LDS transpose so both sides are coalesced
__global__ __launch_bounds__(256) void vq_ema_blend_kernel(const long long* __restrict__ sums,
                                                           const float* __restrict__ ema_size,
                                                           float* __restrict__ ema_w, float* __restrict__ cb, int D,
                                                           int K, float decay, float omd) {
  __shared__ float tile[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int k0 = blockIdx.x * 16, d0 = blockIdx.y * 16;
  const int k = k0 + tx, d = d0 + ty;
  if (k < K && d < D) {
    const int i = d * K + k;
    const float es = (float)sums[i] * VQ_FIX_INV;
    const float w = ema_mix(decay, ema_w[i], omd, es);
    ema_w[i] = w;
    tile[ty][tx] = w / ema_size[k];
  }
  __syncthreads();
  const int kk = k0 + ty, dd = d0 + tx;
  if (kk < K && dd < D) cb[(size_t)kk * D + dd] = tile[tx][ty];
}

// ---- every quantizer of a generator forward at once (one reduce launch, one apply launch) ----
#define VQ_EMA_MAXQ 4
struct EmaQ {
  const unsigned long long* part_sums; const int* part_counts;  // reduce: per-chunk tables (crk_vq_ema_partial)
  int chunks, D, K;
  float keps;                         // (float)(K * eps), the product taken in double like the python scalar
  int* counts; long long* sums;       // integer statistics (caller-owned; all-reduced between reduce and apply)
  float* ema_size; float* ema_w; float* cb;
};
struct EmaMP { EmaQ q[VQ_EMA_MAXQ]; int nq; float decay, omd, eps; };

__global__ __launch_bounds__(256) void vq_ema_reduce_multi_kernel(const EmaMP m) {
  const EmaQ& e = m.q[blockIdx.y];
  const int DK = e.D * e.K, K = e.K, chunks = e.chunks;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < DK) {
    unsigned long long a = 0ull;
    int c = 0;
    for (; c + 8 <= chunks; c += 8) {
      unsigned long long t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = e.part_sums[(size_t)(c + u) * DK + i];
#pragma unroll
      for (int u = 0; u < 8; u++) a += t[u];
    }
    for (; c < chunks; c++) a += e.part_sums[(size_t)c * DK + i];
    reinterpret_cast<unsigned long long*>(e.sums)[i] = a;
  }
  if (i < K) {
    int a = 0, c = 0;
    for (; c + 8 <= chunks; c += 8) {
      int t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = e.part_counts[(size_t)(c + u) * K + i];
#pragma unroll
      for (int u = 0; u < 8; u++) a += t[u];
    }
    for (; c < chunks; c++) a += e.part_counts[(size_t)c * K + i];
    e.counts[i] = a;
  }
}

// cluster sizes of every quantizer (one workgroup each), then every quantizer's blend: two launches per forward.
// (Sizes and blend in ONE launch needs a "last workgroup writes the sizes" step; the device-scope fences that takes
// cost more on this multi-XCD part than the second launch.)
__global__ __launch_bounds__(1024) void vq_ema_size_multi_kernel(const EmaMP m) {
  __shared__ float red[1024];
  __shared__ float sz[4096];
  const EmaQ& e = m.q[blockIdx.x];
  const int tid = threadIdx.x, K = e.K;
  float part = 0.f;
  for (int k = tid; k < K; k += 1024) {
    const float v = ema_mix(m.decay, e.ema_size[k], m.omd, (float)e.counts[k]);
    sz[k] = v;
    part += v;
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float n = red[0];
  const float den = n + e.keps;
  for (int k = tid; k < K; k += 1024) e.ema_size[k] = (sz[k] + m.eps) / den * n;
}

__global__ __launch_bounds__(256) void vq_ema_blend_multi_kernel(const EmaMP m) {
  __shared__ float tile[16][17];
  const EmaQ& e = m.q[blockIdx.z];
  const int K = e.K, D = e.D;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int k0 = blockIdx.x * 16, d0 = blockIdx.y * 16;
  if (k0 >= K || d0 >= D) return;  // the grid spans the largest quantizer
  const int k = k0 + tx, d = d0 + ty;
  if (k < K && d < D) {
    const int i = d * K + k;
    const float es = (float)e.sums[i] * VQ_FIX_INV;
    const float w = ema_mix(m.decay, e.ema_w[i], m.omd, es);
    e.ema_w[i] = w;
    tile[ty][tx] = w / e.ema_size[k];
  }
  __syncthreads();
  const int kk = k0 + ty, dd = d0 + tx;
  if (kk < K && dd < D) e.cb[(size_t)kk * D + dd] = tile[tx][ty];
}

// The blend AND the codebook's image for the split-f16 search in one launch (D = 64, K <= 512): a workgroup owns a 32-code
// tile of one quantizer - it blends the tile's ema_w rows (the arithmetic of vq_ema_blend_multi_kernel, element for element),
// writes the tile's new code vectors and derives the tile's part of the image from them exactly as vq_image_kernel does from
// the codebook it would read back (vq_code_stats / vq_plane_piece: the same bits).  The generator's next forward finds its
// images current instead of spending a launch on them (two per training step).
__global__ __launch_bounds__(256) void vq_ema_blend_image_multi_kernel(const EmaMP m, const VqImgM im) {
  __shared__ float wimg[32 * VQH_WS];
  __shared__ float sw_s[32];
  const EmaQ& e = m.q[blockIdx.y];
  unsigned char* img = im.q[blockIdx.y].img;
  const int K = e.K, KT = ((K + 63) >> 6) * 2, ct = blockIdx.x, tid = threadIdx.x;
  if (ct >= KT) return;  // (the grid spans the largest codebook of the call)
  {
    const int kl = tid & 31, k = ct * 32 + kl;
    const float size = k < K ? e.ema_size[k] : 1.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int d = (tid >> 5) + 8 * j;
      float v = 0.f;
      if (k < K) {
        const int i = d * K + k;
        const float es = (float)e.sums[i] * VQ_FIX_INV;
        const float w = ema_mix(m.decay, e.ema_w[i], m.omd, es);
        e.ema_w[i] = w;
        v = w / size;
      }
      wimg[(size_t)kl * VQH_WS + d] = v;
    }
  }
  __syncthreads();
  vq_f32x4 pv[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int pi = tid + 256 * j, k = ct * 32 + (pi >> 4);
    pv[j] = *reinterpret_cast<const vq_f32x4*>(wimg + (size_t)(pi >> 4) * VQH_WS + 4 * (pi & 15));
    if (k < K) *reinterpret_cast<vq_f32x4*>(e.cb + (size_t)k * 64 + 4 * (pi & 15)) = pv[j];
  }
  float* tabs = reinterpret_cast<float*>(img + (size_t)KT * 8192);
  if (tid < 32) {
    const int k = ct * 32 + tid;
    float w2, us, sw, wm;
    vq_code_stats(wimg + (size_t)tid * VQH_WS, k, K, w2, us, sw, wm);
    tabs[k] = w2; tabs[KT * 32 + k] = us; tabs[2 * KT * 32 + k] = sw; tabs[3 * KT * 32 + k] = wm;
    sw_s[tid] = sw;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int pi = tid + 256 * j;
    vq_plane_piece(pv[j], sw_s[pi >> 4], ct * 32 + (pi >> 4), pi & 15, img, img + (size_t)KT * 4096);
  }
}

// per-chunk tables only (the first half of crk_vq_ema_stats); scratch: crk_vq_ema_scratch_bytes(N, D, K)
extern "C" int crk_vq_ema_partial(const float* x, int ldx, const long long* idx, int N, int D, int K, void* scratch,
                                  void* stream) {
  if (!x || !idx || !scratch || (D & 3) || (ldx & 3)) return CRK_ERR_ARG;
  int sw, chunks, fpc;
  if (vq_ema_plan(N, D, K, &sw, &chunks, &fpc) != CRK_OK) return CRK_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vq_ema_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            152 * 1024) != hipSuccess)
      return CRK_ERR_HIP;
    attr_set = true;
  }
  unsigned long long* part_sums = reinterpret_cast<unsigned long long*>(scratch);
  int* part_counts = reinterpret_cast<int*>(part_sums + (size_t)chunks * D * K);
  const size_t lds = (size_t)sw * K * 8 + (size_t)K * 4;
  hipLaunchKernelGGL(vq_ema_partial_kernel, dim3(chunks, (D + sw - 1) / sw), dim3(256), lds, (hipStream_t)stream, x, ldx, idx,
                     N, D, K, sw, fpc, part_sums, part_counts);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// crk_vq_ema_partial for up to 4 quantizer calls in one launch (the calls of one generator forward)
extern "C" int crk_vq_ema_partial_multi(int nq, const float* const* x, const int* ldx, const long long* const* idx, const int* N,
                                        const int* D, const int* K, void* const* scratch, void* stream) {
  if (nq < 1 || nq > 4 || !x || !ldx || !idx || !N || !D || !K || !scratch) return CRK_ERR_ARG;
  EmaPM m{};
  m.nq = nq;
  int gx = 0, gy = 0;
  size_t lds = 0;
  for (int q = 0; q < nq; q++) {
    if (!x[q] || !idx[q] || !scratch[q] || (D[q] & 3) || (ldx[q] & 3)) return CRK_ERR_ARG;
    int sw, chunks, fpc;
    if (vq_ema_plan(N[q], D[q], K[q], &sw, &chunks, &fpc) != CRK_OK) return CRK_ERR_UNSUPPORTED;
    EmaPQ& e = m.q[q];
    e.x = x[q]; e.idx = idx[q]; e.ldx = ldx[q]; e.N = N[q]; e.D = D[q]; e.K = K[q]; e.SW = sw; e.fpc = fpc; e.chunks = chunks;
    e.slices = (D[q] + sw - 1) / sw;
    e.part_sums = reinterpret_cast<unsigned long long*>(scratch[q]);
    e.part_counts = reinterpret_cast<int*>(e.part_sums + (size_t)chunks * D[q] * K[q]);
    if (chunks > gx) gx = chunks;
    if (e.slices > gy) gy = e.slices;
    const size_t need = (size_t)sw * K[q] * 8 + (size_t)K[q] * 4;
    if (need > lds) lds = need;
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vq_ema_partial_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
      return CRK_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL(vq_ema_partial_multi_kernel, dim3(gx, gy, nq), dim3(256), lds, (hipStream_t)stream, m);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// crk_vq_ema_reduce_multi and the cluster-size half of crk_vq_ema_apply_multi in ONE launch (single process: nothing is
// all-reduced between them): the last workgroup of a quantizer's row sums the per-chunk counts itself (integers: the same
// counts the other workgroups write) and runs vq_ema_size_multi_kernel's arithmetic on them, value for value.
__global__ __launch_bounds__(1024) void vq_ema_reduce_size_multi_kernel(const EmaMP m, int nblk) {
  __shared__ float red[1024];
  __shared__ float sz[4096];
  EmaQ e = m.q[0];
#pragma unroll
  for (int k = 1; k < VQ_EMA_MAXQ; k++)
    if ((int)blockIdx.y == k) e = m.q[k];
  const int DK = e.D * e.K, K = e.K, chunks = e.chunks, tid = threadIdx.x;
  if ((int)blockIdx.x < nblk) {
    const int i = blockIdx.x * 1024 + tid;
    if (i < DK) {
      unsigned long long a = 0ull;
      int c = 0;
      for (; c + 8 <= chunks; c += 8) {
        unsigned long long t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = e.part_sums[(size_t)(c + u) * DK + i];
#pragma unroll
        for (int u = 0; u < 8; u++) a += t[u];
      }
      for (; c < chunks; c++) a += e.part_sums[(size_t)c * DK + i];
      reinterpret_cast<unsigned long long*>(e.sums)[i] = a;
    }
    return;
  }
  float part = 0.f;
  for (int k = tid; k < K; k += 1024) {
    int a = 0, c = 0;
    for (; c + 8 <= chunks; c += 8) {
      int t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = e.part_counts[(size_t)(c + u) * K + k];
#pragma unroll
      for (int u = 0; u < 8; u++) a += t[u];
    }
    for (; c < chunks; c++) a += e.part_counts[(size_t)c * K + k];
    e.counts[k] = a;
    const float v = ema_mix(m.decay, e.ema_size[k], m.omd, (float)a);
    sz[k] = v;
    part += v;
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float n = red[0];
  const float den = n + e.keps;
  for (int k = tid; k < K; k += 1024) e.ema_size[k] = (sz[k] + m.eps) / den * n;
}
extern "C" int crk_vq_ema_reduce_size_multi(int nq, const void* const* scratch, const int* N, const int* D, const int* K,
                                            int* const* counts, long long* const* sums, float* const* ema_size, double decay,
                                            double eps, void* stream) {
  if (nq < 1 || nq > VQ_EMA_MAXQ || !scratch || !N || !D || !K || !counts || !sums || !ema_size) return CRK_ERR_ARG;
  EmaMP m{};
  m.nq = nq; m.decay = (float)decay; m.omd = (float)(1.0 - decay); m.eps = (float)eps;
  int maxdk = 0;
  for (int q = 0; q < nq; q++) {
    int sw, chunks, fpc;
    if (K[q] > 4096 || vq_ema_plan(N[q], D[q], K[q], &sw, &chunks, &fpc) != CRK_OK) return CRK_ERR_UNSUPPORTED;
    EmaQ& e = m.q[q];
    e.part_sums = reinterpret_cast<const unsigned long long*>(scratch[q]);
    e.part_counts = reinterpret_cast<const int*>(e.part_sums + (size_t)chunks * D[q] * K[q]);
    e.chunks = chunks; e.D = D[q]; e.K = K[q]; e.counts = counts[q]; e.sums = sums[q];
    e.ema_size = ema_size[q]; e.keps = (float)(K[q] * eps);
    if (D[q] * K[q] > maxdk) maxdk = D[q] * K[q];
  }
  const int nblk = (maxdk + 1023) / 1024;
  hipLaunchKernelGGL(vq_ema_reduce_size_multi_kernel, dim3(nblk + 1, nq), dim3(1024), 0, (hipStream_t)stream, m, nblk);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// scratch[q] as left by crk_vq_ema_partial(.., N[q], D[q], K[q], ..) -> counts[q] (K int32), sums[q] (D*K int64)
extern "C" int crk_vq_ema_reduce_multi(int nq, const void* const* scratch, const int* N, const int* D, const int* K,
                                       int* const* counts, long long* const* sums, void* stream) {
  if (nq < 1 || nq > VQ_EMA_MAXQ || !scratch || !N || !D || !K || !counts || !sums) return CRK_ERR_ARG;
  EmaMP m{};
  m.nq = nq;
  int maxdk = 0;
  for (int q = 0; q < nq; q++) {
    int sw, chunks, fpc;
    if (vq_ema_plan(N[q], D[q], K[q], &sw, &chunks, &fpc) != CRK_OK) return CRK_ERR_UNSUPPORTED;
    EmaQ& e = m.q[q];
    e.part_sums = reinterpret_cast<const unsigned long long*>(scratch[q]);
    e.part_counts = reinterpret_cast<const int*>(e.part_sums + (size_t)chunks * D[q] * K[q]);
    e.chunks = chunks; e.D = D[q]; e.K = K[q]; e.counts = counts[q]; e.sums = sums[q];
    if (D[q] * K[q] > maxdk) maxdk = D[q] * K[q];
  }
  hipLaunchKernelGGL(vq_ema_reduce_multi_kernel, dim3((maxdk + 255) / 256, nq), dim3(256), 0, (hipStream_t)stream, m);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// crk_vq_ema_apply for every quantizer: one size launch, one blend launch
extern "C" int crk_vq_ema_apply_multi(int nq, const int* const* counts, const long long* const* sums,
                                      float* const* ema_size, float* const* ema_w, float* const* codebook, const int* D,
                                      const int* K, double decay, double eps, void* stream) {
  if (nq < 1 || nq > VQ_EMA_MAXQ || !counts || !sums || !ema_size || !ema_w || !codebook || !D || !K) return CRK_ERR_ARG;
  EmaMP m{};
  m.nq = nq; m.decay = (float)decay; m.omd = (float)(1.0 - decay); m.eps = (float)eps;
  int maxk = 0, maxd = 0;
  for (int q = 0; q < nq; q++) {
    if (K[q] > 4096) return CRK_ERR_ARG;
    EmaQ& e = m.q[q];
    e.D = D[q]; e.K = K[q]; e.counts = const_cast<int*>(counts[q]); e.sums = const_cast<long long*>(sums[q]);
    e.ema_size = ema_size[q]; e.ema_w = ema_w[q]; e.cb = codebook[q]; e.keps = (float)(K[q] * eps);
    if (K[q] > maxk) maxk = K[q];
    if (D[q] > maxd) maxd = D[q];
  }
  hipLaunchKernelGGL(vq_ema_size_multi_kernel, dim3(nq), dim3(1024), 0, (hipStream_t)stream, m);
  hipLaunchKernelGGL(vq_ema_blend_multi_kernel, dim3((maxk + 15) / 16, (maxd + 15) / 16, nq), dim3(256), 0, (hipStream_t)stream, m);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// the blend half of crk_vq_ema_apply_multi (cluster sizes already updated: crk_vq_ema_reduce_size_multi)
extern "C" int crk_vq_ema_blend_multi(int nq, const long long* const* sums, float* const* ema_size, float* const* ema_w,
                                      float* const* codebook, const int* D, const int* K, double decay, void* stream) {
  if (nq < 1 || nq > VQ_EMA_MAXQ || !sums || !ema_size || !ema_w || !codebook || !D || !K) return CRK_ERR_ARG;
  EmaMP m{};
  m.nq = nq; m.decay = (float)decay; m.omd = (float)(1.0 - decay);
  int maxk = 0, maxd = 0;
  for (int q = 0; q < nq; q++) {
    EmaQ& e = m.q[q];
    e.D = D[q]; e.K = K[q]; e.sums = const_cast<long long*>(sums[q]);
    e.ema_size = ema_size[q]; e.ema_w = ema_w[q]; e.cb = codebook[q];
    if (K[q] > maxk) maxk = K[q];
    if (D[q] > maxd) maxd = D[q];
  }
  hipLaunchKernelGGL(vq_ema_blend_multi_kernel, dim3((maxk + 15) / 16, (maxd + 15) / 16, nq), dim3(256), 0, (hipStream_t)stream, m);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// crk_vq_ema_blend_multi that also leaves every codebook's image (crk_vq_image_bytes(K, 64) bytes each) current:
// D = 64 and K <= 512 for every quantizer of the call, CRK_ERR_UNSUPPORTED otherwise (the caller blends, then builds).
extern "C" int crk_vq_ema_blend_image_multi(int nq, const long long* const* sums, float* const* ema_size, float* const* ema_w,
                                            float* const* codebook, const int* D, const int* K, double decay, void* const* images,
                                            void* stream) {
  if (nq < 1 || nq > 4 || nq > VQ_EMA_MAXQ || !sums || !ema_size || !ema_w || !codebook || !D || !K || !images) return CRK_ERR_ARG;
  EmaMP m{};
  VqImgM im{};
  m.nq = nq; m.decay = (float)decay; m.omd = (float)(1.0 - decay);
  int ktmax = 0;
  for (int q = 0; q < nq; q++) {
    if (!sums[q] || !ema_size[q] || !ema_w[q] || !codebook[q] || !images[q] || K[q] <= 0) return CRK_ERR_ARG;
    if (D[q] != 64 || K[q] > 512) return CRK_ERR_UNSUPPORTED;
    EmaQ& e = m.q[q];
    e.D = D[q]; e.K = K[q]; e.sums = const_cast<long long*>(sums[q]);
    e.ema_size = ema_size[q]; e.ema_w = ema_w[q]; e.cb = codebook[q];
    im.q[q].cb = codebook[q]; im.q[q].img = (unsigned char*)images[q]; im.q[q].K = K[q];
    const int kt = ((K[q] + 63) / 64) * 2;
    if (kt > ktmax) ktmax = kt;
  }
  hipLaunchKernelGGL(vq_ema_blend_image_multi_kernel, dim3(ktmax, nq), dim3(256), 0, (hipStream_t)stream, m, im);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

extern "C" int crk_vq_ema_apply(const int* counts, const long long* sums, float* ema_size, float* ema_w,
                                float* codebook, int D, int K, double decay, double eps, void* stream) {
  if (!counts || !sums || !ema_size || !ema_w || !codebook || K > 4096) return CRK_ERR_ARG;
  // python-float semantics of vqvae2.py:316-328: scalars are rounded to fp32 when
  // they meet an fp32 tensor
  const float decay_f = (float)decay, omd_f = (float)(1.0 - decay), eps_f = (float)eps, keps_f = (float)(K * eps);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(vq_ema_size_kernel, dim3(1), dim3(1024), 0, s, counts, ema_size, K, decay_f, omd_f, eps_f, keps_f);
  hipLaunchKernelGGL(vq_ema_blend_kernel, dim3((K + 15) / 16, (D + 15) / 16), dim3(256), 0, s, sums, ema_size, ema_w,
                     codebook, D, K, decay_f, omd_f);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
