// Channel-split fused forward of ALL gated residual blocks of a stack (plain-bf16 arithmetic).
//
// Same semantics and the same HBM side as stack_fwd_kernel (stack_kernels.hip): parallel_wavegan
// ResidualBlock.forward chained as in ParallelWaveGANGenerator / ResidualParallelWaveGANDiscriminator
// (call sites crank/net/module/vqvae2.py:237-273, crank/bin/train.py:108-118),
// one workgroup per window of one utterance, the stack's receptive-field halo recomputed per window, the
// residual stream and the skip sum in fp32 registers across blocks, bf16 planes (block input, tanh, sigmoid,
// z) written for the backward pass.  What differs is WHO computes WHAT inside the workgroup:
//
//   stack_fwd_kernel : a wave owns 32 FRAMES and all 128 gate channels.  Every wave reads every weight
//                      fragment of a layer from LDS (1.25 KB of LDS reads per MFMA), the weights are
//                      staged through LDS one tap at a time behind a barrier per tap (7 barriers per
//                      block), and the window is 32 frames per wave, so 8 waves = 256 rows of which a
//                      balanced T = 500 utterance uses 215.
//   here             : a wave owns 32 CHANNELS (MFMA tile mt = wave & 3) of half the window (fh = wave >> 2),
//                      FT tiles of 32 frames each.  Its weights - one A fragment per (tap, k step) - go
//                      from L2 straight into registers in fragment order (weight_prep_kernel writes that
//                      layout), are used for FT MFMAs each and never touch LDS; only activations do
//                      (1 KB per MFMA).  No barrier inside the tap loop: two per block (gate output tile
//                      complete / next operand tile complete).  The window is 64*FT frames for FT = 2 or 3:
//                      T = 500 runs as 4 windows of 192 rows (173 used) on 256 workgroups instead of 3 x 256
//                      rows on 192 - a quarter fewer MFMAs per SIMD and no idle CUs.
//
// Gate pairing without a cross-lane step: gate tile mt holds tanh channels 16mt..16mt+15 in its rows 0-15
// and sigmoid channels 64+16mt.. in rows 16-31, so accumulator register j < 8 of a lane pairs with register
// j + 8 of the same lane.  The out|skip 1x1 runs as tiles 0,1 = residual channels 0-31 / 32-63 and tiles
// 2,3 = skip channels: waves 0,1 (+4,5) carry the residual stream, waves 2,3 (+6,7) the skip sum, as DATA
// (bias rows, scale, where the result goes) - the instruction stream is the same for every wave.
#include "conv_kernels.h"

#include "stack_common.h"

#define S2_NCU 256
#define S2_PAIR_FACTOR 1.0  // measured value goes here

// Phase-cycle instrumentation (tools/s2_phase_cycles.py builds a second library with -DS2_PROF): per workgroup and
// wave the shader cycles spent in [0] taps [1] gate [2] wait at barrier A [3] out|skip 1x1 + state update
// [4] next operand [5] wait at barrier B, summed over the blocks, [6] prologue (its barrier), [7] whole kernel,
// [8] prologue: first conv / state [9] tables, biases, guard rows [10] conditioning tile [11] block-0 operand
// [12] bias / table requests [13] weight, conditioning, input requests [14] input tile -> LDS [15] its barrier.
// Ablation builds (tools/s2_ablate.sh; timing only, results are wrong): S2_ABL bit 0 no transcendentals, bit 1 no
// MFMAs, bit 2 no LDS fragment reads, bit 3 no weight loads, bit 4 no barriers in the block loop, bit 5 no out|skip 1x1 phase,
// bit 6 no gate
#if defined(S2_ABL) && (S2_ABL & 2)
#define mfma_bf16(a, b, c) s2_fake_mfma(a, b, c)
__device__ __forceinline__ f32x16 s2_fake_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  asm volatile("" ::"v"(a), "v"(b));
  return c;
}
#endif
#if defined(S2_ABL) && (S2_ABL & 4)
#define lds_frag(p) s2_fake_frag(p)
__device__ __forceinline__ bf16x8 s2_fake_frag(const unsigned char* p) {
  const unsigned v = (unsigned)(size_t)p;
  const sk_u32x4 q = {v, v, v, v};
  return __builtin_bit_cast(bf16x8, q);
}
#endif
#ifdef S2_PROF
__device__ unsigned long long s2_prof_buf[256 * 8 * 16];
__device__ unsigned long long s2_prof_res[1024 * 4];  // per workgroup: start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID
extern "C" int crk_debug_s2_prof(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(s2_prof_buf), sizeof(unsigned long long) * 256 * 8 * 16) == hipSuccess ? 0 : 2;
}
extern "C" int crk_debug_s2_res(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(s2_prof_res), sizeof(unsigned long long) * 1024 * 4) == hipSuccess ? 0 : 2;
}
#define S2_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); pacc_[i] += now_ - plast_; plast_ = now_; }
#else
#define S2_T(i)
#endif

// The body of one wave: FT = the frame tiles THIS wave owns, R = the rows of the workgroup's window, rb = the first row of the
// wave's frame half.  RULE (uneven windows instantiate this body once per tile count and send the two frame halves into
// different copies, so their barriers sit in divergent code - legal on this hardware, where s_barrier only counts waves, as
// long as every copy issues the SAME barriers): no __syncthreads() may sit inside a loop or branch whose trip count or
// condition depends on FT.  Today: 2 in the prologue, 2 per block, 2 (FOLD) in the head - none of them FT-dependent.  (Windows whose two halves own different tile counts - 160 rows = 3 + 2 tiles for the k = 3 stacks -
// instantiate the body twice; the two copies hold the same barriers and the same workgroup-wide loops.)
template <int KT, int AKC, int FT, int R, int FH, bool DROP, bool FOLD>
__device__ __forceinline__ void s2_wave(const StackP& p, unsigned char* smem, const int rb) {
  constexpr int XS = SK_XS, NT = 256 * FH;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 3, fh = FH > 1 ? wave >> 2 : 0;
  const bool res_wave = mt < 2;  // carries the residual stream (else: the skip sum)
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;
  const long nbase = (long)b * p.T;
  const long P = (long)p.B * p.T * 64;
#ifdef S2_PROF
  unsigned long long pacc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, plast_ = __builtin_readcyclecounter();
  const unsigned long long pstart_ = plast_, preal_ = __builtin_amdgcn_s_memrealtime();
#endif

  unsigned char* xs = smem;             // [SK_GUARD + R + SK_GUARD][XS] block input as the conv sees it
  unsigned char* zs = smem + p.o_zs;    // [R][XS] gate output
  unsigned char* cs = smem + p.o_cs;    // [R][XS] conditioning (AKC > 0)
  StackLayer* lay_s = reinterpret_cast<StackLayer*>(smem + p.o_tab);
  float* bias_s = reinterpret_cast<float*>(smem + p.o_bias);  // [L][256]: conv 128 | out 64 | skip 64

  // ---- this lane's FT frames ----
  int row[FT], voff_st[FT], voff_b[FT];
  unsigned rmask[FT];
  bool rin[FT];
  const bool save_b = p.xb_hi != nullptr;
  const int ch_st = 32 * (mt & 1) + 4 * half;  // first channel of quad 0 of this lane's state tile (residual or skip plane)
#pragma unroll
  for (int ft = 0; ft < FT; ft++) {
    row[ft] = rb + ft * 32 + l31;
    const int t = t0 - p.hl + row[ft];
    rin[ft] = t >= 0 && t < p.T;
    rmask[ft] = rin[ft] ? 0xffffffffu : 0u;
    const bool rout = rin[ft] && row[ft] >= p.hl && row[ft] < p.hl + p.tmo;
    // fp32 [N,64] planes (block-0 input read by the residual waves, skip sum written by the skip waves)
    voff_st[ft] = (res_wave ? rin[ft] : rout) ? (int)(((nbase + t) * 64 + ch_st) * 4) : SK_OOB;
    // bf16 [N,64] planes: byte offset of channel 0 of this lane's frame
    voff_b[ft] = (rout && save_b) ? (int)(((nbase + t) * 64) * 2) : SK_OOB;
  }
  // lane-record layout of the tanh / sigmoid planes (StackP::ts_stride): the lane's 16-byte piece of frame n sits at
  // (n >> 5) * 4096 + (n & 31) * 16 + half * 512 + (mt >> 1) * 2048 + (mt & 1) * 1024 = voff_b + ts_delta: the frames of a
  // lane are 32 apart, n & 31 is the same for all of them (an out-of-range voff_b stays out of range)
  const int ts_delta = half * 512 + (mt >> 1) * 2048 + (mt & 1) * 1024 - 112 * (int)((nbase + t0 - p.hl + row[0]) & 31);

  // ---- prologue requests, every independent one in flight before anything waits (a dependent global round trip costs
  // 1.4 - 2.8 k cycles here and the prologue used to be a chain of ten: 31 k of the last decoder's 130 k cycles,
  // profiles/round6_b_fwd_prologue.txt).  In the order of what stands behind each: the stack input (LDS tile -> barrier ->
  // first conv -> block-0 operand -> barrier) first, the tap weights of block 0 last. ----
  const uint16_t* wl = p.whi + lane * 8;
#if defined(S2_ABL) && (S2_ABL & 8)
#define S2_WLOAD(off) (sk_u32x4{(unsigned)(off), (unsigned)(off) + 1u, (unsigned)lane, 0x3f803f80u})  // ablation: no weight loads
#else
#define S2_WLOAD(off) (*reinterpret_cast<const sk_u32x4*>(wl + (off)))
#endif
  // FOLD: the stack input window, fp32 rows walked by the whole workgroup in 16-byte pieces (coalesced: a row is up to 512
  // contiguous bytes).  (Round 5 gave each lane of the residual waves its own 32 bytes of a row per k step - the B fragment
  // straight from HBM: 32 cache lines per wave load for 32 bytes each, every line fetched by four k steps and two waves,
  // and a memory round trip per k step: 22 k cycles for the residual waves of the last decoder.)
  constexpr int XQ = FOLD ? (R * 32 + NT - 1) / NT : 1;  // pieces per thread at kp_first = 128
  sk_u32x4 xq[XQ];
  const int ppr = FOLD ? p.kp_first >> 2 : 1;  // 4-channel pieces per row
  const int xr0 = tid / ppr, xc0 = tid - xr0 * ppr, xdr = NT / ppr, xdc = NT - xdr * ppr;
  sk_f32x4 bfq[4];
  sk_u32x4 awf[FOLD ? 8 : 1];  // the first conv's A fragments of this wave's tile (kp_first <= 128: 8 k steps)
  if (FOLD) {
    const __amdgpu_buffer_rsrc_t rxi = sk_rsrc(p.x_in, (long)p.B * p.T * p.ldx_in);
    int xrow = xr0, xcol = xc0;
#pragma unroll
    for (int u = 0; u < XQ; u++) {
      const int c4 = xcol * 4, t = t0 - p.hl + xrow;
      const bool on = xrow < R && t >= 0 && t < p.T && c4 < p.in_ch;  // (in_ch is a multiple of 8: whole pieces)
      xq[u] = __builtin_amdgcn_raw_buffer_load_b128(rxi, on ? (int)(((nbase + t) * p.ldx_in + c4) * 4) : SK_OOB, 0, 0);
      xrow += xdr; xcol += xdc;
      if (xcol >= ppr) { xcol -= ppr; xrow++; }
    }
    if (res_wave) {
      const int KF = p.kp_first >> 4;
#pragma unroll
      for (int kc = 0; kc < 8; kc++) awf[kc] = S2_WLOAD(p.f_first + (mt * KF + (kc < KF ? kc : 0)) * 512);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
      bfq[q] = (res_wave && p.b_first >= 0) ? *reinterpret_cast<const sk_f32x4*>(p.params + p.b_first + 32 * mt + 8 * q + 4 * half) : sk_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // conditioning tile: 16 quads per row (64 channels, zero beyond aux_ch), packed into LDS behind the first conv
  constexpr int CNQ = R * 16, CPER = AKC > 0 ? (CNQ + NT - 1) / NT : 1;
  float av[CPER][4];
  if (AKC > 0) {
#pragma unroll
    for (int it = 0; it < CPER; it++) {
      const int idx = tid + it * NT, r = idx >> 4, c4 = (idx & 15) << 2;
      const int tt = t0 - p.hl + r;
      const bool on = idx < CNQ && tt >= 0 && tt < p.T;
      const long n = nbase + tt;
#pragma unroll
      for (int j = 0; j < 4; j++) av[it][j] = (on && c4 + j < p.aux_ch) ? p.c[n * p.ldc + c4 + j] : 0.f;
    }
  }
  S2_T(12)
  // biases: a wave's 64 consecutive entries of [L][256] belong to ONE block and ONE of conv | out | skip, so the table entry
  // comes through the scalar cache (no table-in-LDS -> barrier -> bias chain) and the values follow in one vector load each
  constexpr int NBI = 16 * 256 / NT;  // <= 16 blocks
  float bv[NBI];
#pragma unroll
  for (int k = 0; k < NBI; k++) {
    const int i0 = __builtin_amdgcn_readfirstlane(tid + k * NT) & ~63;  // first entry of this wave's 64
    const int l = i0 >> 8, c0 = i0 & 255, c = c0 + lane;
    bv[k] = 0.f;
    if (l < p.L) {
      const long long bo = c0 < 128 ? p.layers[l].b_conv : (c0 < 192 ? p.layers[l].b_out : p.layers[l].b_skip);
      // (the out conv's bias enters the residual update as fma(out + x, sqrt(.5), b * sqrt(.5)): stored pre-multiplied)
      if (bo >= 0) bv[k] = p.params[bo + (c0 < 128 ? c : (c0 < 192 ? c - 128 : c - 192))] * ((c0 >= 128 && c0 < 192) ? 0.70710678118654752440f : 1.f);
    }
  }
  int tabv[2] = {0, 0};  // this thread's words of the layer table (the block loop reads the table from LDS)
#pragma unroll
  for (int u = 0; u < 2; u++)
    if (tid + u * NT < p.L * (int)(sizeof(StackLayer) / 4)) tabv[u] = reinterpret_cast<const int*>(p.layers)[tid + u * NT];
  static_assert(16 * sizeof(StackLayer) / 4 <= 2 * NT, "two table words per thread");

  // ---- weights: A fragments straight from L2 (fragment order: 16 bytes per lane, 1 KB per wave-load) ----
  // register sets of tap weights: tap t lives in set t % NWB.  k = 3: all three taps; k = 5: four sets, the fifth tap
  // follows tap 0 into set 0.  NWB1 = taps requested ahead of a block (behind the previous block's gate).
  constexpr int NWB = KT == 3 ? 3 : 4, NWB1 = KT == 3 ? 3 : 2;
  sk_u32x4 wa[NWB][4];
  sk_u32x4 wos[4];
  sk_u32x4 wax[AKC > 0 ? AKC : 1];
  {
    const StackLayer L0 = p.layers[0];
#pragma unroll
    for (int kc = 0; kc < 4; kc++)
#pragma unroll
      for (int tp = 0; tp < NWB1; tp++) wa[tp][kc] = S2_WLOAD(L0.f_conv + ((tp * 4 + mt) * 4 + kc) * 512);
  }

  // ---- state: residual stream (block-0 input) or zero (skip sum) ----
  f32x16 st[FT];
  S2_T(13)
  if (FOLD) {
    // bf16 pieces -> the LDS tile [R][kp_first] the first conv reads its B fragments from, and (window's own frames) the plane
    // its weight gradient reads; then the first conv (1x1, in_ch -> 64): A = its weights (tile mt of the residual waves),
    // accumulator = the residual-stream layout.
    unsigned char* xf = smem + p.o_xf;
    const int xfs = p.kp_first * 2 + 16;  // row stride: + 16 B (conflict-free ds_read_b128)
    const __amdgpu_buffer_rsrc_t rfp = sk_rsrc16(p.fin_hi ? p.fin_hi : (const uint16_t*)p.x_in, (long)p.B * p.T * p.kp_first);
    {
      int xrow = xr0, xcol = xc0;
#pragma unroll
      for (int u = 0; u < XQ; u++) {
        const int c4 = xcol * 4, t = t0 - p.hl + xrow;
        if (xrow < R) {
          const sk_u32x2 h = {pack_bf2(sk_u2f(xq[u][0]), sk_u2f(xq[u][1])), pack_bf2(sk_u2f(xq[u][2]), sk_u2f(xq[u][3]))};
          *reinterpret_cast<sk_u32x2*>(xf + xrow * xfs + c4 * 2) = h;
          const bool ro = t >= 0 && t < p.T && xrow >= p.hl && xrow < p.hl + p.tmo && p.fin_hi != nullptr;
          __builtin_amdgcn_raw_buffer_store_b64(h, rfp, ro ? (int)(((nbase + t) * p.kp_first + c4) * 2) : SK_OOB, 0, 0);
        }
        xrow += xdr; xcol += xdc;
        if (xcol >= ppr) { xcol -= ppr; xrow++; }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int ft = 0; ft < FT; ft++)
#pragma unroll
        for (int j = 0; j < 4; j++) st[ft][4 * q + j] = bfq[q][j];
    S2_T(14)
    __syncthreads();  // the input tile is complete
    S2_T(15)
    if (res_wave) {
      const int KF = p.kp_first >> 4;
      const unsigned char* xb = xf + (rb + l31) * xfs + half * 16;
#pragma unroll
      for (int kc = 0; kc < 8; kc++) {
        if (kc < KF) {
#pragma unroll
          for (int ft = 0; ft < FT; ft++)
            st[ft] = mfma_bf16(__builtin_bit_cast(bf16x8, awf[kc]), lds_frag(xb + ft * 32 * xfs + kc * 32), st[ft]);
        }
      }
#pragma unroll
      for (int ft = 0; ft < FT; ft++)
#pragma unroll
        for (int i = 0; i < 16; i++) st[ft][i] = rin[ft] ? st[ft][i] : 0.f;
    }
  } else {
    const __amdgpu_buffer_rsrc_t rx0 = sk_rsrc(p.x0, P);
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        sk_u32x4 v = {0u, 0u, 0u, 0u};
        if (res_wave) v = __builtin_amdgcn_raw_buffer_load_b128(rx0, voff_st[ft] + q * 32, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) st[ft][4 * q + j] = sk_u2f(v[j]);
      }
  }

  S2_T(8)
  // ---- layer table, biases (requested at the top), guard rows, conditioning tile ----
#pragma unroll
  for (int u = 0; u < 2; u++)
    if (tid + u * NT < p.L * (int)(sizeof(StackLayer) / 4)) reinterpret_cast<int*>(lay_s)[tid + u * NT] = tabv[u];
#pragma unroll
  for (int k = 0; k < NBI; k++) {
    const int i = tid + k * NT;
    if (i < p.L * 256) bias_s[i] = bv[k];
  }
  for (int i = tid; i < SK_GUARD * XS / 16; i += NT) {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(xs)[i] = z4;
    reinterpret_cast<uint4*>(xs + (SK_GUARD + R) * XS)[i] = z4;
  }
  S2_T(9)
  if (AKC > 0) {
#pragma unroll
    for (int it = 0; it < CPER; it++) {
      const int idx = tid + it * NT, r = idx >> 4, c4 = (idx & 15) << 2;
      if (idx < CNQ) {
        const int tt = t0 - p.hl + r;
        sk_u32x2 hi, lo;
        sk_quad<false>(av[it][0], av[it][1], av[it][2], av[it][3], hi, lo);
        *reinterpret_cast<sk_u32x2*>(cs + r * XS + c4 * 2) = hi;
        if (p.cb_hi && c4 < p.aux_pad && tt >= 0 && tt < p.T && r >= p.hl && r < p.hl + p.tmo)
          *reinterpret_cast<sk_u32x2*>(p.cb_hi + (nbase + tt) * p.aux_pad + c4) = hi;
      }
    }
  }

  S2_T(10)
  const float rs = 0.70710678118654752440f;
  const float scale = res_wave ? rs : 1.f;
  // the second-dispatched half of the workgroup loses every arbitration for the SIMD it shares with an older wave
  // (taps: 3.3 k cycles for waves 0-3, 4.9 k for waves 4-7, the difference spent waiting at the barrier): static priority
  if (FH > 1 && fh) __builtin_amdgcn_s_setprio(1);

// the residual waves' state of frame tile ft as the next block's conv operand: 2 x 16 channels per frame -> two
// 16-byte pieces to the LDS tile and to the bf16 plane the weight gradient reads (dropout applied, as the conv sees it).
// The conv must see zeros outside the utterance (its zero padding): the mask is applied HERE, to the 8 packed words of a
// frame, not to the fp32 state (16 v_cndmask per tile and block whose SGPR mask two waves of a SIMD do not dual-issue) -
// the state of an out-of-utterance frame is never stored and only ever feeds its own frame.  Expects dseed and r_xh in scope.
#define S2_PUT_OPERAND_FT(ft)                                                                                   \
  {                                                                                                             \
    _Pragma("unroll") for (int g = 0; g < 2; g++) {                                                             \
      sk_u32x2 qh[2], ql[2];                                                                                    \
      _Pragma("unroll") for (int gg = 0; gg < 2; gg++) {                                                        \
        const int q = 2 * g + gg;                                                                               \
        float v[4];                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                         \
          v[j] = st[ft][4 * q + j];                                                                             \
          if (DROP && p.drop_p > 0.f && rin[ft])                                                                \
            v[j] *= dropout_scale(dseed, (unsigned long long)(nbase + t0 - p.hl + row[ft]) * 64 + 32 * mt + 8 * q + 4 * half + j, p.drop_p); \
        }                                                                                                       \
        sk_quad<false>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);                                                 \
      }                                                                                                         \
      sk_u32x4 fh_ = sk_frag_bits(sk_swap_frag(qh[0], qh[1]));                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; j++) fh_[j] &= rmask[ft]; /* zero outside the utterance */        \
      const int cb_ = (32 * mt + 16 * g + 8 * half) * 2;                                                        \
      *reinterpret_cast<sk_u32x4*>(xs + (SK_GUARD + row[ft]) * XS + cb_) = fh_;                                 \
      __builtin_amdgcn_raw_buffer_store_b128(fh_, r_xh, voff_b[ft] + cb_, 0, 0);                                \
    }                                                                                                           \
  }
  if (res_wave) {
    const unsigned long long dseed = (DROP ? crk_seed(p.drop_seed, p.drop_seed_ptr) : 0ull) + 0x9E3779B97F4A7C15ull * 1ull;
    const __amdgpu_buffer_rsrc_t r_xh = sk_rsrc16(save_b ? p.xb_hi : (const uint16_t*)p.skip, P);
#pragma unroll
    for (int ft = 0; ft < FT; ft++) S2_PUT_OPERAND_FT(ft)
  }
  S2_T(11)
  __syncthreads();  // tables, guard rows, conditioning tile, block-0 operand tile
  S2_T(6)

  f32x16 acc[FT];
  for (int l = 0; l < p.L; l++) {
    const StackLayer LY = lay_s[l];
    // ---- accumulators start from the conv bias of their rows ----
    {
      const float* bc = bias_s + l * 256 + 16 * mt + 4 * half;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const sk_f32x4 bq = *reinterpret_cast<const sk_f32x4*>(bc + (q < 2 ? 8 * q : 64 + 8 * (q - 2)));
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[ft][4 * q + j] = bq[j];
      }
    }
    // ---- dilated conv (+ conditioning 1x1).  Per output element the products are accumulated tap by tap, k step by
    // k step, conditioning last - in that order whatever the loop nest.  Two phases:
    //   1 (k = 5 only) taps 0, 1 TAP-major: one A fragment against the FT frame tiles, B fragments two steps ahead,
    //     pinned with scheduling fences (left alone the compiler sinks every load to just in front of its consumer);
    //   2 the last three taps and the conditioning steps FRAME-TILE-major: the M2 MFMAs of tile ft, then its gate.
    //     The gate of tile ft (transcendentals, packing, stores: VALU) has no dependence on the MFMAs of tile
    //     ft + 1, so the two overlap inside the wave - in phase-per-phase order the VALU work of a block (as long
    //     as its MFMA work) ran with the matrix pipe idle.
    const unsigned char* xb0 = xs + (SK_GUARD + rb + l31 + LY.off0) * XS + half * 16;
    const unsigned char* cb0 = cs + (rb + l31) * XS + half * 16;
    constexpr int T1 = KT > 3 ? KT - 3 : 0, NS1 = T1 * 4, M2 = (KT - T1) * 4 + AKC;
    if (NS1 > 0) {
      constexpr int NBQ = FT >= 4 ? 2 : 3;  // NBQ - 1 steps of B fragments in flight
      bf16x8 bq[NBQ][FT];
#define S2_BREAD(sx)                                                                                            \
  {                                                                                                             \
    const unsigned char* src_ = xb0 + ((sx) >> 2) * LY.dil * XS + ((sx) & 3) * 32;                              \
    _Pragma("unroll") for (int ft = 0; ft < FT; ft++) bq[(sx) % NBQ][ft] = lds_frag(src_ + ft * 32 * XS);       \
  }
      S2_BREAD(0)
      if (NBQ > 2) S2_BREAD(1)
#pragma unroll
      for (int sx = 0; sx < NS1; sx++) {
        const int tap = sx >> 2, kc = sx & 3;
        if (kc == 0) {  // tap t + NWB - 1 into the register set tap t - 1 has left
          if (tap == 0 && NWB - 1 < KT) {
#pragma unroll
            for (int tp = NWB - 2; tp <= NWB - 1; tp++)
              if (tp >= NWB1) {
#pragma unroll
                for (int k2 = 0; k2 < 4; k2++) wa[tp % NWB][k2] = S2_WLOAD(LY.f_conv + ((tp * 4 + mt) * 4 + k2) * 512);
              }
          } else if (tap > 0 && tap + NWB - 1 < KT) {
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) wa[(tap + NWB - 1) % NWB][k2] = S2_WLOAD(LY.f_conv + (((tap + NWB - 1) * 4 + mt) * 4 + k2) * 512);
          }
        }
        if (sx + NBQ - 1 < NS1) S2_BREAD(sx + NBQ - 1)
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 a = __builtin_bit_cast(bf16x8, wa[tap % NWB][kc]);
#pragma unroll
        for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(a, bq[sx % NBQ][ft], acc[ft]);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef S2_BREAD
    }
    // conditioning and out|skip fragments of this block (and, k = 3, the tap the prefetch left out)
    if (AKC > 0) {
#pragma unroll
      for (int k2 = 0; k2 < AKC; k2++) wax[k2] = S2_WLOAD(LY.f_aux + (mt * 4 + k2) * 512);
    }
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) wos[k2] = S2_WLOAD(LY.f_os + (mt * 4 + k2) * 512);
    {
      const __amdgpu_buffer_rsrc_t r_zh = sk_rsrc16(save_b ? p.zb_hi + (long)l * P : (const uint16_t*)p.skip, P);
      const bool ts_rec = p.ts_stride > 0;
      const long tsP = ts_rec ? (long)p.ts_stride : P;
      const __amdgpu_buffer_rsrc_t r_th = sk_rsrc16(save_b ? p.tb_hi + (long)l * tsP : (const uint16_t*)p.skip, tsP);
      const __amdgpu_buffer_rsrc_t r_gh = sk_rsrc16(save_b ? p.sg_hi + (long)l * tsP : (const uint16_t*)p.skip, tsP);
      const int cb = (16 * mt + 8 * half) * 2;  // this lane's 8-channel piece of a 64-channel row
      constexpr int NB2 = 4, NU = FT * M2;     // ring of single B fragments, NB2 - 1 MFMAs ahead
      bf16x8 b2[NB2];
#define S2_B2ADDR(u) ((((u) % M2) < (KT - T1) * 4 ? xb0 + (T1 + (((u) % M2) >> 2)) * LY.dil * XS + (((u) % M2) & 3) * 32 \
                                                   : cb0 + (((u) % M2) - (KT - T1) * 4) * 32) + ((u) / M2) * 32 * XS)
#pragma unroll
      for (int u = 0; u < NB2 - 1; u++) b2[u] = lds_frag(S2_B2ADDR(u));
// the M2 MFMAs of frame tile ft (B fragments NB2 - 1 MFMAs ahead)
#define S2_CHAIN(ft)                                                                                            \
  _Pragma("unroll") for (int m = 0; m < M2; m++) {                                                              \
    const int u = (ft) * M2 + m;                                                                                \
    if (u + NB2 - 1 < NU) b2[(u + NB2 - 1) % NB2] = lds_frag(S2_B2ADDR(u + NB2 - 1));                           \
    const bf16x8 a = __builtin_bit_cast(bf16x8, m < (KT - T1) * 4 ? wa[(T1 + (m >> 2)) % NWB][m & 3] : wax[m < (KT - T1) * 4 ? 0 : m - (KT - T1) * 4]); \
    acc[ft] = mfma_bf16(a, b2[u % NB2], acc[ft]);                                                               \
  }
// gate of frame tile ft: register j < 8 (tanh row) pairs with register j + 8 (sigmoid row)
#define S2_GATE(ft)                                                                                             \
  {                                                                                                             \
    sk_u32x2 zq[2], tq[2], sq[2], dummy;                                                                        \
    _Pragma("unroll") for (int gg = 0; gg < 2; gg++) {                                                          \
      float ta[4], sb[4];                                                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                           \
        ta[j] = sk_tanh(acc[ft][4 * gg + j], false);                                                            \
        sb[j] = sk_sigmoid(acc[ft][8 + 4 * gg + j], false);                                                     \
      }                                                                                                         \
      sk_quad<false>(ta[0], ta[1], ta[2], ta[3], tq[gg], dummy);                                                \
      sk_quad<false>(sb[0], sb[1], sb[2], sb[3], sq[gg], dummy);                                                \
      sk_quad<false>(ta[0] * sb[0], ta[1] * sb[1], ta[2] * sb[2], ta[3] * sb[3], zq[gg], dummy);                \
    }                                                                                                           \
    const sk_u32x4 zf = sk_frag_bits(sk_swap_frag(zq[0], zq[1]));                                               \
    *reinterpret_cast<sk_u32x4*>(zs + row[ft] * XS + cb) = zf;                                                  \
    if (ts_rec) { /* the lane's own two quads, as they are: 1 KB runs per (32 frames, tile) */                    \
      const sk_u32x4 tpc_ = {tq[0][0], tq[0][1], tq[1][0], tq[1][1]}, spc_ = {sq[0][0], sq[0][1], sq[1][0], sq[1][1]}; \
      __builtin_amdgcn_raw_buffer_store_b128(tpc_, r_th, voff_b[ft] + ts_delta, 0, 0);                            \
      __builtin_amdgcn_raw_buffer_store_b128(spc_, r_gh, voff_b[ft] + ts_delta, 0, 0);                            \
    } else {                                                                                                      \
      __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(tq[0], tq[1])), r_th, voff_b[ft] + cb, 0, 0); \
      __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(sq[0], sq[1])), r_gh, voff_b[ft] + cb, 0, 0); \
    }                                                                                                             \
    __builtin_amdgcn_raw_buffer_store_b128(zf, r_zh, voff_b[ft] + cb, 0, 0);                                    \
  }
      // software pipeline over the frame tiles: the MFMAs of tile ft are issued between the gate instructions of tile
      // ft - 1 (per MFMA: its B fragment read and its share of the gate's ~92 VALU / transcendental instructions; the
      // compiler on its own emits the MFMA chain, then the gate, and the matrix pipe idles through every gate)
      S2_CHAIN(0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ft = 1; ft < FT; ft++) {
        S2_CHAIN(ft)
#if !(defined(S2_ABL) && (S2_ABL & 64))
        S2_GATE(ft - 1)
#endif
#pragma unroll
        for (int m = 0; m < M2; m++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
          __builtin_amdgcn_sched_group_barrier(0x402, (92 + M2 - 1) / M2, 0);  // VALU / transcendental: the gate's ~92, spread over the chain
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#if !(defined(S2_ABL) && (S2_ABL & 64))
      S2_GATE(FT - 1)
#else
      _Pragma("unroll") for (int ft = 0; ft < FT; ft++) asm volatile("" ::"v"(acc[ft]));  // (keeps the MFMAs alive)
#endif
#undef S2_CHAIN
#undef S2_GATE
#undef S2_B2ADDR
    }
    S2_T(0)
    // the next block's first taps: in flight behind the barrier and the 1x1 (every register set is free now)
    if (l + 1 < p.L) {
      const StackLayer LN = lay_s[l + 1];
#pragma unroll
      for (int tp = 0; tp < NWB1; tp++)
#pragma unroll
        for (int kc = 0; kc < 4; kc++) wa[tp][kc] = S2_WLOAD(LN.f_conv + ((tp * 4 + mt) * 4 + kc) * 512);
    }
    S2_T(1)
#if !(defined(S2_ABL) && (S2_ABL & 16))
    __syncthreads();  // gate-output tile complete; every tap read of the operand tile done
#endif
    S2_T(2)
    // ---- out | skip 1x1 on z, frame tile by frame tile: tile mt of [out 0-31 | out 32-63 | skip 0-31 | skip 32-63];
    // the state update and the next operand of tile ft overlap the MFMAs of tile ft + 1 ----
#if defined(S2_ABL) && (S2_ABL & 32)
    if (p.L > 100)
#endif
    {
      // The MFMA chain accumulates ON the state (C operand = st): residual waves x <- fma(x + out, sqrt(.5), b sqrt(.5)),
      // skip waves s <- fma(s + skip, 1, b) - one instruction per element for both kinds, no accumulator initialisation,
      // no masking (S2_PUT_OPERAND_FT masks the packed operand).  stack_fwd_kernel evaluates the same expression.
      const float* bo = bias_s + l * 256 + 128 + 32 * mt + 4 * half;
      sk_f32x4 bsc[4];
#pragma unroll
      for (int q = 0; q < 4; q++) bsc[q] = *reinterpret_cast<const sk_f32x4*>(bo + 8 * q);
      const unsigned char* zb0 = zs + (rb + l31) * XS + half * 16;
      const unsigned long long dseed = (DROP ? crk_seed(p.drop_seed, p.drop_seed_ptr) : 0ull) + 0x9E3779B97F4A7C15ull * (unsigned long long)(l + 2);
      const __amdgpu_buffer_rsrc_t r_xh = sk_rsrc16(save_b ? p.xb_hi + (long)(l + 1) * P : (const uint16_t*)p.skip, P);
      bf16x8 zq[FT][4];
#pragma unroll
      for (int ft = 0; ft < FT; ft++)
#pragma unroll
        for (int kc = 0; kc < 4; kc++) zq[ft][kc] = lds_frag(zb0 + ft * 32 * XS + kc * 32);
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
#pragma unroll
        for (int kc = 0; kc < 4; kc++) st[ft] = mfma_bf16(__builtin_bit_cast(bf16x8, wos[kc]), zq[ft][kc], st[ft]);
#pragma unroll
        for (int i = 0; i < 16; i++) st[ft][i] = __builtin_fmaf(st[ft][i], scale, bsc[i >> 2][i & 3]);
        if (res_wave && l + 1 < p.L) S2_PUT_OPERAND_FT(ft)
      }
    }
    S2_T(3)
    S2_T(4)
#if !(defined(S2_ABL) && (S2_ABL & 16))
    __syncthreads();  // next operand tile complete; every read of the gate-output tile done
#endif
    S2_T(5)
  }

  if (FOLD) {
    // ---- the stack's head right here: relu(skip * sqrt(1/L)) -> 1x1 (64 -> 64) -> relu -> 1x1 (64 -> out_ch).  Both
    // operands pass through the two LDS tiles (and are the planes their weight gradients and the head's data
    // gradient read); M = 64 = two tiles for the first conv (residual waves), ceil(out_ch / 32) tiles for the second.
    const __amdgpu_buffer_rsrc_t r_s = sk_rsrc16(p.head_hi ? p.head_hi : (const uint16_t*)p.x_in, P);
    const __amdgpu_buffer_rsrc_t r_h = sk_rsrc16(p.head_hi ? p.head_hi + P : (const uint16_t*)p.x_in, P);
    // both convs' weight tiles (register sets 0 and 1 of the tap weights are free) and bias quads requested here: nothing of the head waits for
    // a load it has only just asked for
    sk_f32x4 bh1[4], bh2[4];
    if (res_wave) {
#pragma unroll
      for (int kc = 0; kc < 4; kc++) wa[0][kc] = S2_WLOAD(p.f_h1 + (mt * 4 + kc) * 512);
    }
    if (32 * mt < p.out_ch) {
#pragma unroll
      for (int kc = 0; kc < 4; kc++) wa[1][kc] = S2_WLOAD(p.f_h2 + (mt * 4 + kc) * 512);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ch = 32 * mt + 8 * q + 4 * half;
      bh1[q] = (res_wave && p.b_h1 >= 0) ? *reinterpret_cast<const sk_f32x4*>(p.params + p.b_h1 + ch) : sk_f32x4{0.f, 0.f, 0.f, 0.f};
      bh2[q] = (p.b_h2 >= 0 && ch < p.out_ch) ? *reinterpret_cast<const sk_f32x4*>(p.params + p.b_h2 + ch) : sk_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!res_wave) {
#pragma unroll
      for (int ft = 0; ft < FT; ft++)
#pragma unroll
        for (int g = 0; g < 2; g++) {
          sk_u32x2 qh[2], ql[2];
#pragma unroll
          for (int gg = 0; gg < 2; gg++) {
            const int q = 2 * g + gg;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = fmaxf(st[ft][4 * q + j] * p.head_scale, 0.f);
            sk_quad<false>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);
          }
          const sk_u32x4 fs = sk_frag_bits(sk_swap_frag(qh[0], qh[1]));
          const int cb_ = (32 * (mt - 2) + 16 * g + 8 * half) * 2;
          *reinterpret_cast<sk_u32x4*>(zs + row[ft] * XS + cb_) = fs;
          __builtin_amdgcn_raw_buffer_store_b128(fs, r_s, voff_b[ft] + cb_, 0, 0);
        }
    }
    __syncthreads();
    if (res_wave) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const sk_f32x4 bq = bh1[q];
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[ft][4 * q + j] = bq[j];
      }
      const unsigned char* zb0 = zs + (rb + l31) * XS + half * 16;
#pragma unroll
      for (int kc = 0; kc < 4; kc++)
#pragma unroll
        for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(__builtin_bit_cast(bf16x8, wa[0][kc]), lds_frag(zb0 + ft * 32 * XS + kc * 32), acc[ft]);
#pragma unroll
      for (int ft = 0; ft < FT; ft++)
#pragma unroll
        for (int g = 0; g < 2; g++) {
          sk_u32x2 qh[2], ql[2];
#pragma unroll
          for (int gg = 0; gg < 2; gg++) {
            const int q = 2 * g + gg;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = rin[ft] ? fmaxf(acc[ft][4 * q + j], 0.f) : 0.f;
            sk_quad<false>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);
          }
          const sk_u32x4 fh1 = sk_frag_bits(sk_swap_frag(qh[0], qh[1]));
          const int cb_ = (32 * mt + 16 * g + 8 * half) * 2;
          *reinterpret_cast<sk_u32x4*>(xs + (SK_GUARD + row[ft]) * XS + cb_) = fh1;
          __builtin_amdgcn_raw_buffer_store_b128(fh1, r_h, voff_b[ft] + cb_, 0, 0);
        }
    }
    __syncthreads();
    if (32 * mt < p.out_ch) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const sk_f32x4 bq = bh2[q];
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[ft][4 * q + j] = bq[j];
      }
      const unsigned char* hb0 = xs + (SK_GUARD + rb + l31) * XS + half * 16;
#pragma unroll
      for (int kc = 0; kc < 4; kc++)
#pragma unroll
        for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(__builtin_bit_cast(bf16x8, wa[1][kc]), lds_frag(hb0 + ft * 32 * XS + kc * 32), acc[ft]);
      const __amdgpu_buffer_rsrc_t ry = sk_rsrc(p.y, (long)p.B * p.T * p.ldy);
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        const long nn = nbase + t0 - p.hl + row[ft];
        const bool ro = rin[ft] && row[ft] >= p.hl && row[ft] < p.hl + p.tmo;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int ch = 32 * mt + 8 * q + 4 * half;
          sk_u32x4 v;
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = sk_f2u(acc[ft][4 * q + j]);
          __builtin_amdgcn_raw_buffer_store_b128(v, ry, (ro && ch < p.out_ch) ? (int)((nn * p.ldy + ch) * 4) : SK_OOB, 0, 0);
        }
      }
    }
  } else if (!res_wave) {
    // ---- running skip sum of the window's own frames (skip waves) ----
    const __amdgpu_buffer_rsrc_t r_sk = sk_rsrc(p.skip, P);
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        sk_u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = sk_f2u(st[ft][4 * q + j]);
        __builtin_amdgcn_raw_buffer_store_b128(v, r_sk, voff_st[ft] + q * 32, 0, 0);
      }
  }
#ifdef S2_PROF
  pacc_[7] = __builtin_readcyclecounter() - pstart_;
  if (blockIdx.x < 256 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 16; i++) s2_prof_buf[(blockIdx.x * 8 + wave) * 16 + i] = pacc_[i];
  }
  if (blockIdx.x < 1024 && tid == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    s2_prof_res[blockIdx.x * 4 + 0] = preal_; s2_prof_res[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    s2_prof_res[blockIdx.x * 4 + 2] = hwid; s2_prof_res[blockIdx.x * 4 + 3] = xcc;
  }
#endif
}

template <int KT, int AKC, int FT, int FH, bool DROP, bool FOLD, int FT1 = FT>
__global__ __launch_bounds__(256 * FH, 2) void stack2_fwd_kernel(const StackP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int R = FH > 1 ? 32 * (FT + FT1) : 32 * FT;
  const int fh = FH > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
  if constexpr (FT1 == FT) {
    s2_wave<KT, AKC, FT, R, FH, DROP, FOLD>(p, smem, fh * 32 * FT);
  } else {  // frame half 0 owns FT tiles, frame half 1 owns FT1: the two waves of a SIMD are one of each
    if (fh == 0) s2_wave<KT, AKC, FT, R, FH, DROP, FOLD>(p, smem, 0);
    else s2_wave<KT, AKC, FT1, R, FH, DROP, FOLD>(p, smem, 32 * FT);
  }
}

// Window shapes: (ft, fh) = (2, 2) 128 rows / 8 waves, (3, 2) 192 rows / 8 waves, and (round 4) 160 rows = 3 + 2 tiles for
// the k = 3 stacks (T = 500: 4 windows of 160 rows on 256 workgroups where 192-row windows made 3 on 192 - a quarter of the
// CUs idle - and 128-row ones 5 on 320, two rounds).  (Measured and dropped: (4, 1), 128
// rows on 4 waves with two independent workgroups per CU - both do co-reside, but a CU then finishes 256 rows in
// 66 us against 192 rows in 52 us here, and the smaller windows recompute 16 % more halo: slower in total.  Round 5 repeated
// it with the CU's second workgroup started 0 - 12 k cycles late (the two then sit in different phases of a block instead of
// both in their tap phase): k = 3 stacks 45.2 - 46.9 us whatever the skew against 35.5 us for the 160-row shape
// (profiles/round5_fwd_two_workgroups_skew.txt) - a wave's time is its own dependency chain, not contention with its
// SIMD neighbour.)
// Cost model: MFMA rounds per SIMD = ceil(workgroups / CUs) x tiles per wave; CRK_S2_CFG=<ft><fh> overrides.
int stack2_fwd_plan(StackP& p) {
  if ((p.ktaps != 3 && p.ktaps != 5) || p.max_off > SK_GUARD || p.aux_ch > 64 || p.L > 16) return CRK_ERR_UNSUPPORTED;
  if (p.ktaps == 3 && p.aux_ch > 0) return CRK_ERR_UNSUPPORTED;  // (no caller; keeps the instantiation count down)
  static int cfg_env = -1;
  if (cfg_env < 0) cfg_env = crk_sw().s2_cfg;
  // (ft, fh, ft1): tiles per wave of frame half 0, frame halves, tiles per wave of frame half 1.  CRK_S2_CFG = <ft><fh> or
  // <ft><fh><ft1> pins a shape.  The uneven 160-row shape exists for k = 3 without dropout (the stacks it pays for).
  static const int shapes[3][3] = {{2, 2, 2}, {3, 2, 2}, {3, 2, 3}};
  int best = -1; double best_cost = 0;
  for (int i = 0; i < 3; i++) {
    const int ft = shapes[i][0], fh = shapes[i][1], ft1 = shapes[i][2];
    const int code = ft1 == ft ? ft * 10 + fh : (ft * 10 + fh) * 10 + ft1;
    if (cfg_env && cfg_env != code) continue;
    if (ft1 != ft && (p.ktaps != 3 || p.drop_p > 0.f)) continue;
    const int rows = 32 * (ft + ft1) * fh / 2;
    const int tmo = rows - p.hl - p.hr;
    if (tmo < 16) continue;
    const long wgs = (long)p.B * ceil_div(p.T, tmo);
    const long slots = S2_NCU * (fh == 1 ? 2 : 1);
    // MFMA rounds per SIMD: the two waves of a SIMD are one of each frame half -> (ft + ft1) / 2 tiles per wave on average
    const double cost = (double)((wgs + slots - 1) / slots) * 0.5 * (ft + ft1) * (fh == 1 ? S2_PAIR_FACTOR : 1.0);
    if (best < 0 || cost <= best_cost) { best = i; best_cost = cost; }
  }
  if (best < 0) return CRK_ERR_UNSUPPORTED;
  p.ft = shapes[best][0]; p.fh = shapes[best][1]; p.ft1 = shapes[best][2] == shapes[best][0] ? 0 : shapes[best][2];
  const int R = 32 * (p.ft + (p.ft1 ? p.ft1 : p.ft)) * p.fh / 2;
  p.tmo = R - p.hl - p.hr;
  p.tiles_per_utt = ceil_div(p.T, p.tmo);
  p.tmo = ceil_div(p.T, p.tiles_per_utt);
  int off = (SK_GUARD * 2 + R) * SK_XS;
  p.o_zs = off; off += R * SK_XS;
  p.o_cs = off; if (p.aux_ch > 0) off += R * SK_XS;
  p.o_bias = off; off += p.L * 256 * 4;
  p.o_tab = off; off += p.L * (int)sizeof(StackLayer);
  off = (off + 15) & ~15;
  if (p.x_in) {  // folded first conv: its input tile [R][kp_first] bf16
    if (p.kp_first < 16 || p.kp_first > 128 || (p.kp_first & 15)) return CRK_ERR_UNSUPPORTED;
    p.o_xf = off; off += R * (p.kp_first * 2 + 16);
  }
  p.lds_bytes = (off + 15) & ~15;
  return p.lds_bytes <= (p.fh == 1 ? 80 : 160) * 1024 ? CRK_OK : CRK_ERR_UNSUPPORTED;
}

template <int KT, int AKC, bool DROP, bool FOLD>
static int s2_launch_shape(const StackP& p, dim3 grid, hipStream_t s) {
#define S2_GO(FTV, FHV)                                                                                              \
  {                                                                                                                  \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      if (hipFuncSetAttribute((const void*)stack2_fwd_kernel<KT, AKC, FTV, FHV, DROP, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              160 * 1024) != hipSuccess) return CRK_ERR_HIP;                                         \
      attr = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((stack2_fwd_kernel<KT, AKC, FTV, FHV, DROP, FOLD>), grid, dim3(256 * FHV), p.lds_bytes, s, p);       \
  }
  if (p.ft == 2) S2_GO(2, 2) else if (p.ft1 == 0) S2_GO(3, 2)
  else if constexpr (KT == 3 && !DROP) {  // 160 rows: frame half 0 owns three tiles, frame half 1 two
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)stack2_fwd_kernel<KT, AKC, 3, 2, DROP, FOLD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024) != hipSuccess) return CRK_ERR_HIP;
      attr = true;
    }
    hipLaunchKernelGGL((stack2_fwd_kernel<KT, AKC, 3, 2, DROP, FOLD, 2>), grid, dim3(512), p.lds_bytes, s, p);
  } else return CRK_ERR_UNSUPPORTED;
#undef S2_GO
  return CRK_OK;
}

int launch_stack2_fwd(const StackP& p, hipStream_t s) {
  dim3 grid(p.B * p.tiles_per_utt);
  const double nfr = (double)p.B * p.T;
  conv_prof_bytes(1, nfr * (256.0 + 4.0 * p.aux_ch + 256.0 + (p.xb_hi ? 512.0 * p.L + 2.0 * (p.aux_ch > 0 ? p.aux_pad : 0) : 0.0)));
  conv_prof_begin(1, 2.0 * nfr * (p.L * (128.0 * (64.0 * p.ktaps + p.aux_ch) + 128.0 * 64.0) +
                                   (p.x_in ? 64.0 * p.in_ch + 64.0 * 64.0 + 64.0 * p.out_ch : 0.0)), s);
  const int akc = p.aux_ch > 0 ? (p.aux_ch + 15) / 16 : 0;
  int rc = CRK_OK;
  const bool fold = p.x_in != nullptr;
#define S2_DISPATCH(KTV, AKCV) (fold ? s2_launch_shape<KTV, AKCV, false, true>(p, grid, s) : s2_launch_shape<KTV, AKCV, false, false>(p, grid, s))
  if (p.ktaps == 3) rc = S2_DISPATCH(3, 0);
  else if (p.drop_p > 0.f) {
    if (akc || fold) return CRK_ERR_UNSUPPORTED;
    rc = s2_launch_shape<5, 0, true, false>(p, grid, s);
  } else if (akc == 0) rc = S2_DISPATCH(5, 0);
  else if (akc == 1) rc = S2_DISPATCH(5, 1);
  else if (akc <= 3) rc = S2_DISPATCH(5, 3);
  else rc = S2_DISPATCH(5, 4);
#undef S2_DISPATCH
  conv_prof_end(1, s);
  if (rc != CRK_OK) return rc;
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
