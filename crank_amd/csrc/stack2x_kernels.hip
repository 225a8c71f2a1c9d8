// Channel-split fused forward of a generator stack in SPLIT-OPERAND arithmetic ("bf16x3": every product as three bf16 MFMAs,
// w_hi x_hi + w_hi x_lo + w_lo x_hi, ~fp32 accuracy) - the forward of the `bf16x3f` mode, whose losses meet the 1e-3 bar
// against the fp32 reference (crank/net/module/vqvae2.py:237-273 through parallel_wavegan's ResidualBlock chain; SURVEY.md
// Appendix A.1-A.3) while its backward runs in plain bf16.
//
// Same decomposition as stack2_fwd_kernel (stack2_kernels.hip): first 1x1 conv in the prologue, all gated residual blocks,
// ReLU -> 1x1 -> ReLU -> 1x1 head in the epilogue, one launch per stack; a wave owns one 32-channel MFMA tile (tanh rows
// 0-15 | sigmoid rows 16-31) of half the window's frame tiles; weights go from L2 to registers in fragment order, only
// activations pass through LDS.  What differs:
//   * every operand exists twice - hi = bf16(v), lo = bf16(v - hi): weight fragments from the hi AND the lo plane
//     (weight_prep writes both in fragment order), activation tiles xs / zs as hi and lo LDS tiles;
//   * the conditioning tile holds one DWORD per channel (hi | lo << 16) at a row stride of exactly the padded channel count:
//     the same 27 KB as one bf16 tile, so the dec0 window (192 rows, 34 conditioning channels) still fits 160 KB of LDS.
//     A fragment read runs past the row's end into the next row (finite values) - against weight columns that are zero;
//   * the weight stream is a ring of S2X_RING (hi, lo) fragment pairs requested S2X_RING k-steps ahead (a k-step is
//     3 FT MFMAs here, 306 cycles per wave: the L2 latency hides behind one or two of them);
//   * the saved planes are the hi planes in exactly the layout stack2_fwd_kernel writes (lane records for tanh / sigmoid),
//     so the plain-bf16 data-gradient chain (stack2b_kernels.hip) and weight-gradient kernel consume them unchanged;
//   * gate transcendentals on the hardware exp2 / rcp path (1 ulp-class: four orders below the mode's 1e-3 bar).
// Summation order per output element: tap by tap, k step by k step, conditioning last, and within a k step hi.hi, hi.lo,
// lo.hi - the order of stack_fwd_kernel<PRECISE = true> (stack_kernels.hip), to which the results agree to fp32 rounding of
// the gate (that kernel uses libm expf and an IEEE division; tests/test_gpu_nets.py).
#include "conv_kernels.h"

#include "stack_common.h"

#define S2X_RING 4

template <int KT, int AKC, int FT, int R>
__device__ __forceinline__ void s2x_wave(const StackP& p, unsigned char* smem, const int rb) {
  constexpr int XS = SK_XS, NT = 512;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 3, fh = wave >> 2;
  const bool res_wave = mt < 2;  // carries the residual stream (else: the skip sum)
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;
  const long nbase = (long)b * p.T;
  const long P = (long)p.B * p.T * 64;

  unsigned char* xs = smem;              // [SK_GUARD + R + SK_GUARD][XS] block input as the conv sees it, hi
  unsigned char* xl = smem + p.o_xlo;    // ... lo
  unsigned char* zs = smem + p.o_zs;     // [R][XS] gate output, hi
  unsigned char* zl = smem + p.o_zlo;    // ... lo
  unsigned char* cs = smem + p.o_cs;     // [R][cs_stride] (+ 256 zero bytes) conditioning: one dword (hi | lo << 16) per channel (AKC > 0)
  StackLayer* lay_s = reinterpret_cast<StackLayer*>(smem + p.o_tab);
  float* bias_s = reinterpret_cast<float*>(smem + p.o_bias);  // [L][256]: conv 128 | out 64 | skip 64

  int row[FT], voff_b[FT];
  unsigned rmask[FT];
  bool rin[FT];
  const bool save_b = p.xb_hi != nullptr;
#pragma unroll
  for (int ft = 0; ft < FT; ft++) {
    row[ft] = rb + ft * 32 + l31;
    const int t = t0 - p.hl + row[ft];
    rin[ft] = t >= 0 && t < p.T;
    rmask[ft] = rin[ft] ? 0xffffffffu : 0u;
    const bool rout = rin[ft] && row[ft] >= p.hl && row[ft] < p.hl + p.tmo;
    voff_b[ft] = (rout && save_b) ? (int)(((nbase + t) * 64) * 2) : SK_OOB;  // bf16 [N,64] planes: channel 0 of the lane's frame
  }
  // lane-record layout of the tanh / sigmoid planes (StackP::ts_stride), as in stack2_fwd_kernel
  const int ts_delta = half * 512 + (mt >> 1) * 2048 + (mt & 1) * 1024 - 112 * (int)((nbase + t0 - p.hl + row[0]) & 31);

  const uint16_t* wl_h = p.whi + lane * 8;
  const uint16_t* wl_l = p.wlo + lane * 8;
#define S2X_WH(off) (*reinterpret_cast<const sk_u32x4*>(wl_h + (off)))
#define S2X_WL(off) (*reinterpret_cast<const sk_u32x4*>(wl_l + (off)))
// three MFMAs of one k step: hi.hi, hi.lo, lo.hi
#define S2X_MMA(acc_, ah_, al_, bh_, bl_)                                                     \
  {                                                                                           \
    acc_ = mfma_bf16(__builtin_bit_cast(bf16x8, ah_), bh_, acc_);                             \
    acc_ = mfma_bf16(__builtin_bit_cast(bf16x8, ah_), bl_, acc_);                             \
    acc_ = mfma_bf16(__builtin_bit_cast(bf16x8, al_), bh_, acc_);                             \
  }
  // k step s of a block: taps first (4 k steps each), then the conditioning k steps
  constexpr int NS = KT * 4 + AKC;
#define S2X_AOFF(LY_, s) ((s) < KT * 4 ? (LY_).f_conv + (((((s) >> 2) * 4 + mt) * 4 + ((s) & 3)) * 512) \
                                       : (LY_).f_aux + ((mt * 4 + ((s) - KT * 4)) * 512))
  sk_u32x4 rg_h[S2X_RING], rg_l[S2X_RING];
  {
    const StackLayer L0 = p.layers[0];
#pragma unroll
    for (int s = 0; s < S2X_RING; s++) { rg_h[s] = S2X_WH(S2X_AOFF(L0, s)); rg_l[s] = S2X_WL(S2X_AOFF(L0, s)); }
  }

  // ---- state: the stack's first conv (1x1, in_ch -> 64) on the residual waves; zero on the skip waves ----
  f32x16 st[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ft++)
#pragma unroll
    for (int i = 0; i < 16; i++) st[ft][i] = 0.f;
  if (res_wave) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const sk_f32x4 bq = p.b_first >= 0 ? *reinterpret_cast<const sk_f32x4*>(p.params + p.b_first + 32 * mt + 8 * q + 4 * half) : sk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ft = 0; ft < FT; ft++)
#pragma unroll
        for (int j = 0; j < 4; j++) st[ft][4 * q + j] = bq[j];
    }
    const int KF = p.kp_first >> 4;
    const __amdgpu_buffer_rsrc_t rxi = sk_rsrc(p.x_in, (long)p.B * p.T * p.ldx_in);
    const __amdgpu_buffer_rsrc_t rfp = sk_rsrc16(p.fin_hi ? p.fin_hi : (const uint16_t*)p.x_in, (long)p.B * p.T * p.kp_first);
    for (int kc = 0; kc < KF; kc++) {
      const sk_u32x4 ah = S2X_WH(p.f_first + (mt * KF + kc) * 512), al = S2X_WL(p.f_first + (mt * KF + kc) * 512);
      const int c0 = 16 * kc + 8 * half;
      sk_u32x4 xa[FT], xc[FT];
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        const long nn = nbase + t0 - p.hl + row[ft];
        const int vo = (rin[ft] && c0 < p.in_ch) ? (int)((nn * p.ldx_in + c0) * 4) : SK_OOB;
        xa[ft] = __builtin_amdgcn_raw_buffer_load_b128(rxi, vo, 0, 0);
        xc[ft] = __builtin_amdgcn_raw_buffer_load_b128(rxi, vo + 16, 0, 0);
      }
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        sk_u32x2 h0, l0, h1, l1;
        sk_quad<true>(sk_u2f(xa[ft][0]), sk_u2f(xa[ft][1]), sk_u2f(xa[ft][2]), sk_u2f(xa[ft][3]), h0, l0);
        sk_quad<true>(sk_u2f(xc[ft][0]), sk_u2f(xc[ft][1]), sk_u2f(xc[ft][2]), sk_u2f(xc[ft][3]), h1, l1);
        const sk_u32x4 fb = {h0[0], h0[1], h1[0], h1[1]}, fl = {l0[0], l0[1], l1[0], l1[1]};
        if (mt == 0) {
          const long nn = nbase + t0 - p.hl + row[ft];
          const bool ro = rin[ft] && row[ft] >= p.hl && row[ft] < p.hl + p.tmo && p.fin_hi != nullptr;
          __builtin_amdgcn_raw_buffer_store_b128(fb, rfp, ro ? (int)((nn * p.kp_first + c0) * 2) : SK_OOB, 0, 0);
        }
        S2X_MMA(st[ft], ah, al, __builtin_bit_cast(bf16x8, fb), __builtin_bit_cast(bf16x8, fl))
      }
    }
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int i = 0; i < 16; i++) st[ft][i] = rin[ft] ? st[ft][i] : 0.f;
  }

  // ---- layer table, biases, guard rows, conditioning tile ----
  for (int i = tid; i < p.L * (int)(sizeof(StackLayer) / 4); i += NT)
    reinterpret_cast<int*>(lay_s)[i] = reinterpret_cast<const int*>(p.layers)[i];
  __syncthreads();
  {
    constexpr int NBI = 16 * 256 / NT;  // <= 16 blocks
    float bv[NBI];
#pragma unroll
    for (int k = 0; k < NBI; k++) {
      const int i = tid + k * NT, l = i >> 8, c = i & 255;
      bv[k] = 0.f;
      if (l < p.L) {
        const long long bo = c < 128 ? lay_s[l].b_conv : (c < 192 ? lay_s[l].b_out : lay_s[l].b_skip);
        // (the out conv's bias enters the residual update as fma(out + x, sqrt(.5), b * sqrt(.5)): stored pre-multiplied)
        if (bo >= 0) bv[k] = p.params[bo + (c < 128 ? c : (c < 192 ? c - 128 : c - 192))] * ((c >= 128 && c < 192) ? 0.70710678118654752440f : 1.f);
      }
    }
#pragma unroll
    for (int k = 0; k < NBI; k++) {
      const int i = tid + k * NT;
      if (i < p.L * 256) bias_s[i] = bv[k];
    }
  }
  for (int i = tid; i < SK_GUARD * XS / 16; i += NT) {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(xs)[i] = z4;
    reinterpret_cast<uint4*>(xs + (SK_GUARD + R) * XS)[i] = z4;
    reinterpret_cast<uint4*>(xl)[i] = z4;
    reinterpret_cast<uint4*>(xl + (SK_GUARD + R) * XS)[i] = z4;
  }
  if (AKC > 0) {
    // conditioning tile: one dword per channel, q4 quads per row, + 256 zero bytes behind the last row (the last rows'
    // fragment reads run past their end); every load issued before any is consumed
    const int q4 = p.cs_stride >> 4;  // quads per row
    const int NQ = R * q4 + 16;
    constexpr int PER = (R * 17 + 16 + NT - 1) / NT;  // aux_ch <= 64: <= 17 quads per row
    float av[PER][4];
#pragma unroll
    for (int it = 0; it < PER; it++) {
      const int idx = tid + it * NT, r = idx / q4, c4 = (idx - r * q4) << 2;
      const int tt = t0 - p.hl + r;
      const bool on = idx < NQ && r < R && tt >= 0 && tt < p.T;
      const long n = nbase + tt;
#pragma unroll
      for (int j = 0; j < 4; j++) av[it][j] = (on && c4 + j < p.aux_ch) ? p.c[n * p.ldc + c4 + j] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < PER; it++) {
      const int idx = tid + it * NT, r = idx / q4, c4 = (idx - r * q4) << 2;
      if (idx < NQ) {
        const int tt = t0 - p.hl + r;
        sk_u32x2 hi, lo;
        sk_quad<true>(av[it][0], av[it][1], av[it][2], av[it][3], hi, lo);
        const sk_u32x4 d = {(hi[0] & 0xffffu) | (lo[0] << 16), (hi[0] >> 16) | (lo[0] & 0xffff0000u),
                            (hi[1] & 0xffffu) | (lo[1] << 16), (hi[1] >> 16) | (lo[1] & 0xffff0000u)};
        *reinterpret_cast<sk_u32x4*>(cs + idx * 16) = d;
        if (p.cb_hi && c4 < p.aux_pad && r < R && tt >= 0 && tt < p.T && r >= p.hl && r < p.hl + p.tmo)
          *reinterpret_cast<sk_u32x2*>(p.cb_hi + (nbase + tt) * p.aux_pad + c4) = hi;
      }
    }
    // (channels aux4 .. aux_pad of the saved bf16 plane: zero, as stack2_fwd_kernel leaves them)
    if (p.cb_hi && p.aux_pad > (q4 << 2)) {
      for (int idx = tid; idx < R * ((p.aux_pad >> 2) - q4); idx += NT) {
        const int r = idx / ((p.aux_pad >> 2) - q4), c4 = ((idx - r * ((p.aux_pad >> 2) - q4)) + q4) << 2;
        const int tt = t0 - p.hl + r;
        if (tt >= 0 && tt < p.T && r >= p.hl && r < p.hl + p.tmo)
          *reinterpret_cast<sk_u32x2*>(p.cb_hi + (nbase + tt) * p.aux_pad + c4) = sk_u32x2{0u, 0u};
      }
    }
  }

  const float rs = 0.70710678118654752440f;
  const float scale = res_wave ? rs : 1.f;
  if (fh) __builtin_amdgcn_s_setprio(1);  // (as in stack2_fwd_kernel: the second-dispatched half loses every arbitration otherwise)

// the residual waves' state of frame tile ft as the next block's conv operand: hi and lo tiles, hi plane (what the plain-bf16
// weight gradient reads); zero outside the utterance (the conv's zero padding)
#define S2X_PUT_OPERAND_FT(ft)                                                                                  \
  {                                                                                                             \
    _Pragma("unroll") for (int g = 0; g < 2; g++) {                                                             \
      sk_u32x2 qh[2], ql[2];                                                                                    \
      _Pragma("unroll") for (int gg = 0; gg < 2; gg++) {                                                        \
        const int q = 2 * g + gg;                                                                               \
        sk_quad<true>(st[ft][4 * q], st[ft][4 * q + 1], st[ft][4 * q + 2], st[ft][4 * q + 3], qh[gg], ql[gg]); \
      }                                                                                                         \
      sk_u32x4 fh_ = sk_frag_bits(sk_swap_frag(qh[0], qh[1])), fl_ = sk_frag_bits(sk_swap_frag(ql[0], ql[1]));  \
      _Pragma("unroll") for (int j = 0; j < 4; j++) { fh_[j] &= rmask[ft]; fl_[j] &= rmask[ft]; }               \
      const int cb_ = (32 * mt + 16 * g + 8 * half) * 2;                                                        \
      *reinterpret_cast<sk_u32x4*>(xs + (SK_GUARD + row[ft]) * XS + cb_) = fh_;                                 \
      *reinterpret_cast<sk_u32x4*>(xl + (SK_GUARD + row[ft]) * XS + cb_) = fl_;                                 \
      __builtin_amdgcn_raw_buffer_store_b128(fh_, r_xh, voff_b[ft] + cb_, 0, 0);                                \
    }                                                                                                           \
  }
  if (res_wave) {
    const __amdgpu_buffer_rsrc_t r_xh = sk_rsrc16(save_b ? p.xb_hi : (const uint16_t*)p.x_in, P);
#pragma unroll
    for (int ft = 0; ft < FT; ft++) S2X_PUT_OPERAND_FT(ft)
  }
  __syncthreads();  // tables, guard rows, conditioning tile, block-0 operand tiles

  f32x16 acc[FT];
  sk_u32x4 wo_h[4], wo_l[4];
  for (int l = 0; l < p.L; l++) {
    const StackLayer LY = lay_s[l];
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
    // ---- dilated conv (+ conditioning 1x1), tap-major: one (hi, lo) A pair against the FT frame tiles; B pairs one k step
    // ahead (units u = s * FT + ft; a conditioning unit's slot holds the 8 packed dwords of its channels) ----
    {
      const unsigned char* xb0 = xs + (SK_GUARD + rb + l31 + LY.off0) * XS + half * 16;
      const unsigned char* cb0 = cs + (rb + l31) * p.cs_stride + half * 32;
      constexpr int NU = NS * FT, NBR = FT + 1;  // B ring: the FT units of the next k step + the one in use
      sk_u32x4 bh[NBR], bl[NBR];
#define S2X_BLOAD(u)                                                                                            \
  {                                                                                                             \
    const int s_ = (u) / FT, f_ = (u) - s_ * FT;                                                            \
    if (s_ < KT * 4) {                                                                                          \
      const unsigned char* src_ = xb0 + (s_ >> 2) * LY.dil * XS + (s_ & 3) * 32 + f_ * 32 * XS;                 \
      bh[(u) % NBR] = *reinterpret_cast<const sk_u32x4*>(src_);                                                 \
      bl[(u) % NBR] = *reinterpret_cast<const sk_u32x4*>(src_ + p.o_xlo);                                       \
    } else {                                                                                                    \
      const unsigned char* src_ = cb0 + (s_ - KT * 4) * 64 + f_ * 32 * p.cs_stride;                             \
      bh[(u) % NBR] = *reinterpret_cast<const sk_u32x4*>(src_);                                                 \
      bl[(u) % NBR] = *reinterpret_cast<const sk_u32x4*>(src_ + 16);                                            \
    }                                                                                                           \
  }
#pragma unroll
      for (int u = 0; u < FT; u++) S2X_BLOAD(u)
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const sk_u32x4 ah = rg_h[s % S2X_RING], al = rg_l[s % S2X_RING];
#pragma unroll
        for (int ft = 0; ft < FT; ft++) {
          const int u = s * FT + ft;
          if (u + FT < NU) S2X_BLOAD(u + FT)
          sk_u32x4 xh = bh[u % NBR], xo = bl[u % NBR];
          if (s >= KT * 4) {  // packed (hi | lo << 16) dwords of 8 channels -> the hi and the lo fragment
            const sk_u32x4 d0 = xh, d1 = xo;
            xh = sk_u32x4{(d0[0] & 0xffffu) | (d0[1] << 16), (d0[2] & 0xffffu) | (d0[3] << 16),
                          (d1[0] & 0xffffu) | (d1[1] << 16), (d1[2] & 0xffffu) | (d1[3] << 16)};
            xo = sk_u32x4{(d0[0] >> 16) | (d0[1] & 0xffff0000u), (d0[2] >> 16) | (d0[3] & 0xffff0000u),
                          (d1[0] >> 16) | (d1[1] & 0xffff0000u), (d1[2] >> 16) | (d1[3] & 0xffff0000u)};
          }
          S2X_MMA(acc[ft], ah, al, __builtin_bit_cast(bf16x8, xh), __builtin_bit_cast(bf16x8, xo))
        }
        // the slot is free: the pair S2X_RING k steps ahead
        if (s + S2X_RING < NS) {
          rg_h[s % S2X_RING] = S2X_WH(S2X_AOFF(LY, s + S2X_RING));
          rg_l[s % S2X_RING] = S2X_WL(S2X_AOFF(LY, s + S2X_RING));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef S2X_BLOAD
    }
    // out|skip fragments of this block, then the next block's first pairs: both in flight behind the gate and the barrier
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) { wo_h[k2] = S2X_WH(LY.f_os + (mt * 4 + k2) * 512); wo_l[k2] = S2X_WL(LY.f_os + (mt * 4 + k2) * 512); }
    if (l + 1 < p.L) {
      const StackLayer LN = lay_s[l + 1];
#pragma unroll
      for (int s = 0; s < S2X_RING; s++) { rg_h[s] = S2X_WH(S2X_AOFF(LN, s)); rg_l[s] = S2X_WL(S2X_AOFF(LN, s)); }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- gate: register j < 8 (tanh row) pairs with register j + 8 (sigmoid row) ----
    {
      const __amdgpu_buffer_rsrc_t r_zh = sk_rsrc16(save_b ? p.zb_hi + (long)l * P : (const uint16_t*)p.x_in, P);
      const bool ts_rec = p.ts_stride > 0;
      const long tsP = ts_rec ? (long)p.ts_stride : P;
      const __amdgpu_buffer_rsrc_t r_th = sk_rsrc16(save_b ? p.tb_hi + (long)l * tsP : (const uint16_t*)p.x_in, tsP);
      const __amdgpu_buffer_rsrc_t r_gh = sk_rsrc16(save_b ? p.sg_hi + (long)l * tsP : (const uint16_t*)p.x_in, tsP);
      const int cb = (16 * mt + 8 * half) * 2;  // this lane's 8-channel piece of a 64-channel row
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        sk_u32x2 zqh[2], zql[2], tq[2], sq[2], dummy;
#pragma unroll
        for (int gg = 0; gg < 2; gg++) {
          float ta[4], sb[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            ta[j] = sk_tanh(acc[ft][4 * gg + j], false);
            sb[j] = sk_sigmoid(acc[ft][8 + 4 * gg + j], false);
          }
          sk_quad<false>(ta[0], ta[1], ta[2], ta[3], tq[gg], dummy);
          sk_quad<false>(sb[0], sb[1], sb[2], sb[3], sq[gg], dummy);
          sk_quad<true>(ta[0] * sb[0], ta[1] * sb[1], ta[2] * sb[2], ta[3] * sb[3], zqh[gg], zql[gg]);
        }
        const sk_u32x4 zf = sk_frag_bits(sk_swap_frag(zqh[0], zqh[1])), zfl = sk_frag_bits(sk_swap_frag(zql[0], zql[1]));
        *reinterpret_cast<sk_u32x4*>(zs + row[ft] * XS + cb) = zf;
        *reinterpret_cast<sk_u32x4*>(zl + row[ft] * XS + cb) = zfl;
        if (ts_rec) {
          const sk_u32x4 tpc_ = {tq[0][0], tq[0][1], tq[1][0], tq[1][1]}, spc_ = {sq[0][0], sq[0][1], sq[1][0], sq[1][1]};
          __builtin_amdgcn_raw_buffer_store_b128(tpc_, r_th, voff_b[ft] + ts_delta, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(spc_, r_gh, voff_b[ft] + ts_delta, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(tq[0], tq[1])), r_th, voff_b[ft] + cb, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(sq[0], sq[1])), r_gh, voff_b[ft] + cb, 0, 0);
        }
        __builtin_amdgcn_raw_buffer_store_b128(zf, r_zh, voff_b[ft] + cb, 0, 0);
      }
    }
    __syncthreads();  // gate-output tiles complete; every tap read of the operand tiles done
    // ---- out | skip 1x1 on z, accumulated ON the state: residual waves x <- fma(x + out, sqrt(.5), b sqrt(.5)), skip waves
    // s <- (s + skip) + b; then the next block's operand ----
    {
      const float* bo = bias_s + l * 256 + 128 + 32 * mt + 4 * half;
      sk_f32x4 bsc[4];
#pragma unroll
      for (int q = 0; q < 4; q++) bsc[q] = *reinterpret_cast<const sk_f32x4*>(bo + 8 * q);
      const unsigned char* zb0 = zs + (rb + l31) * XS + half * 16;
      const __amdgpu_buffer_rsrc_t r_xh = sk_rsrc16(save_b ? p.xb_hi + (long)(l + 1) * P : (const uint16_t*)p.x_in, P);
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        bf16x8 zh[4], zo[4];
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
          zh[kc] = lds_frag(zb0 + ft * 32 * XS + kc * 32);
          zo[kc] = lds_frag(zb0 + (p.o_zlo - p.o_zs) + ft * 32 * XS + kc * 32);
        }
#pragma unroll
        for (int kc = 0; kc < 4; kc++) S2X_MMA(st[ft], wo_h[kc], wo_l[kc], zh[kc], zo[kc])
#pragma unroll
        for (int i = 0; i < 16; i++) st[ft][i] = __builtin_fmaf(st[ft][i], scale, bsc[i >> 2][i & 3]);
        if (res_wave && l + 1 < p.L) S2X_PUT_OPERAND_FT(ft)
      }
    }
    __syncthreads();  // next operand tiles complete; every read of the gate-output tiles done
  }

  // ---- the stack's head: relu(skip * sqrt(1/L)) -> 1x1 (64 -> 64) -> relu -> 1x1 (64 -> out_ch); both operands through the
  // LDS tiles (hi and lo), their hi halves are the planes the plain-bf16 backward reads ----
  {
    const __amdgpu_buffer_rsrc_t r_s = sk_rsrc16(p.head_hi ? p.head_hi : (const uint16_t*)p.x_in, P);
    const __amdgpu_buffer_rsrc_t r_h = sk_rsrc16(p.head_hi ? p.head_hi + P : (const uint16_t*)p.x_in, P);
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
            sk_quad<true>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);
          }
          const sk_u32x4 fs = sk_frag_bits(sk_swap_frag(qh[0], qh[1])), fsl = sk_frag_bits(sk_swap_frag(ql[0], ql[1]));
          const int cb_ = (32 * (mt - 2) + 16 * g + 8 * half) * 2;
          *reinterpret_cast<sk_u32x4*>(zs + row[ft] * XS + cb_) = fs;
          *reinterpret_cast<sk_u32x4*>(zl + row[ft] * XS + cb_) = fsl;
          __builtin_amdgcn_raw_buffer_store_b128(fs, r_s, voff_b[ft] + cb_, 0, 0);
        }
    }
    __syncthreads();
    if (res_wave) {
      sk_u32x4 w1h[4], w1l[4];
#pragma unroll
      for (int kc = 0; kc < 4; kc++) { w1h[kc] = S2X_WH(p.f_h1 + (mt * 4 + kc) * 512); w1l[kc] = S2X_WL(p.f_h1 + (mt * 4 + kc) * 512); }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const sk_f32x4 bq = p.b_h1 >= 0 ? *reinterpret_cast<const sk_f32x4*>(p.params + p.b_h1 + 32 * mt + 8 * q + 4 * half) : sk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[ft][4 * q + j] = bq[j];
      }
      const unsigned char* zb0 = zs + (rb + l31) * XS + half * 16;
#pragma unroll
      for (int kc = 0; kc < 4; kc++)
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
          S2X_MMA(acc[ft], w1h[kc], w1l[kc], lds_frag(zb0 + ft * 32 * XS + kc * 32), lds_frag(zb0 + (p.o_zlo - p.o_zs) + ft * 32 * XS + kc * 32))
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
            sk_quad<true>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);
          }
          const sk_u32x4 fh1 = sk_frag_bits(sk_swap_frag(qh[0], qh[1])), fl1 = sk_frag_bits(sk_swap_frag(ql[0], ql[1]));
          const int cb_ = (32 * mt + 16 * g + 8 * half) * 2;
          *reinterpret_cast<sk_u32x4*>(xs + (SK_GUARD + row[ft]) * XS + cb_) = fh1;
          *reinterpret_cast<sk_u32x4*>(xl + (SK_GUARD + row[ft]) * XS + cb_) = fl1;
          __builtin_amdgcn_raw_buffer_store_b128(fh1, r_h, voff_b[ft] + cb_, 0, 0);
        }
    }
    __syncthreads();
    if (32 * mt < p.out_ch) {
      sk_u32x4 w2h[4], w2l[4];
#pragma unroll
      for (int kc = 0; kc < 4; kc++) { w2h[kc] = S2X_WH(p.f_h2 + (mt * 4 + kc) * 512); w2l[kc] = S2X_WL(p.f_h2 + (mt * 4 + kc) * 512); }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int ch = 32 * mt + 8 * q + 4 * half;
        const sk_f32x4 bq = (p.b_h2 >= 0 && ch < p.out_ch) ? *reinterpret_cast<const sk_f32x4*>(p.params + p.b_h2 + ch) : sk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[ft][4 * q + j] = bq[j];
      }
      const unsigned char* hb0 = xs + (SK_GUARD + rb + l31) * XS + half * 16;
#pragma unroll
      for (int kc = 0; kc < 4; kc++)
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
          S2X_MMA(acc[ft], w2h[kc], w2l[kc], lds_frag(hb0 + ft * 32 * XS + kc * 32), lds_frag(hb0 + p.o_xlo + ft * 32 * XS + kc * 32))
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
  }
}

// eight waves: frame half 0 owns FT tiles, frame half 1 owns FT1 (the two waves of a SIMD are one of each)
template <int KT, int AKC, int FT, int FT1>
__global__ __launch_bounds__(512, 1) void stack2x_fwd_kernel(const StackP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int R = 32 * (FT + FT1);
  const int fh = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  if constexpr (FT1 == FT) {
    s2x_wave<KT, AKC, FT, R>(p, smem, fh * 32 * FT);
  } else {  // (the two copies hold the same barriers and the same workgroup-wide loops)
    if (fh == 0) s2x_wave<KT, AKC, FT, R>(p, smem, 0);
    else s2x_wave<KT, AKC, FT1, R>(p, smem, 32 * FT);
  }
}

// Window shapes: 192 rows (3 + 3 tiles) and, for k = 3, 160 rows (3 + 2).  The plan is the plain kernel's (same tmo and
// workgroup count: the saved planes and the data-gradient chain's windows do not depend on the forward's arithmetic) with
// this kernel's LDS carve-up: two operand tiles, two gate-output tiles, the packed conditioning tile.
int stack2x_fwd_plan(StackP& p) {
  if (!p.x_in || p.drop_p > 0.f) return CRK_ERR_UNSUPPORTED;  // generator stacks (folded first conv / head), no dropout
  const int rc = stack2_fwd_plan(p);
  if (rc != CRK_OK) return rc;
  if (p.fh != 2 || p.ft != 3) return CRK_ERR_UNSUPPORTED;
  const int R = 32 * (p.ft + (p.ft1 ? p.ft1 : p.ft));
  const int tile = (SK_GUARD * 2 + R) * SK_XS;
  int off = tile;
  p.o_xlo = off; off += tile;
  p.o_zs = off; off += R * SK_XS;
  p.o_zlo = off; off += R * SK_XS;
  p.o_cs = off; p.cs_stride = 0;
  if (p.aux_ch > 0) {
    int q4 = (p.aux_ch + 3) >> 2;
    if (!(q4 & 1)) q4++;               // an odd number of 16-byte quads per row: conflict-free ds_read_b128 down the rows
    p.cs_stride = q4 * 16;
    off += R * p.cs_stride + 256;
  }
  p.o_bias = off; off += p.L * 256 * 4;
  p.o_tab = off; off += p.L * (int)sizeof(StackLayer);
  p.lds_bytes = (off + 15) & ~15;
  return p.lds_bytes <= 160 * 1024 ? CRK_OK : CRK_ERR_UNSUPPORTED;
}

template <int KT, int AKC>
static int s2x_launch_shape(const StackP& p, dim3 grid, hipStream_t s) {
#define S2X_GO(FTV, FT1V)                                                                                            \
  {                                                                                                                  \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      if (hipFuncSetAttribute((const void*)stack2x_fwd_kernel<KT, AKC, FTV, FT1V>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              160 * 1024) != hipSuccess) return CRK_ERR_HIP;                                         \
      attr = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((stack2x_fwd_kernel<KT, AKC, FTV, FT1V>), grid, dim3(512), p.lds_bytes, s, p);                \
  }
  if (p.ft1 == 0) S2X_GO(3, 3)
  else if constexpr (KT == 3) { if (p.ft1 == 2) S2X_GO(3, 2) else return CRK_ERR_UNSUPPORTED; }
  else return CRK_ERR_UNSUPPORTED;
#undef S2X_GO
  return CRK_OK;
}

int launch_stack2x_fwd(const StackP& p, hipStream_t s) {
  dim3 grid(p.B * p.tiles_per_utt);
  const double nfr = (double)p.B * p.T;
  conv_prof_bytes(1, nfr * (256.0 + 4.0 * p.aux_ch + 256.0 + (p.xb_hi ? 512.0 * p.L + 2.0 * (p.aux_ch > 0 ? p.aux_pad : 0) : 0.0)));
  conv_prof_begin(1, 2.0 * nfr * (p.L * (128.0 * (64.0 * p.ktaps + p.aux_ch) + 128.0 * 64.0) + 64.0 * p.in_ch + 64.0 * 64.0 + 64.0 * p.out_ch), s);
  const int akc = p.aux_ch > 0 ? (p.aux_ch + 15) / 16 : 0;
  int rc;
  if (p.ktaps == 3) rc = akc == 0 ? s2x_launch_shape<3, 0>(p, grid, s) : CRK_ERR_UNSUPPORTED;
  else if (akc == 0) rc = s2x_launch_shape<5, 0>(p, grid, s);
  else if (akc == 1) rc = s2x_launch_shape<5, 1>(p, grid, s);
  else if (akc <= 3) rc = s2x_launch_shape<5, 3>(p, grid, s);
  else rc = s2x_launch_shape<5, 4>(p, grid, s);
  conv_prof_end(1, s);
  if (rc != CRK_OK) return rc;
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
