// Tile-pipelined channel-split forward of ALL gated residual blocks of a generator stack (plain-bf16 arithmetic; round 6).
//
// Same arithmetic, same planes, same summation order per output element as stack2_fwd_kernel (stack2_kernels.hip; reference:
// parallel_wavegan's ResidualBlock.forward chained, call sites crank/net/module/vqvae2.py:237-273) - bit-identical results
// (tests/test_gpu_properties.py).  Same decomposition too: a wave owns one 32-channel tile (mt) of one frame half (fh) of a
// 192- / 160-row window, FT tiles of 32 frames, weights as MFMA A fragments straight from L2 into registers.
//
// What differs is the SCHEDULE of a block.  stack2_fwd_kernel runs a block as two workgroup-wide phases, each closed by a
// barrier: "taps + gate of all FT tiles", then "out|skip 1x1 + state update + next operand of all FT tiles".  The second phase
// holds 12 of the block's 84 MFMAs per wave and is a chain of latencies - LDS round trip, four dependent MFMAs, ~50 VALU
// instructions, LDS write - that nothing covers: it and the two barrier waits are 36 % of a block's cycles
// (profiles/round3_stack2_fwd_phase_cycles_pinned_schedule.txt: taps 6.9 k, 1x1 2.2 k, barriers 1.7 k).  Here a block is
// THREE equal stages, one tile each, closed by one barrier each:
//
//     stage s of block l:   T_l(tile s)  =  taps (+ conditioning) + gate of tile s            -> z_l(tile s) in LDS
//                           O_l(tile s-1) = out|skip 1x1 + state update of tile s - 1         -> x_{l+1}(tile s - 1) in LDS
//                           (s = 0: O_{l-1}(tile FT - 1))
//
// so every interval between two barriers carries 24 - 28 MFMAs of one accumulator chain plus an independent 4-MFMA chain
// with its VALU tail, and the two waves of a SIMD run the two parts in opposite order (frame half 0: O then T, frame half
// 1: T then O): one wave's VALU-dense part sits beside its partner's MFMA-dense part.
//
// What makes the overlap legal:
//   * a tap of T_{l+1}(tile j) reaches at most SK_GUARD = 16 rows into tiles j - 1 and j + 1.  Frame half 0 walks its tiles
//     from the window's middle outwards (2, 1, 0), frame half 1 likewise (3, 4, 5): the first tile of a block's walk has
//     only first and second tiles for neighbours, whose x_{l+1} rows were written one and two stages earlier - the third
//     tile's rows, written in the same stage as T_{l+1}(first tile), are never within its reach;
//   * the operand tile is double buffered by block parity: O_l writes x_{l+1} into the buffer T_l does not read;
//   * z_l(tile s) is read (stage s + 1) before z_{l+1}(tile s) is written (three stages later).
// A wave of a 2-tile frame half (160-row windows: 3 + 2 tiles) idles in T of stage 2 and in O of stage 0; 2 + 2 windows
// cannot be pipelined this way (the first tile's neighbour would be the last) and stay with stack2_fwd_kernel, as do dropout
// stacks and stacks without the folded first conv / head.
//
// Weights are resident per block (KT x 4 + AKC + 4 fragments = 64 - 112 registers).  In the block's last T stage every tap's
// register set is re-requested from the next block's weights right behind the tap's last MFMA, the out|skip fragments behind
// the block's last O: the requests are in flight across a barrier and at least 20 MFMAs.
#include "conv_kernels.h"

#include "stack_common.h"

// Phase cycles (tools/s2p_phase_cycles.py builds a second library with -DS2P_PROF): per workgroup and wave the shader cycles in
// [0] prologue [1] O parts [2] T chains [3] gates [4] waits at the stage barriers [5] head [6] whole kernel
#ifdef S2P_PROF
__device__ unsigned long long s2p_prof_buf[256 * 8 * 8];
extern "C" int crk_debug_s2p_prof(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(s2p_prof_buf), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? 0 : 2;
}
#define S2P_MARK(i) { const unsigned long long now_ = __builtin_readcyclecounter(); pacc_[i] += now_ - plast_; plast_ = now_; }
#else
#define S2P_MARK(i)
#endif

template <int KT, int AKC, int FT, int R, bool DESC>
__device__ __forceinline__ void s2p_wave(const StackP& p, unsigned char* smem, const int rb) {
  constexpr int XS = SK_XS, NT = 512;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 3;
  const bool res_wave = mt < 2;  // carries the residual stream (else: the skip sum)
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;
  const long nbase = (long)b * p.T;
  const long P = (long)p.B * p.T * 64;
#ifdef S2P_PROF
  unsigned long long pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast_ = __builtin_readcyclecounter();
  const unsigned long long pstart_ = plast_;
#endif

  // operand tile of block l: buffer l & 1 (selected as an integer offset from the LDS base: a runtime-indexed array of LDS
  // pointers decays to generic pointers and every access through it becomes a FLAT instruction)
  unsigned char* xs0 = smem;            // [SK_GUARD + R + SK_GUARD][XS]
  unsigned char* zs = smem + p.o_zs;    // [R][XS] gate output
  unsigned char* cs = smem + p.o_cs;    // [R][XS] conditioning (AKC > 0)
  StackLayer* lay_s = reinterpret_cast<StackLayer*>(smem + p.o_tab);
  float* bias_s = reinterpret_cast<float*>(smem + p.o_bias);  // [L][256]: conv 128 | out 64 | skip 64
#define S2P_XS(par) (smem + ((par) ? p.o_xlo : 0))

  // ---- this lane's FT frames ----
  int row[FT], voff_st[FT], voff_b[FT];
  unsigned rmask[FT];
  bool rin[FT];
  const bool save_b = p.xb_hi != nullptr;
  const int ch_st = 32 * (mt & 1) + 4 * half;
#pragma unroll
  for (int ft = 0; ft < FT; ft++) {
    row[ft] = rb + ft * 32 + l31;
    const int t = t0 - p.hl + row[ft];
    rin[ft] = t >= 0 && t < p.T;
    rmask[ft] = rin[ft] ? 0xffffffffu : 0u;
    const bool rout = rin[ft] && row[ft] >= p.hl && row[ft] < p.hl + p.tmo;
    voff_st[ft] = (res_wave ? rin[ft] : rout) ? (int)(((nbase + t) * 64 + ch_st) * 4) : SK_OOB;
    voff_b[ft] = (rout && save_b) ? (int)(((nbase + t) * 64) * 2) : SK_OOB;
  }
  // lane-record layout of the tanh / sigmoid planes (StackP::ts_stride; see stack2_kernels.hip)
  const int ts_delta = half * 512 + (mt >> 1) * 2048 + (mt & 1) * 1024 - 112 * (int)((nbase + t0 - p.hl + row[0]) & 31);

  // ---- weights: A fragments straight from L2 (fragment order: 16 bytes per lane, 1 KB per wave-load) ----
  const uint16_t* wl = p.whi + lane * 8;
#define S2P_WLOAD(off) (*reinterpret_cast<const sk_u32x4*>(wl + (off)))
  sk_u32x4 wa[KT][4];
  sk_u32x4 wos[4];
  sk_u32x4 wax[AKC > 0 ? AKC : 1];
  {
    const StackLayer L0 = p.layers[0];
#pragma unroll
    for (int tp = 0; tp < KT; tp++)
#pragma unroll
      for (int kc = 0; kc < 4; kc++) wa[tp][kc] = S2P_WLOAD(L0.f_conv + ((tp * 4 + mt) * 4 + kc) * 512);
    if (AKC > 0) {
#pragma unroll
      for (int k2 = 0; k2 < AKC; k2++) wax[k2] = S2P_WLOAD(L0.f_aux + (mt * 4 + k2) * 512);
    }
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) wos[k2] = S2P_WLOAD(L0.f_os + (mt * 4 + k2) * 512);
  }

  // ---- state: the stack's first conv (1x1, in_ch -> 64) right here (as stack2_fwd_kernel<FOLD>) ----
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
      const bf16x8 a = __builtin_bit_cast(bf16x8, S2P_WLOAD(p.f_first + (mt * KF + kc) * 512));
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
        const sk_u32x4 fb = {pack_bf2(sk_u2f(xa[ft][0]), sk_u2f(xa[ft][1])), pack_bf2(sk_u2f(xa[ft][2]), sk_u2f(xa[ft][3])),
                             pack_bf2(sk_u2f(xc[ft][0]), sk_u2f(xc[ft][1])), pack_bf2(sk_u2f(xc[ft][2]), sk_u2f(xc[ft][3]))};
        if (mt == 0) {
          const long nn = nbase + t0 - p.hl + row[ft];
          const bool ro = rin[ft] && row[ft] >= p.hl && row[ft] < p.hl + p.tmo && p.fin_hi != nullptr;
          __builtin_amdgcn_raw_buffer_store_b128(fb, rfp, ro ? (int)((nn * p.kp_first + c0) * 2) : SK_OOB, 0, 0);
        }
        st[ft] = mfma_bf16(a, __builtin_bit_cast(bf16x8, fb), st[ft]);
      }
    }
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int i = 0; i < 16; i++) st[ft][i] = rin[ft] ? st[ft][i] : 0.f;
  }

  // ---- layer table, biases, guard rows of BOTH operand buffers, conditioning tile ----
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
    reinterpret_cast<uint4*>(xs0)[i] = z4;
    reinterpret_cast<uint4*>(xs0 + (SK_GUARD + R) * XS)[i] = z4;
    reinterpret_cast<uint4*>(xs0 + p.o_xlo)[i] = z4;
    reinterpret_cast<uint4*>(xs0 + p.o_xlo + (SK_GUARD + R) * XS)[i] = z4;
  }
  if (AKC > 0) {
    constexpr int NQ = R * 16, PER = (NQ + NT - 1) / NT;
    float av[PER][4];
#pragma unroll
    for (int it = 0; it < PER; it++) {
      const int idx = tid + it * NT, r = idx >> 4, c4 = (idx & 15) << 2;
      const int tt = t0 - p.hl + r;
      const bool on = idx < NQ && tt >= 0 && tt < p.T;
      const long n = nbase + tt;
#pragma unroll
      for (int j = 0; j < 4; j++) av[it][j] = (on && c4 + j < p.aux_ch) ? p.c[n * p.ldc + c4 + j] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < PER; it++) {
      const int idx = tid + it * NT, r = idx >> 4, c4 = (idx & 15) << 2;
      if (idx < NQ) {
        const int tt = t0 - p.hl + r;
        sk_u32x2 hi, lo;
        sk_quad<false>(av[it][0], av[it][1], av[it][2], av[it][3], hi, lo);
        *reinterpret_cast<sk_u32x2*>(cs + r * XS + c4 * 2) = hi;
        if (p.cb_hi && c4 < p.aux_pad && tt >= 0 && tt < p.T && r >= p.hl && r < p.hl + p.tmo)
          *reinterpret_cast<sk_u32x2*>(p.cb_hi + (nbase + tt) * p.aux_pad + c4) = hi;
      }
    }
  }

  const float rs = 0.70710678118654752440f;
  const float scale = res_wave ? rs : 1.f;
  if (!DESC) __builtin_amdgcn_s_setprio(1);  // (the second-dispatched half loses every arbitration otherwise: stack2_kernels.hip)

// the residual waves' state of frame tile ft as the next block's conv operand (masked to zero outside the utterance) -> the
// operand buffer xsn and the bf16 plane r_xh the weight gradient reads
#define S2P_PUT_OPERAND(ft, xsn, r_xh)                                                                          \
  {                                                                                                             \
    _Pragma("unroll") for (int g = 0; g < 2; g++) {                                                             \
      sk_u32x2 qh[2], ql[2];                                                                                    \
      _Pragma("unroll") for (int gg = 0; gg < 2; gg++) {                                                        \
        const int q = 2 * g + gg;                                                                               \
        sk_quad<false>(st[ft][4 * q], st[ft][4 * q + 1], st[ft][4 * q + 2], st[ft][4 * q + 3], qh[gg], ql[gg]); \
      }                                                                                                         \
      sk_u32x4 fh_ = sk_frag_bits(sk_swap_frag(qh[0], qh[1]));                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; j++) fh_[j] &= rmask[ft];                                         \
      const int cb_ = (32 * mt + 16 * g + 8 * half) * 2;                                                        \
      *reinterpret_cast<sk_u32x4*>((xsn) + (SK_GUARD + row[ft]) * XS + cb_) = fh_;                              \
      __builtin_amdgcn_raw_buffer_store_b128(fh_, r_xh, voff_b[ft] + cb_, 0, 0);                                \
    }                                                                                                           \
  }
  if (res_wave) {
    const __amdgpu_buffer_rsrc_t r_xh = sk_rsrc16(save_b ? p.xb_hi : (const uint16_t*)p.x_in, P);
#pragma unroll
    for (int ft = 0; ft < FT; ft++) S2P_PUT_OPERAND(ft, xs0, r_xh)
  }
  __syncthreads();  // tables, guard rows, conditioning tile, block-0 operand tile
  S2P_MARK(0)

  // u-th tile of this wave's walk: frame half 0 from the window's middle outwards
#define S2P_FT(u) (DESC ? FT - 1 - (u) : (u))
  constexpr int MC = KT * 4, M = MC + AKC;
  constexpr int NG = KT + (AKC > 0 ? 1 : 0);  // bursts of a chain: one per tap, one for the conditioning steps

// T_lt(tile u): taps (+ conditioning) and gate of one frame tile, operand buffer xsc, layer record LYT (block lt).  RELOAD:
// this is the block's last T stage - every tap's fragments are re-requested from layer record LYN behind the tap's last
// MFMA (and the conditioning fragments behind theirs).
#define S2P_CHAIN(u, lt, LYT, xsc, RELOAD, LYN)                                                                 \
  {                                                                                                             \
    constexpr int ft_ = S2P_FT(u);                                                                              \
    {                                                                                                           \
      const float* bc = bias_s + (lt) * 256 + 16 * mt + 4 * half;                                              \
      _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                           \
        const sk_f32x4 bq = *reinterpret_cast<const sk_f32x4*>(bc + (q < 2 ? 8 * q : 64 + 8 * (q - 2)));       \
        _Pragma("unroll") for (int j = 0; j < 4; j++) acc[4 * q + j] = bq[j];                                   \
      }                                                                                                         \
    }                                                                                                           \
    const unsigned char* xb_ = (xsc) + (SK_GUARD + rb + ft_ * 32 + l31 + (LYT).off0) * XS + half * 16;          \
    const unsigned char* cb0_ = cs + (rb + ft_ * 32 + l31) * XS + half * 16;                                    \
    const int ts_ = (LYT).dil * XS;                                                                             \
    /* One tile = ONE accumulator: every MFMA of the chain depends on the one before it, and an instruction between two      \
       dependent MFMAs - a fragment read, a wait that does not even stall - costs ~45 cycles on top of the MFMA's 32           \
       (MI355X_MICROARCH.md, "one extra issue slot between two MFMAs on the same accumulator"; measured here: 98 cycles per   \
       MFMA with a read and a wait in each gap).  So the chain runs as BURSTS of four MFMAs (one tap, or the conditioning      \
       steps) with nothing between them: the group's four B fragments are requested a whole group ahead, one wait (the asm   \
       statement makes the compiler finish every operand of the burst in front of it) and then the MFMAs back to back. */     \
    bf16x8 bg[2][4];                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 4 && j < M; j++)                                                      \
      bg[0][j] = lds_frag(j < MC ? xb_ + (j & 3) * 32 : cb0_ + (j - MC) * 32);                                  \
    _Pragma("unroll") for (int g = 0; g < NG; g++) {                                                            \
      if (g + 1 < NG) {                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                         \
          const int m2 = 4 * (g + 1) + j;                                                                       \
          if (m2 < M) bg[(g + 1) & 1][j] = lds_frag(m2 < MC ? xb_ + (m2 >> 2) * ts_ + (m2 & 3) * 32 : cb0_ + (m2 - MC) * 32); \
        }                                                                                                       \
      }                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      if (g < KT) {                                                                                             \
        asm volatile("" ::"v"(bg[g & 1][0]), "v"(bg[g & 1][1]), "v"(bg[g & 1][2]), "v"(bg[g & 1][3]),              \
                     "v"(wa[g < KT ? g : 0][0]), "v"(wa[g < KT ? g : 0][1]), "v"(wa[g < KT ? g : 0][2]), "v"(wa[g < KT ? g : 0][3])); \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                                           \
          acc = mfma_bf16(__builtin_bit_cast(bf16x8, wa[g < KT ? g : 0][j]), bg[g & 1][j], acc);                \
      } else {                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < AKC; j++) asm volatile("" ::"v"(bg[g & 1][j]), "v"(wax[j]));       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int j = 0; j < AKC; j++)                                                         \
          acc = mfma_bf16(__builtin_bit_cast(bf16x8, wax[j]), bg[g & 1][j], acc);                               \
      }                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      if (RELOAD && g < KT) {                                                                                   \
        _Pragma("unroll") for (int k2 = 0; k2 < 4; k2++)                                                        \
          wa[g < KT ? g : 0][k2] = S2P_WLOAD((LYN).f_conv + (((g * 4) + mt) * 4 + k2) * 512);                   \
      }                                                                                                         \
      if (RELOAD && AKC > 0 && g == NG - 1) {                                                                   \
        _Pragma("unroll") for (int k2 = 0; k2 < AKC; k2++) wax[k2] = S2P_WLOAD((LYN).f_aux + (mt * 4 + k2) * 512); \
      }                                                                                                         \
    }                                                                                                           \
    S2P_MARK(2)                                                                                                 \
  }
// gate of the tile whose chain ran in the wave's previous half-stage (acc) -> z tile in LDS, tanh / sigmoid / z planes
#define S2P_GATE(u, lt)                                                                                         \
  {                                                                                                             \
    constexpr int ft_ = S2P_FT(u);                                                                              \
    {                                                                                                           \
      const __amdgpu_buffer_rsrc_t r_zh = sk_rsrc16(save_b ? p.zb_hi + (long)(lt) * P : (const uint16_t*)p.x_in, P);     \
      const __amdgpu_buffer_rsrc_t r_th = sk_rsrc16(save_b ? p.tb_hi + (long)(lt) * tsP : (const uint16_t*)p.x_in, tsP); \
      const __amdgpu_buffer_rsrc_t r_gh = sk_rsrc16(save_b ? p.sg_hi + (long)(lt) * tsP : (const uint16_t*)p.x_in, tsP); \
      const int cbz = (16 * mt + 8 * half) * 2;                                                                 \
      sk_u32x2 zq_[2], tq_[2], sq_[2], dm_;                                                                     \
      _Pragma("unroll") for (int gg = 0; gg < 2; gg++) {                                                        \
        float ta[4], sb[4];                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                         \
          ta[j] = sk_tanh(acc[4 * gg + j], false);                                                              \
          sb[j] = sk_sigmoid(acc[8 + 4 * gg + j], false);                                                       \
        }                                                                                                       \
        sk_quad<false>(ta[0], ta[1], ta[2], ta[3], tq_[gg], dm_);                                               \
        sk_quad<false>(sb[0], sb[1], sb[2], sb[3], sq_[gg], dm_);                                               \
        sk_quad<false>(ta[0] * sb[0], ta[1] * sb[1], ta[2] * sb[2], ta[3] * sb[3], zq_[gg], dm_);               \
      }                                                                                                         \
      const sk_u32x4 zf = sk_frag_bits(sk_swap_frag(zq_[0], zq_[1]));                                           \
      *reinterpret_cast<sk_u32x4*>(zs + row[ft_] * XS + cbz) = zf;                                              \
      if (ts_rec) {                                                                                             \
        const sk_u32x4 tpc_ = {tq_[0][0], tq_[0][1], tq_[1][0], tq_[1][1]}, spc_ = {sq_[0][0], sq_[0][1], sq_[1][0], sq_[1][1]}; \
        __builtin_amdgcn_raw_buffer_store_b128(tpc_, r_th, voff_b[ft_] + ts_delta, 0, 0);                       \
        __builtin_amdgcn_raw_buffer_store_b128(spc_, r_gh, voff_b[ft_] + ts_delta, 0, 0);                       \
      } else {                                                                                                  \
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(tq_[0], tq_[1])), r_th, voff_b[ft_] + cbz, 0, 0); \
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(sq_[0], sq_[1])), r_gh, voff_b[ft_] + cbz, 0, 0); \
      }                                                                                                         \
      __builtin_amdgcn_raw_buffer_store_b128(zf, r_zh, voff_b[ft_] + cbz, 0, 0);                                \
    }                                                                                                           \
    S2P_MARK(3)                                                                                                 \
  }

// O_lo(tile u): out | skip 1x1 on z accumulated ON the state, state update, and (PUT) the next block's operand rows of the tile
#define S2P_O(u, lo, PUT, xsn)                                                                                  \
  {                                                                                                             \
    constexpr int ft_ = S2P_FT(u);                                                                              \
    const unsigned char* zb_ = zs + (rb + ft_ * 32 + l31) * XS + half * 16;                                     \
    bf16x8 zq_[4];                                                                                              \
    _Pragma("unroll") for (int kc = 0; kc < 4; kc++) zq_[kc] = lds_frag(zb_ + kc * 32);                         \
    const float* bo_ = bias_s + (lo) * 256 + 128 + 32 * mt + 4 * half;                                          \
    sk_f32x4 bsc_[4];                                                                                           \
    _Pragma("unroll") for (int q = 0; q < 4; q++) bsc_[q] = *reinterpret_cast<const sk_f32x4*>(bo_ + 8 * q);    \
    _Pragma("unroll") for (int kc = 0; kc < 4; kc++) st[ft_] = mfma_bf16(__builtin_bit_cast(bf16x8, wos[kc]), zq_[kc], st[ft_]); \
    _Pragma("unroll") for (int i = 0; i < 16; i++) st[ft_][i] = __builtin_fmaf(st[ft_][i], scale, bsc_[i >> 2][i & 3]); \
    if ((PUT) && res_wave) {                                                                                    \
      const __amdgpu_buffer_rsrc_t r_xh_ = sk_rsrc16(save_b ? p.xb_hi + (long)((lo) + 1) * P : (const uint16_t*)p.x_in, P); \
      S2P_PUT_OPERAND(ft_, xsn, r_xh_)                                                                          \
    }                                                                                                           \
    S2P_MARK(1)                                                                                                 \
  }

  const bool ts_rec = p.ts_stride > 0;
  const long tsP = ts_rec ? (long)p.ts_stride : P;

  // ---- the pipeline: six half-stages per block, a barrier behind each.  A wave alternates "chain" half-stages (the MFMA
  // chain of one tile: matrix pipe) with "gate + O" half-stages (transcendentals, packing, LDS / plane stores, a 4-MFMA
  // chain: vector ALU and memory); frame half 1 runs one half-stage behind frame half 0, so the two waves of a SIMD are in
  // opposite kinds of half-stage at any time.  Slot of a wave's own walk:
  //     0 chain(t0)   1 gate(t0) + O_{l-1}(t2)   2 chain(t1)   3 gate(t1) + O_l(t0)   4 chain(t2)   5 gate(t2) + O_l(t1)
  // (2-tile waves: 4 = O_l(t1), 5 idle.)  Global half-stage g = 6 l + slot (+ 1 for frame half 1).  chain_{l+1}(t0) at
  // 6 l + 6 reads x_{l+1} of t0 (written at 6 l + 3), t1 (6 l + 5) and of the other half's t0 (6 l + 4 / 6 l + 3); the rows
  // O_l(t2) writes at 6 l + 7 are out of its reach.  Both halves execute 6 L + 2 barriers.
  f32x16 acc;
  if (!DESC) { S2P_MARK(0) __syncthreads(); S2P_MARK(4) }
  for (int l = 0; l < p.L; l++) {
    const bool has_next = l + 1 < p.L;
    const StackLayer LY = lay_s[l];
    const StackLayer LN = lay_s[has_next ? l + 1 : l];  // (the last block re-requests its own weights: no branch in the chain)
    unsigned char* xsc = S2P_XS(l & 1);
    unsigned char* xsn = S2P_XS((l + 1) & 1);
    // slot 0
    S2P_CHAIN(0, l, LY, xsc, false, LN)
    __syncthreads();
    S2P_MARK(4)
    // slot 1
    S2P_GATE(0, l)
    if (l > 0) {
      if constexpr (FT == 3) {
        S2P_O(2, l - 1, true, xsc)
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) wos[k2] = S2P_WLOAD(LY.f_os + (mt * 4 + k2) * 512);
      }
    }
    __syncthreads();
    S2P_MARK(4)
    // slot 2
    S2P_CHAIN(1, l, LY, xsc, FT == 2, LN)
    __syncthreads();
    S2P_MARK(4)
    // slot 3
    S2P_GATE(1, l)
    S2P_O(0, l, has_next, xsn)
    __syncthreads();
    S2P_MARK(4)
    // slot 4
    if constexpr (FT == 3) {
      S2P_CHAIN(2, l, LY, xsc, true, LN)
    } else {
      S2P_O(1, l, has_next, xsn)
#pragma unroll
      for (int k2 = 0; k2 < 4; k2++) wos[k2] = S2P_WLOAD(LN.f_os + (mt * 4 + k2) * 512);
    }
    __syncthreads();
    S2P_MARK(4)
    // slot 5
    if constexpr (FT == 3) {
      S2P_GATE(2, l)
      S2P_O(1, l, has_next, xsn)
    }
    __syncthreads();
    S2P_MARK(4)
  }
  if constexpr (FT == 3) { S2P_O(2, p.L - 1, false, xs0) }
  __syncthreads();
  S2P_MARK(4)
  if (DESC) { __syncthreads(); S2P_MARK(4) }

  // ---- the stack's head right here: relu(skip * sqrt(1/L)) -> 1x1 (64 -> 64) -> relu -> 1x1 (64 -> out_ch) (as
  // stack2_fwd_kernel<FOLD>; both operands pass through LDS tiles and are the planes their weight gradients read) ----
  {
    f32x16 acc[FT];
    unsigned char* xs = xs0;
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
      sk_u32x4 w1[4];
#pragma unroll
      for (int kc = 0; kc < 4; kc++) w1[kc] = S2P_WLOAD(p.f_h1 + (mt * 4 + kc) * 512);
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
        for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(__builtin_bit_cast(bf16x8, w1[kc]), lds_frag(zb0 + ft * 32 * XS + kc * 32), acc[ft]);
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
      sk_u32x4 w2[4];
#pragma unroll
      for (int kc = 0; kc < 4; kc++) w2[kc] = S2P_WLOAD(p.f_h2 + (mt * 4 + kc) * 512);
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
        for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(__builtin_bit_cast(bf16x8, w2[kc]), lds_frag(hb0 + ft * 32 * XS + kc * 32), acc[ft]);
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
#ifdef S2P_PROF
  S2P_MARK(5)
  pacc_[6] = __builtin_readcyclecounter() - pstart_;
  if (blockIdx.x < 256 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; i++) s2p_prof_buf[(blockIdx.x * 8 + wave) * 8 + i] = pacc_[i];
  }
#endif
#undef S2P_CHAIN
#undef S2P_GATE
#undef S2P_O
#undef S2P_PUT_OPERAND
#undef S2P_FT
#undef S2P_XS
#undef S2P_WLOAD
}

// FT0 / FT1: tiles per wave of frame half 0 / 1 (3, 3) or (3, 2); waves 0-3 = frame half 0 walk their tiles downwards
template <int KT, int AKC, int FT0, int FT1>
__global__ __launch_bounds__(512, 2) void stack2p_fwd_kernel(const StackP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int R = 32 * (FT0 + FT1);
  const int fh = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  if (fh == 0) s2p_wave<KT, AKC, FT0, R, true>(p, smem, 0);
  else s2p_wave<KT, AKC, FT1, R, false>(p, smem, 32 * FT0);
}

// LDS carve-up of the pipelined kernel on top of stack2_fwd_plan's window shape: a second operand buffer (StackP::o_xlo).
// CRK_OK: p.pipe is set and launch_stack2_fwd takes this kernel.
int stack2p_fwd_plan(StackP& p) {
  p.pipe = 0;
  static int on = -1;
  if (on < 0) { const char* e = getenv("CRK_S2_PIPE"); on = e ? atoi(e) : 1; }
  if (!on || p.x_in == nullptr || p.drop_p > 0.f || p.fh != 2 || p.ft != 3) return CRK_ERR_UNSUPPORTED;
  const int ft1 = p.ft1 ? p.ft1 : p.ft;
  if (ft1 != 3 && ft1 != 2) return CRK_ERR_UNSUPPORTED;
  const int akc = p.aux_ch > 0 ? (p.aux_ch + 15) / 16 : 0;
  if (!((p.ktaps == 3 && akc == 0) || (p.ktaps == 5 && (akc == 0 || akc == 3)))) return CRK_ERR_UNSUPPORTED;
  const int R = 32 * (p.ft + ft1);
  int off = (SK_GUARD * 2 + R) * SK_XS;
  p.o_xlo = off; off += (SK_GUARD * 2 + R) * SK_XS;
  p.o_zs = off; off += R * SK_XS;
  p.o_cs = off; if (p.aux_ch > 0) off += R * SK_XS;
  p.o_bias = off; off += p.L * 256 * 4;
  p.o_tab = off; off += p.L * (int)sizeof(StackLayer);
  const int lds = (off + 15) & ~15;
  if (lds > 160 * 1024) return CRK_ERR_UNSUPPORTED;
  p.lds_bytes = lds;
  p.pipe = 1;
  return CRK_OK;
}

template <int KT, int AKC, int FT0, int FT1>
static int s2p_go(const StackP& p, dim3 grid, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)stack2p_fwd_kernel<KT, AKC, FT0, FT1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return CRK_ERR_HIP;
    attr = true;
  }
  hipLaunchKernelGGL((stack2p_fwd_kernel<KT, AKC, FT0, FT1>), grid, dim3(512), p.lds_bytes, s, p);
  return CRK_OK;
}

// (called by launch_stack2_fwd between its profiling brackets)
int launch_stack2p_fwd_body(const StackP& p, hipStream_t s) {
  dim3 grid(p.B * p.tiles_per_utt);
  const int akc = p.aux_ch > 0 ? (p.aux_ch + 15) / 16 : 0;
  const int ft1 = p.ft1 ? p.ft1 : p.ft;
  if (p.ktaps == 3 && ft1 == 2) return s2p_go<3, 0, 3, 2>(p, grid, s);
  if (p.ktaps == 3) return s2p_go<3, 0, 3, 3>(p, grid, s);
  if (akc == 0 && ft1 == 3) return s2p_go<5, 0, 3, 3>(p, grid, s);
  if (akc == 3 && ft1 == 3) return s2p_go<5, 3, 3, 3>(p, grid, s);
  if (akc == 0) return s2p_go<5, 0, 3, 2>(p, grid, s);
  return s2p_go<5, 3, 3, 2>(p, grid, s);
}
