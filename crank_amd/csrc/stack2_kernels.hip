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
// [4] next operand [5] wait at barrier B, summed over the blocks, [6] prologue, [7] whole kernel.
#ifdef S2_PROF
__device__ unsigned long long s2_prof_buf[256 * 8 * 8];
__device__ unsigned long long s2_prof_res[1024 * 4];  // per workgroup: start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID
extern "C" int crk_debug_s2_prof(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(s2_prof_buf), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? 0 : 2;
}
extern "C" int crk_debug_s2_res(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(s2_prof_res), sizeof(unsigned long long) * 1024 * 4) == hipSuccess ? 0 : 2;
}
#define S2_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); pacc_[i] += now_ - plast_; plast_ = now_; }
#else
#define S2_T(i)
#endif

template <int KT, int AKC, int FT, int FH, bool DROP>
__global__ __launch_bounds__(256 * FH, 2) void stack2_fwd_kernel(const StackP p) {
  constexpr int R = 32 * FT * FH, XS = SK_XS, NT = 256 * FH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 3, fh = FH > 1 ? wave >> 2 : 0;
  const bool res_wave = mt < 2;  // carries the residual stream (else: the skip sum)
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;
  const long nbase = (long)b * p.T;
  const long P = (long)p.B * p.T * 64;
#ifdef S2_PROF
  unsigned long long pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast_ = __builtin_readcyclecounter();
  const unsigned long long pstart_ = plast_, preal_ = __builtin_amdgcn_s_memrealtime();
#endif

  unsigned char* xs = smem;             // [SK_GUARD + R + SK_GUARD][XS] block input as the conv sees it
  unsigned char* zs = smem + p.o_zs;    // [R][XS] gate output
  unsigned char* cs = smem + p.o_cs;    // [R][XS] conditioning (AKC > 0)
  StackLayer* lay_s = reinterpret_cast<StackLayer*>(smem + p.o_tab);
  float* bias_s = reinterpret_cast<float*>(smem + p.o_bias);  // [L][256]: conv 128 | out 64 | skip 64

  // ---- this lane's FT frames ----
  int row[FT], voff_st[FT], voff_b[FT];
  bool rin[FT];
  const bool save_b = p.xb_hi != nullptr;
  const int ch_st = 32 * (mt & 1) + 4 * half;  // first channel of quad 0 of this lane's state tile (residual or skip plane)
#pragma unroll
  for (int ft = 0; ft < FT; ft++) {
    row[ft] = fh * 32 * FT + ft * 32 + l31;
    const int t = t0 - p.hl + row[ft];
    rin[ft] = t >= 0 && t < p.T;
    const bool rout = rin[ft] && row[ft] >= p.hl && row[ft] < p.hl + p.tmo;
    // fp32 [N,64] planes (block-0 input read by the residual waves, skip sum written by the skip waves)
    voff_st[ft] = (res_wave ? rin[ft] : rout) ? (int)(((nbase + t) * 64 + ch_st) * 4) : SK_OOB;
    // bf16 [N,64] planes: byte offset of channel 0 of this lane's frame
    voff_b[ft] = (rout && save_b) ? (int)(((nbase + t) * 64) * 2) : SK_OOB;
  }

  // ---- weights: A fragments straight from L2 (fragment order: 16 bytes per lane, 1 KB per wave-load) ----
  const uint16_t* wl = p.whi + lane * 8;
#define S2_WLOAD(off) (*reinterpret_cast<const sk_u32x4*>(wl + (off)))
  constexpr int NWB = FT >= 4 ? 2 : 3;  // register sets of tap weights (taps in flight); the 4-tile shape has no room for 3
  sk_u32x4 wa[NWB][4];
  sk_u32x4 wos[4];
  sk_u32x4 wax[AKC > 0 ? AKC : 1];
  {
    const StackLayer L0 = p.layers[0];
#pragma unroll
    for (int kc = 0; kc < 4; kc++)
#pragma unroll
      for (int tp = 0; tp < NWB - 1; tp++) wa[tp][kc] = S2_WLOAD(L0.f_conv + ((tp * 4 + mt) * 4 + kc) * 512);
  }

  // ---- state: residual stream (block-0 input) or zero (skip sum) ----
  f32x16 st[FT];
  {
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

  // ---- layer table, biases, guard rows, conditioning tile ----
  // (table first, then every bias load of a thread in flight together: a loop that reads the table entry and the
  // bias behind it per element is two dependent L2 round trips per iteration - it was 11 % of the kernel)
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
        if (bo >= 0) bv[k] = p.params[bo + (c < 128 ? c : (c < 192 ? c - 128 : c - 192))];
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
  }
  if (AKC > 0) {
    // conditioning tile: 16 quads per row (64 channels, zero beyond aux_ch), every load issued before any is consumed
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
  // the second-dispatched half of the workgroup loses every arbitration for the SIMD it shares with an older wave
  // (taps: 3.3 k cycles for waves 0-3, 4.9 k for waves 4-7, the difference spent waiting at the barrier): static priority
  if (FH > 1 && fh) __builtin_amdgcn_s_setprio(1);

// the residual waves' state as the next block's conv operand: 2 x 16 channels per frame -> two 16-byte pieces
// to the LDS tile and to the bf16 plane the weight gradient reads (dropout applied, as the conv sees it)
#define S2_PUT_OPERAND(layer)                                                                                   \
  if (res_wave) {                                                                                               \
    const unsigned long long dseed = p.drop_seed + 0x9E3779B97F4A7C15ull * (unsigned long long)((layer) + 1);   \
    const __amdgpu_buffer_rsrc_t r_xh = sk_rsrc16(save_b ? p.xb_hi + (long)(layer) * P : (const uint16_t*)p.skip, P); \
    _Pragma("unroll") for (int ft = 0; ft < FT; ft++) {                                                         \
      _Pragma("unroll") for (int g = 0; g < 2; g++) {                                                           \
        sk_u32x2 qh[2], ql[2];                                                                                  \
        _Pragma("unroll") for (int gg = 0; gg < 2; gg++) {                                                      \
          const int q = 2 * g + gg;                                                                             \
          float v[4];                                                                                           \
          _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                       \
            v[j] = st[ft][4 * q + j];                                                                           \
            if (DROP && p.drop_p > 0.f && rin[ft])                                                              \
              v[j] *= dropout_scale(dseed, (unsigned long long)(nbase + t0 - p.hl + row[ft]) * 64 + 32 * mt + 8 * q + 4 * half + j, p.drop_p); \
          }                                                                                                     \
          sk_quad<false>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);                                               \
        }                                                                                                       \
        const sk_u32x4 fh_ = sk_frag_bits(sk_swap_frag(qh[0], qh[1]));                                          \
        const int cb = (32 * mt + 16 * g + 8 * half) * 2;                                                       \
        *reinterpret_cast<sk_u32x4*>(xs + (SK_GUARD + row[ft]) * XS + cb) = fh_;                                \
        __builtin_amdgcn_raw_buffer_store_b128(fh_, r_xh, voff_b[ft] + cb, 0, 0);                               \
      }                                                                                                         \
    }                                                                                                           \
  }
  S2_PUT_OPERAND(0)
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
    // ---- dilated conv (+ conditioning 1x1) as one flat list of k steps: step s = 4 * tap + kc, then AKC steps on
    // the conditioning tile.  Software pipeline, pinned with scheduling fences (left alone the compiler sinks every
    // load to just in front of its consumer: the MFMAs then wait for an LDS round trip each and - because it reuses
    // the registers of consumed fragments for the next weight loads - for a full L2 round trip per tap):
    //   weights  : the A fragments of tap t + 2 are requested at the first step of tap t (three register sets),
    //   operands : the B fragments of step s + 2 are read from LDS in front of the MFMAs of step s.
    const unsigned char* xb0 = xs + (SK_GUARD + fh * 32 * FT + l31 + LY.off0) * XS + half * 16;
    const unsigned char* cb0 = cs + (fh * 32 * FT + l31) * XS + half * 16;
    constexpr int NS = KT * 4 + AKC, NBQ = FT >= 4 ? 2 : 3;  // NBQ - 1 steps of B fragments in flight
    bf16x8 bq[NBQ][FT];
#define S2_BREAD(sx)                                                                                            \
  {                                                                                                             \
    const unsigned char* src_ = (sx) < KT * 4 ? xb0 + ((sx) >> 2) * LY.dil * XS + ((sx) & 3) * 32 : cb0 + ((sx) - KT * 4) * 32; \
    _Pragma("unroll") for (int ft = 0; ft < FT; ft++) bq[(sx) % NBQ][ft] = lds_frag(src_ + ft * 32 * XS);         \
  }
    S2_BREAD(0)
    if (NBQ > 2) S2_BREAD(1)
#pragma unroll
    for (int sx = 0; sx < NS; sx++) {
      const int tap = sx >> 2, kc = sx & 3;
      if (sx < KT * 4 && kc == 0) {
        if (tap + NWB - 1 < KT) {
#pragma unroll
          for (int k2 = 0; k2 < 4; k2++) wa[(tap + NWB - 1) % NWB][k2] = S2_WLOAD(LY.f_conv + (((tap + NWB - 1) * 4 + mt) * 4 + k2) * 512);
        }
        if (tap == KT - 2) {  // conditioning and out|skip fragments of this block
          if (AKC > 0) {
#pragma unroll
            for (int k2 = 0; k2 < AKC; k2++) wax[k2] = S2_WLOAD(LY.f_aux + (mt * 4 + k2) * 512);
          }
#pragma unroll
          for (int k2 = 0; k2 < 4; k2++) wos[k2] = S2_WLOAD(LY.f_os + (mt * 4 + k2) * 512);
        }
      }
      if (sx + NBQ - 1 < NS) S2_BREAD(sx + NBQ - 1)
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 a = __builtin_bit_cast(bf16x8, sx < KT * 4 ? wa[tap % NWB][kc] : wax[sx < KT * 4 ? 0 : sx - KT * 4]);
#pragma unroll
      for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(a, bq[sx % NBQ][ft], acc[ft]);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef S2_BREAD
    S2_T(0)
    // the next block's first two taps: in flight behind the gate and the 1x1
    if (l + 1 < p.L) {
      const StackLayer LN = lay_s[l + 1];
#pragma unroll
      for (int kc = 0; kc < 4; kc++)
#pragma unroll
        for (int tp = 0; tp < NWB - 1; tp++) wa[tp][kc] = S2_WLOAD(LN.f_conv + ((tp * 4 + mt) * 4 + kc) * 512);
    }
    // ---- gate: register j < 8 (tanh row) pairs with register j + 8 (sigmoid row) ----
    {
      const __amdgpu_buffer_rsrc_t r_zh = sk_rsrc16(save_b ? p.zb_hi + (long)l * P : (const uint16_t*)p.skip, P);
      const __amdgpu_buffer_rsrc_t r_th = sk_rsrc16(save_b ? p.tb_hi + (long)l * P : (const uint16_t*)p.skip, P);
      const __amdgpu_buffer_rsrc_t r_gh = sk_rsrc16(save_b ? p.sg_hi + (long)l * P : (const uint16_t*)p.skip, P);
      const int cb = (16 * mt + 8 * half) * 2;  // this lane's 8-channel piece of a 64-channel row
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        sk_u32x2 zq[2], tq[2], sq[2], dummy;
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
          sk_quad<false>(ta[0] * sb[0], ta[1] * sb[1], ta[2] * sb[2], ta[3] * sb[3], zq[gg], dummy);
        }
        const sk_u32x4 zf = sk_frag_bits(sk_swap_frag(zq[0], zq[1]));
        *reinterpret_cast<sk_u32x4*>(zs + row[ft] * XS + cb) = zf;
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(tq[0], tq[1])), r_th, voff_b[ft] + cb, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(sq[0], sq[1])), r_gh, voff_b[ft] + cb, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(zf, r_zh, voff_b[ft] + cb, 0, 0);
      }
    }
    S2_T(1)
    __syncthreads();  // gate-output tile complete; every tap read of the operand tile done
    S2_T(2)
    // ---- out | skip 1x1 on z: tile mt of [out 0-31 | out 32-63 | skip 0-31 | skip 32-63] ----
    {
      const float* bo = bias_s + l * 256 + 128 + 32 * mt + 4 * half;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const sk_f32x4 bq = *reinterpret_cast<const sk_f32x4*>(bo + 8 * q);
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[ft][4 * q + j] = bq[j];
      }
      const unsigned char* zb0 = zs + (fh * 32 * FT + l31) * XS + half * 16;
      constexpr int NZ = FT >= 4 ? 2 : 4;  // k steps of the gate-output tile read ahead of their MFMAs
      bf16x8 zq[NZ][FT];
#pragma unroll
      for (int kc = 0; kc < NZ; kc++)
#pragma unroll
        for (int ft = 0; ft < FT; ft++) zq[kc][ft] = lds_frag(zb0 + ft * 32 * XS + kc * 32);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kc = 0; kc < 4; kc++) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, wos[kc]);
#pragma unroll
        for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(a, zq[kc % NZ][ft], acc[ft]);
        if (NZ < 4 && kc + NZ < 4) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ft = 0; ft < FT; ft++) zq[kc % NZ][ft] = lds_frag(zb0 + ft * 32 * XS + (kc + NZ) * 32);
        }
      }
    }
    // residual waves: x <- (out + x) * sqrt(.5), zero outside the utterance; skip waves: s <- s + skip
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const float o = (acc[ft][i] + st[ft][i]) * scale;
        st[ft][i] = rin[ft] ? o : 0.f;
      }
    S2_T(3)
    if (l + 1 < p.L) S2_PUT_OPERAND(l + 1)
    S2_T(4)
    __syncthreads();  // next operand tile complete; every read of the gate-output tile done
    S2_T(5)
  }

  // ---- running skip sum of the window's own frames (skip waves) ----
  if (!res_wave) {
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
    for (int i = 0; i < 8; i++) s2_prof_buf[(blockIdx.x * 8 + wave) * 8 + i] = pacc_[i];
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

// Window shapes: (ft, fh) = (2, 2) 128 rows / 8 waves, (3, 2) 192 rows / 8 waves, (4, 1) 128 rows / 4 waves with two
// workgroups per CU (each SIMD then carries one wave of either: while one is in its gate / update phase (VALU) the
// other one is in its taps (MFMA)).  Cost model: MFMA rounds per SIMD = ceil(workgroups / resident slots) x rows per
// SIMD-resident wave pair; CRK_S2_CFG=<ft><fh> overrides (debugging / A-B timing).
int stack2_fwd_plan(StackP& p) {
  if ((p.ktaps != 3 && p.ktaps != 5) || p.max_off > SK_GUARD || p.aux_ch > 64 || p.L > 16) return CRK_ERR_UNSUPPORTED;
  if (p.ktaps == 3 && p.aux_ch > 0) return CRK_ERR_UNSUPPORTED;  // (no caller; keeps the instantiation count down)
  static int cfg_env = -1;
  if (cfg_env < 0) { const char* e = getenv("CRK_S2_CFG"); cfg_env = e ? atoi(e) : 0; }
  static const int shapes[3][2] = {{2, 2}, {3, 2}, {4, 1}};
  int best = -1; double best_cost = 0;
  for (int i = 0; i < 3; i++) {
    const int ft = shapes[i][0], fh = shapes[i][1];
    if (cfg_env && cfg_env != ft * 10 + fh) continue;
    if (p.drop_p > 0.f && i != 0) continue;  // (the mask hashing needs the registers of the larger shapes)
    const int tmo = 32 * ft * fh - p.hl - p.hr;
    if (tmo < 16) continue;
    const long wgs = (long)p.B * ceil_div(p.T, tmo);
    const long slots = S2_NCU * (fh == 1 ? 2 : 1);
    const double cost = (double)((wgs + slots - 1) / slots) * ft * (fh == 1 ? S2_PAIR_FACTOR : 1.0);
    if (best < 0 || cost <= best_cost) { best = i; best_cost = cost; }
  }
  if (best < 0) return CRK_ERR_UNSUPPORTED;
  p.ft = shapes[best][0]; p.fh = shapes[best][1];
  const int R = 32 * p.ft * p.fh;
  p.tmo = R - p.hl - p.hr;
  p.tiles_per_utt = ceil_div(p.T, p.tmo);
  p.tmo = ceil_div(p.T, p.tiles_per_utt);
  int off = (SK_GUARD * 2 + R) * SK_XS;
  p.o_zs = off; off += R * SK_XS;
  p.o_cs = off; if (p.aux_ch > 0) off += R * SK_XS;
  p.o_bias = off; off += p.L * 256 * 4;
  p.o_tab = off; off += p.L * (int)sizeof(StackLayer);
  p.lds_bytes = (off + 15) & ~15;
  return p.lds_bytes <= (p.fh == 1 ? 80 : 160) * 1024 ? CRK_OK : CRK_ERR_UNSUPPORTED;
}

template <int KT, int AKC, bool DROP>
static int s2_launch_shape(const StackP& p, dim3 grid, hipStream_t s) {
#define S2_GO(FTV, FHV)                                                                                              \
  {                                                                                                                  \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      if (hipFuncSetAttribute((const void*)stack2_fwd_kernel<KT, AKC, FTV, FHV, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              160 * 1024) != hipSuccess) return CRK_ERR_HIP;                                         \
      attr = true;                                                                                                   \
      if (getenv("CRK_DEBUG_OCC")) {                                                                                 \
        int nb_ = -1;                                                                                                \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, stack2_fwd_kernel<KT, AKC, FTV, FHV, DROP>, 256 * FHV, p.lds_bytes); \
        fprintf(stderr, "[crank_hip] stack2_fwd<%d,%d,%d,%d>: %d blocks/CU at %d B LDS, grid %u\n", KT, AKC, FTV, FHV, nb_, p.lds_bytes, grid.x); \
      }                                                                                                              \
    }                                                                                                                \
    hipLaunchKernelGGL((stack2_fwd_kernel<KT, AKC, FTV, FHV, DROP>), grid, dim3(256 * FHV), p.lds_bytes, s, p);       \
  }
  if (p.ft == 2) S2_GO(2, 2) else if constexpr (!DROP) { if (p.ft == 3) S2_GO(3, 2) else S2_GO(4, 1) }
#undef S2_GO
  return CRK_OK;
}

int launch_stack2_fwd(const StackP& p, hipStream_t s) {
  dim3 grid(p.B * p.tiles_per_utt);
  const double nfr = (double)p.B * p.T;
  conv_prof_bytes(1, nfr * (256.0 + 4.0 * p.aux_ch + 256.0 + (p.xb_hi ? 512.0 * p.L + 2.0 * (p.aux_ch > 0 ? p.aux_pad : 0) : 0.0)));
  conv_prof_begin(1, 2.0 * nfr * p.L * (128.0 * (64.0 * p.ktaps + p.aux_ch) + 128.0 * 64.0), s);
  const int akc = p.aux_ch > 0 ? (p.aux_ch + 15) / 16 : 0;
  int rc = CRK_OK;
  if (p.ktaps == 3) rc = s2_launch_shape<3, 0, false>(p, grid, s);
  else if (p.drop_p > 0.f) {
    if (akc) return CRK_ERR_UNSUPPORTED;
    rc = s2_launch_shape<5, 0, true>(p, grid, s);
  } else if (akc == 0) rc = s2_launch_shape<5, 0, false>(p, grid, s);
  else if (akc == 1) rc = s2_launch_shape<5, 1, false>(p, grid, s);
  else if (akc <= 3) rc = s2_launch_shape<5, 3, false>(p, grid, s);
  else rc = s2_launch_shape<5, 4, false>(p, grid, s);
  conv_prof_end(1, s);
  if (rc != CRK_OK) return rc;
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
