// Fused forward of ALL gated residual blocks of a stack (SURVEY.md K3 x L) in one launch.
//
// One workgroup of NW waves owns a window of R = 32*NW frames of one utterance: TMo = R - halo
// output frames plus the receptive-field halo of the whole stack, recomputed per window.
// The residual stream and the running skip sum never leave the chip: each wave keeps its
// 32 frames x 64 channels of both in fp32 MFMA-layout registers across layers; only a bf16
// operand copy of the residual lives in LDS, with zero guard rows so that a dilated tap is a
// plain row offset.  Weights stream through LDS one tap at a time, double buffered and
// prefetched into registers one chunk ahead, across layer boundaries.  HBM traffic per layer
// is what the backward pass needs (block input, tanh, sigmoid, z of the window's own frames)
// and nothing else; with saving disabled (no-grad forwards) it is zero.
//
// MFMA role assignment: A = weights (rows = output channels), B = activations (columns =
// frames).  The accumulator layout then gives every lane ONE frame and, per register quad,
// FOUR consecutive channels: epilogue traffic is 16-byte row stores / 8-byte LDS writes with a
// single per-lane validity.  The gate output z goes from the accumulator layout to the B
// operand layout of the 1x1 out|skip product without touching LDS: the two lanes that own a
// frame exchange one quad per 16 channels with v_permlane32_swap.
//
// Occupancy: two waves per SIMD (either one 8-wave workgroup or two 4-wave workgroups per
// CU, <= 256 registers per lane), so one wave's LDS / HBM waits hide behind the other's
// MFMAs.  The bf16x3 variant (hi/lo operand planes, 3 MFMAs per product) runs the 4-wave
// shape with one workgroup per CU: it exists for parity, not for speed.
//
// Reference semantics: parallel_wavegan ResidualBlock.forward chained as in
// ParallelWaveGANGenerator.forward / ResidualParallelWaveGANDiscriminator.forward
// (call sites crank/net/module/vqvae2.py:237-273,
// crank/bin/train.py:108-118).  Frames outside [0,T) are forced to zero after every
// block, which is exactly the zero padding each Conv1d applies to its input.
#include "conv_kernels.h"

#include "stack_common.h"

template <bool PRECISE, bool DROP, int NW>
__global__ __launch_bounds__(NW * 64, PRECISE ? 1 : 2) void stack_fwd_kernel(const StackP p) {
  constexpr int NT = NW * 64, R = NW * 32, XS = SK_XS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;  // first output frame of this window
  const long nbase = (long)b * p.T;
  const int CS = p.aux_pad * 2 + 16;

  unsigned char* xs_hi = smem;  // [SK_GUARD + R + SK_GUARD][XS]
  unsigned char* xs_lo = smem + p.o_xlo;
  unsigned char* cs_hi = smem + p.o_chi;
  unsigned char* cs_lo = smem + p.o_clo;
  // weight buffer i as an OFFSET from the LDS base: a runtime-indexed array of pointers would decay to
  // generic pointers and every weight-fragment access to a FLAT instruction (vmcnt + lgkmcnt, i.e.
  // serialised with the global weight prefetch) instead of ds_read / ds_write
#define WS_HI(i) (smem + p.o_whi + ((PRECISE || (i) == 0) ? 0 : p.w_bytes))
  unsigned char* ws_lo = smem + p.o_wlo;

  // ---- this lane's frame ----
  const int row = wave * 32 + l31;         // window row
  const int t = t0 - p.hl + row;           // frame (may be outside [0,T))
  const bool rin = t >= 0 && t < p.T;      // inside the utterance
  const bool rout = rin && row >= p.hl && row < p.hl + p.tmo;  // one of the window's own output frames
  // byte offset of (this frame, channel 4*half) inside an [N,64] fp32 plane; register quad g of
  // channel tile h2 sits (h2*32 + 8*g)*4 bytes further (an immediate)
  const int voff_in = rin ? (int)(((nbase + t) * 64 + 4 * half) * 4) : SK_OOB;
  const int voff_out = rout ? voff_in : SK_OOB;
#define SK_QOFF(h2, g) (((h2) * 32 + 8 * (g)) * 4)
  const long P = (long)p.B * p.T * 64;  // one [N,64] plane of the saved workspace
  // bf16 operand planes [N,64] (weight-gradient inputs): this lane's 8-channel fragment of 16-group kc
  // sits at ((frame*64 + 8*half) + 16*kc) * 2 bytes
  const bool save_b = p.xb_hi != nullptr;
  const int voff_b = (rout && save_b) ? (int)(((nbase + t) * 64 + 8 * half) * 2) : SK_OOB;

  // ---- weight-chunk copy geometry (fixed per thread) ----
  // every chunk (tap, conditioning 1x1, out|skip pair) is 128 rows x 64 k: 8 pieces per row
  int wdst[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int idx = tid + u * NT;
    wdst[u] = idx < 1024 ? (idx >> 3) * XS + (idx & 7) * 16 : -1;  // NT = 384: half of the last round is unused
  }
#define SK_C1(u, dhi)                                                               \
  if (u * NT < 1024 && (1024 % NT == 0 || wdst[u] >= 0)) {                          \
    *reinterpret_cast<sk_u32x4*>((dhi) + wdst[u]) = wr.h##u;                        \
    if (PRECISE) *reinterpret_cast<sk_u32x4*>(ws_lo + wdst[u]) = wr.l##u;           \
  }
#define SK_COMMIT(dhi) { SK_C1(0, dhi) SK_C1(1, dhi) SK_C1(2, dhi) SK_C1(3, dhi) }

  // ---- first weight chunk on its way; guard rows zeroed; aux tile staged ----
  SkRegs wr;
  sk_fetch<PRECISE, NT>(wr, p.whi + p.layers[0].w_conv, p.wlo + p.layers[0].w_conv, 1024, tid);
  // ---- residual stream (block 0 input) into MFMA-layout registers; operand copy to LDS ----
  // register i of tile h2 <-> channel h2*32 + (i&3) + 8*(i>>2) + 4*half of this lane's frame
  f32x16 res[2], skp[2];
  {
    const __amdgpu_buffer_rsrc_t rx0 = sk_rsrc(p.x0, P);
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const sk_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rx0, voff_in + (SK_QOFF(h2, g)), 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          res[h2][4 * g + j] = sk_u2f(q[j]);
          skp[h2][4 * g + j] = 0.f;
        }
      }
  }
  // ---- layer table and every bias of the stack into LDS once: inside the block loop they would be
  // global loads sitting in front of the first MFMAs of every stage (a full L2 latency, twice per block) ----
  StackLayer* lay_s = reinterpret_cast<StackLayer*>(smem + p.o_tab);   // [L]
  float* bias_s = reinterpret_cast<float*>(smem + p.o_bias);            // [L][256]: conv 128 | out 64 | skip 64
  for (int i = tid; i < p.L * 256; i += NT) {
    const int l = i >> 8, c = i & 255;
    const StackLayer Y = p.layers[l];
    const long long off = c < 128 ? (Y.b_conv >= 0 ? Y.b_conv + c : -1)
                        : (c < 192 ? (Y.b_out >= 0 ? Y.b_out + (c - 128) : -1) : (Y.b_skip >= 0 ? Y.b_skip + (c - 192) : -1));
    // (the out conv's bias enters the residual update as fma(out + x, sqrt(.5), b * sqrt(.5)): stored pre-multiplied)
    bias_s[i] = off >= 0 ? p.params[off] * ((c >= 128 && c < 192) ? 0.70710678118654752440f : 1.f) : 0.f;
  }
  for (int i = tid; i < p.L * (int)(sizeof(StackLayer) / 4); i += NT)
    reinterpret_cast<int*>(lay_s)[i] = reinterpret_cast<const int*>(p.layers)[i];
  for (int i = tid; i < SK_GUARD * XS / 16; i += NT) {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(xs_hi)[i] = z4;
    reinterpret_cast<uint4*>(xs_hi + (SK_GUARD + R) * XS)[i] = z4;
    if (PRECISE) {
      reinterpret_cast<uint4*>(xs_lo)[i] = z4;
      reinterpret_cast<uint4*>(xs_lo + (SK_GUARD + R) * XS)[i] = z4;
    }
  }
  if (p.aux_ch > 0) {
    // conditioning tile: <= 8 quads per thread; every load is issued before the first one is consumed
    // (a loop that loads and converts per iteration pays one memory round trip per iteration)
    const int qc = p.aux_pad >> 2;
    float av[8][4];
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int idx = tid + it * NT;
      const int r = idx / qc, c4 = (idx - r * qc) << 2;
      const int tt = t0 - p.hl + r;
      const bool on = idx < R * qc && tt >= 0 && tt < p.T;
      const long n = nbase + tt;
#pragma unroll
      for (int j = 0; j < 4; j++) av[it][j] = (on && c4 + j < p.aux_ch) ? p.c[n * p.ldc + c4 + j] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int idx = tid + it * NT;
      if (idx < R * qc) {
        const int r = idx / qc, c4 = (idx - r * qc) << 2;
        const int tt = t0 - p.hl + r;
        sk_u32x2 hi, lo;
        sk_quad<PRECISE>(av[it][0], av[it][1], av[it][2], av[it][3], hi, lo);
        *reinterpret_cast<sk_u32x2*>(cs_hi + r * CS + c4 * 2) = hi;
        if (PRECISE) *reinterpret_cast<sk_u32x2*>(cs_lo + r * CS + c4 * 2) = lo;
        if (p.cb_hi && tt >= 0 && tt < p.T && r >= p.hl && r < p.hl + p.tmo) {
          const long o = (nbase + tt) * p.aux_pad + c4;
          *reinterpret_cast<sk_u32x2*>(p.cb_hi + o) = hi;
          if (PRECISE) *reinterpret_cast<sk_u32x2*>(p.cb_lo + o) = lo;
        }
      }
    }
  }

  unsigned char* my_xs_hi = xs_hi + (SK_GUARD + row) * XS + 8 * half * 2;
  unsigned char* my_xs_lo = xs_lo + (SK_GUARD + row) * XS + 8 * half * 2;
// the block input as the conv sees it (dropout applied): quads -> 8-channel fragments (lane-pair
// exchange) -> one 16-byte LDS write per 16 channels, and the same bytes to the bf16 plane the
// weight gradient reads
#define SK_PUT_OPERAND(layer)                                                                                   \
  {                                                                                                             \
    const unsigned long long dseed = (DROP ? crk_seed(p.drop_seed, p.drop_seed_ptr) : 0ull) + 0x9E3779B97F4A7C15ull * (unsigned long long)((layer) + 1);   \
    const __amdgpu_buffer_rsrc_t r_xh = sk_rsrc16(save_b ? p.xb_hi + (long)(layer) * P : (const uint16_t*)p.skip, P); \
    const __amdgpu_buffer_rsrc_t r_xl = sk_rsrc16((save_b && PRECISE) ? p.xb_lo + (long)(layer) * P : (const uint16_t*)p.skip, P); \
    _Pragma("unroll") for (int kc = 0; kc < 4; kc++) {                                                          \
      const int h2 = kc >> 1, g0 = (kc & 1) * 2;                                                                \
      sk_u32x2 qh[2], ql[2];                                                                                    \
      _Pragma("unroll") for (int gg = 0; gg < 2; gg++) {                                                        \
        const int g = g0 + gg;                                                                                  \
        float v[4];                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                         \
          v[j] = res[h2][4 * g + j];                                                                            \
          if (DROP && p.drop_p > 0.f && rin)                                                                    \
            v[j] *= dropout_scale(dseed, (unsigned long long)(nbase + t) * 64 + h2 * 32 + 8 * g + 4 * half + j, \
                                  p.drop_p);                                                                    \
        }                                                                                                       \
        sk_quad<PRECISE>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);                                               \
      }                                                                                                         \
      const sk_u32x4 fh = sk_frag_bits(sk_swap_frag(qh[0], qh[1]));                                             \
      *reinterpret_cast<sk_u32x4*>(my_xs_hi + kc * 32) = fh;                                                    \
      __builtin_amdgcn_raw_buffer_store_b128(fh, r_xh, voff_b + (kc * 32), 0, 0);                                     \
      if (PRECISE) {                                                                                            \
        const sk_u32x4 fl = sk_frag_bits(sk_swap_frag(ql[0], ql[1]));                                           \
        *reinterpret_cast<sk_u32x4*>(my_xs_lo + kc * 32) = fl;                                                  \
        __builtin_amdgcn_raw_buffer_store_b128(fl, r_xl, voff_b + (kc * 32), 0, 0);                                   \
      }                                                                                                         \
    }                                                                                                           \
  }
  SK_PUT_OPERAND(0)
  SK_COMMIT(WS_HI(0))

  const float rs = 0.70710678118654752440f;
  int cur = 0;  // weight buffer holding the chunk about to be consumed
  f32x16 acc[4];
  const int nch = p.ktaps + (p.aux_ch > 0 ? 1 : 0);
  const unsigned char* wf_lo = ws_lo + l31 * XS + half * 16;

// one B fragment against the 4 output-channel tiles of the current weight chunk
#define SK_MMA(wf_hi, kc, x_hi, x_lo)                                              \
  _Pragma("unroll") for (int nt = 0; nt < 4; nt++) {                               \
    const bf16x8 w_hi = lds_frag((wf_hi) + nt * 32 * XS + (kc) * 32);              \
    acc[nt] = mfma_bf16(w_hi, x_hi, acc[nt]);                                      \
    if (PRECISE) {                                                                 \
      const bf16x8 w_lo = lds_frag(wf_lo + nt * 32 * XS + (kc) * 32);              \
      acc[nt] = mfma_bf16(w_hi, x_lo, acc[nt]);                                    \
      acc[nt] = mfma_bf16(w_lo, x_hi, acc[nt]);                                    \
    }                                                                              \
  }
// accumulators start from the bias of their output channel (rows of D = channels); bcol = column of
// the block's 256-entry bias row (conv 0..127, out 128..191, skip 192..255)
#define SK_INIT_ACC(nt, bcol)                                                      \
  _Pragma("unroll") for (int g = 0; g < 4; g++) {                                  \
    const sk_f32x4 bq = *reinterpret_cast<const sk_f32x4*>(bias_s + l * 256 + (bcol) + 8 * g + 4 * half); \
    _Pragma("unroll") for (int j = 0; j < 4; j++) acc[nt][4 * g + j] = bq[j];      \
  }

  __syncthreads();  // layer table, biases, operand tile, first weight chunk: all staged
  for (int l = 0; l < p.L; l++) {
    const StackLayer LY = lay_s[l];
    SK_INIT_ACC(0, 0)
    SK_INIT_ACC(1, 32)
    SK_INIT_ACC(2, 64)
    SK_INIT_ACC(3, 96)

    // ---- dilated conv taps (+ conditioning 1x1): chunk ch of the layer.  One code path for both kinds:
    // the conditioning tile has the operand tile's row stride and 64 (zero-padded) columns, its weights
    // are a 128 x 64 chunk like a tap's, so only the B-operand address differs.  (Two code paths that
    // both update the accumulators make the register allocator copy all 64 of them around every chunk.)
    for (int ch = 0; ch < nch; ch++) {
      const bool is_aux = ch >= p.ktaps;
      __syncthreads();  // chunk `cur` committed by everybody; everything before it consumed
      {  // prefetch the following chunk (next tap / conditioning / out|skip pair)
        const long long noff = ch + 1 < p.ktaps ? LY.w_conv + (long long)(ch + 1) * 128 * 64 : (ch + 1 < nch ? LY.w_aux : LY.w_os);
        sk_fetch<PRECISE, NT>(wr, p.whi + noff, p.wlo + noff, 1024, tid);
      }
      const unsigned char* wf_hi = WS_HI(cur) + l31 * XS + half * 16;  // weight fragments: A operand
      // B operand: this wave's frames, shifted by the tap (operand tile) or as they are (conditioning tile)
      const int boff = (is_aux ? p.o_chi + row * CS : (SK_GUARD + row + LY.off0 + ch * LY.dil) * XS) + half * 16;
      const unsigned char* xf_hi = smem + boff;
      const unsigned char* xf_lo = smem + boff + (is_aux ? p.o_clo - p.o_chi : p.o_xlo);
      if constexpr (PRECISE) {
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
          const bf16x8 x_hi = lds_frag(xf_hi + kc * 32);
          const bf16x8 x_lo = lds_frag(xf_lo + kc * 32);
          SK_MMA(wf_hi, kc, x_hi, x_lo)
        }
      } else {
        // software pipeline over the four k-steps: the fragments of two steps are in flight before the
        // first MFMA, the loads of step k+2 are issued between the MFMAs of step k (left to itself the
        // scheduler keeps one or two loads ahead and every other MFMA waits a full LDS round trip)
        bf16x8 xb[2], wa[2][4];
#define SK_LD(buf, kc)                                                               \
  {                                                                                  \
    xb[buf] = lds_frag(xf_hi + (kc) * 32);                                           \
    _Pragma("unroll") for (int nt = 0; nt < 4; nt++) wa[buf][nt] = lds_frag(wf_hi + nt * 32 * XS + (kc) * 32); \
  }
        SK_LD(0, 0)
        SK_LD(1, 1)
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
#pragma unroll
          for (int nt = 0; nt < 4; nt++) acc[nt] = mfma_bf16(wa[kc & 1][nt], xb[kc & 1], acc[nt]);
          if (kc < 2) SK_LD(kc & 1, kc + 2)
        }
#undef SK_LD
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);  // 10 LDS reads
#define SK_SG(nld) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, nld, 0);
        SK_SG(2) SK_SG(1) SK_SG(1) SK_SG(1)                  // step 0: MFMA, then 1-2 loads of step 2
        SK_SG(2) SK_SG(1) SK_SG(1) SK_SG(1)                  // step 1 / loads of step 3
#undef SK_SG
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    // steps 2, 3
      }
      if (PRECISE) __syncthreads();  // single weight buffer: consumed before it is overwritten
      // (fence: the writes go to the OTHER buffer, so the scheduler is free to hoist them - and the wait
      // for the prefetch they consume - above the MFMAs, which would expose the whole L2 latency)
      __builtin_amdgcn_sched_barrier(0);
      SK_COMMIT(WS_HI(PRECISE ? 0 : cur ^ 1))
      if (!PRECISE) cur ^= 1;
    }

    // ---- gate -> saved activations; z stays in registers; 1x1 out|skip; residual update ----
    __syncthreads();  // out|skip chunk committed; all tap reads of the operand tile done
    const bool have_next = l + 1 < p.L;
    if (have_next)
      sk_fetch<PRECISE, NT>(wr, p.whi + lay_s[l + 1].w_conv, p.wlo + lay_s[l + 1].w_conv, 1024, tid);
    bf16x8 zf_hi[4], zf_lo[4];
    {
      const __amdgpu_buffer_rsrc_t r_zh = sk_rsrc16(save_b ? p.zb_hi + (long)l * P : (const uint16_t*)p.skip, P);
      const __amdgpu_buffer_rsrc_t r_zl = sk_rsrc16((save_b && PRECISE) ? p.zb_lo + (long)l * P : (const uint16_t*)p.skip, P);
      const __amdgpu_buffer_rsrc_t r_th = sk_rsrc16(save_b ? p.tb_hi + (long)l * P : (const uint16_t*)p.skip, P);
      const __amdgpu_buffer_rsrc_t r_tl = sk_rsrc16((save_b && PRECISE) ? p.tb_lo + (long)l * P : (const uint16_t*)p.skip, P);
      const __amdgpu_buffer_rsrc_t r_gh = sk_rsrc16(save_b ? p.sg_hi + (long)l * P : (const uint16_t*)p.skip, P);
      const __amdgpu_buffer_rsrc_t r_gl = sk_rsrc16((save_b && PRECISE) ? p.sg_lo + (long)l * P : (const uint16_t*)p.skip, P);
#pragma unroll
      for (int kc = 0; kc < 4; kc++) {  // 16 channels: quads g0 and g0+1 of tile h2
        const int h2 = kc >> 1, g0 = (kc & 1) * 2;
        sk_u32x2 zq_hi[2], zq_lo[2], tq_hi[2], tq_lo[2], sq_hi[2], sq_lo[2];
#pragma unroll
        for (int gg = 0; gg < 2; gg++) {
          const int g = g0 + gg;
          float ta[4], sb[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            ta[j] = sk_tanh(acc[h2][4 * g + j], PRECISE);
            sb[j] = sk_sigmoid(acc[h2 + 2][4 * g + j], PRECISE);
          }
          sk_quad<PRECISE>(ta[0], ta[1], ta[2], ta[3], tq_hi[gg], tq_lo[gg]);
          sk_quad<PRECISE>(sb[0], sb[1], sb[2], sb[3], sq_hi[gg], sq_lo[gg]);
          sk_quad<PRECISE>(ta[0] * sb[0], ta[1] * sb[1], ta[2] * sb[2], ta[3] * sb[3], zq_hi[gg], zq_lo[gg]);
        }
        // tanh / sigmoid / z as 8-channel bf16 fragments: the gate backward and the weight gradient read them
        zf_hi[kc] = sk_swap_frag(zq_hi[0], zq_hi[1]);
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(tq_hi[0], tq_hi[1])), r_th, voff_b + (kc * 32), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(sq_hi[0], sq_hi[1])), r_gh, voff_b + (kc * 32), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(zf_hi[kc]), r_zh, voff_b + (kc * 32), 0, 0);
        if (PRECISE) {
          zf_lo[kc] = sk_swap_frag(zq_lo[0], zq_lo[1]);
          __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(tq_lo[0], tq_lo[1])), r_tl, voff_b + (kc * 32), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(sq_lo[0], sq_lo[1])), r_gl, voff_b + (kc * 32), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(zf_lo[kc]), r_zl, voff_b + (kc * 32), 0, 0);
        }
      }
    }
    // out | skip 1x1: the MFMA chains accumulate ON the residual stream / the skip sum (same expression, same order as
    // stack2_fwd_kernel): x <- fma(x + out, sqrt(.5), b_out sqrt(.5)), s <- (s + skip) + b_skip
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) { acc[h2] = res[h2]; acc[h2 + 2] = skp[h2]; }
    {
      const unsigned char* wf_hi = WS_HI(cur) + l31 * XS + half * 16;
#pragma unroll
      for (int kc = 0; kc < 4; kc++) SK_MMA(wf_hi, kc, zf_hi[kc], zf_lo[kc])
    }
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const sk_f32x4 bo = *reinterpret_cast<const sk_f32x4*>(bias_s + l * 256 + 128 + 32 * h2 + 8 * g + 4 * half);
        const sk_f32x4 bs = *reinterpret_cast<const sk_f32x4*>(bias_s + l * 256 + 192 + 32 * h2 + 8 * g + 4 * half);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float o = __builtin_fmaf(acc[h2][4 * g + j], rs, bo[j]);
          res[h2][4 * g + j] = rin ? o : 0.f;
          skp[h2][4 * g + j] = __builtin_fmaf(acc[h2 + 2][4 * g + j], 1.f, bs[j]);
        }
      }
    if (have_next) SK_PUT_OPERAND(l + 1)  // everybody is past this layer's tap reads (barrier above)
    if (PRECISE) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (have_next) SK_COMMIT(WS_HI(PRECISE ? 0 : cur ^ 1))
    if (!PRECISE) cur ^= 1;
  }

  // ---- running skip sum of the window's own frames ----
  const __amdgpu_buffer_rsrc_t r_sk = sk_rsrc(p.skip, P);
#pragma unroll
  for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      sk_u32x4 q;
#pragma unroll
      for (int j = 0; j < 4; j++) q[j] = sk_f2u(skp[h2][4 * g + j]);
      __builtin_amdgcn_raw_buffer_store_b128(q, r_sk, voff_out + (SK_QOFF(h2, g)), 0, 0);
    }
}

int stack_fwd_plan(StackP& p, bool precise) {
  static int nw_env = -1;
  if (nw_env < 0) nw_env = crk_sw().sk_nw_fwd;
  const int XS = SK_XS;
  p.nw = precise ? 4 : (nw_env == 4 || nw_env == 6 || nw_env == 8 ? nw_env : 8);
  if (p.nw == 8 && 256 - p.hl - p.hr < 32) return CRK_ERR_UNSUPPORTED;
  const int R = p.nw * 32;
  p.tmo = R - p.hl - p.hr;
  if (p.tmo < 32 || p.max_off > SK_GUARD || p.ktaps > 8) return CRK_ERR_UNSUPPORTED;
  if (p.aux_ch > 0 && p.aux_pad != 64) return CRK_ERR_UNSUPPORTED;  // the conditioning chunk is consumed as 4 k-steps of 16
  // balance the windows of an utterance: same count, equal share of frames
  p.tiles_per_utt = ceil_div(p.T, p.tmo);
  p.tmo = ceil_div(p.T, p.tiles_per_utt);
  const int xbytes = (SK_GUARD * 2 + R) * XS;
  const int cbytes = p.aux_ch > 0 ? R * (p.aux_pad * 2 + 16) : 0;
  p.w_bytes = 128 * XS;
  int off = xbytes;
  p.o_xlo = off; if (precise) off += xbytes;
  p.o_chi = off; off += cbytes;
  p.o_clo = off; if (precise) off += cbytes;
  p.o_whi = off; off += precise ? p.w_bytes : 2 * p.w_bytes;
  p.o_wlo = off; if (precise) off += p.w_bytes;
  p.o_bias = off; off += p.L * 256 * 4;
  p.o_tab = off; off += p.L * (int)sizeof(StackLayer);
  p.lds_bytes = (off + 15) & ~15;
  return p.lds_bytes <= 160 * 1024 ? CRK_OK : CRK_ERR_UNSUPPORTED;
}

int launch_stack_fwd(const StackP& p, bool precise, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[8] = {(const void*)stack_fwd_kernel<true, true, 4>,  (const void*)stack_fwd_kernel<true, false, 4>,
                          (const void*)stack_fwd_kernel<false, true, 4>, (const void*)stack_fwd_kernel<false, false, 4>,
                          (const void*)stack_fwd_kernel<false, true, 6>, (const void*)stack_fwd_kernel<false, false, 6>,
                          (const void*)stack_fwd_kernel<false, true, 8>, (const void*)stack_fwd_kernel<false, false, 8>};
    for (int i = 0; i < 8; i++)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return CRK_ERR_HIP;
    attr_set = true;
  }
  dim3 grid(p.B * p.tiles_per_utt);
  const double nfr = (double)p.B * p.T;
  // algorithmic bytes: block-0 input and conditioning read, skip sum written; saving launches add 4 bf16 planes per block
  conv_prof_bytes(1, nfr * (256.0 + 4.0 * p.aux_ch + 256.0 + (p.xb_hi ? 512.0 * p.L + 2.0 * (p.aux_ch > 0 ? p.aux_pad : 0) : 0.0)));
  conv_prof_begin(1, 2.0 * nfr * p.L * (128.0 * (64.0 * p.ktaps + p.aux_ch) + 128.0 * 64.0), s);
  const bool drop = p.drop_p > 0.f;
#define SK_LAUNCH(PR, DR, NWV) hipLaunchKernelGGL((stack_fwd_kernel<PR, DR, NWV>), grid, dim3(NWV * 64), p.lds_bytes, s, p)
  if (precise) { if (drop) SK_LAUNCH(true, true, 4); else SK_LAUNCH(true, false, 4); }
  else if (p.nw == 4) { if (drop) SK_LAUNCH(false, true, 4); else SK_LAUNCH(false, false, 4); }
  else if (p.nw == 6) { if (drop) SK_LAUNCH(false, true, 6); else SK_LAUNCH(false, false, 6); }
  else { if (drop) SK_LAUNCH(false, true, 8); else SK_LAUNCH(false, false, 8); }
#undef SK_LAUNCH
  conv_prof_end(1, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// =====================================================================================
// Fused DATA-GRADIENT chain of all gated residual blocks of a stack, last block first.
//
// Per block l (parallel_wavegan ResidualBlock backward; the forward is restated above):
//   dz   = sqrt(.5) * Wout^T dX_{l+1} + Wskip^T dS           1x1, K = 128 -> 64
//   dG_l = [dz * sb * (1 - ta^2) | dz * ta * sb * (1 - sb)]   gate backward, 128 channels
//   dX_l = sqrt(.5) * dX_{l+1} + mask_l * convT(dG_l)         dilated taps, K = 128 -> 64
//   dc  += Waux^T dG_l                                        conditioning gradient
// dS (gradient wrt the skip sum) is the same for every block; dX_L = 0 (the last block's
// residual output is unused).  mask_l is the regenerated dropout keep mask of the block input;
// for the discriminator dX_0 is additionally multiplied by LeakyReLU'(X_0).
//
// Same decomposition as the forward kernel: a window of R = 32*NW frames per workgroup with
// the stack's (mirrored) halo recomputed, every lane owns one frame, dX_{l+1} lives in fp32
// accumulator-layout registers across blocks and reaches the B operand of the 1x1 through
// v_permlane32_swap; dS fragments are built once.  Only dG passes through LDS (the taps need
// it at shifted frames).  HBM traffic per block: ta, sb read; dG_l, dX_l written (the weight
// gradient consumes them afterwards) - nothing else.
// Phase cycles (tools/skb_phase_cycles.py, -DSKB_PROF): per workgroup and wave [0] prologue (folded head) [1] waiting at the
// chunk barriers [2] out|skip 1x1 [3] gate backward [4] taps (+ conditioning 1x1) [5] dX epilogue [6] weight commit [7] kernel
#ifdef SKB_PROF
__device__ unsigned long long skb_prof_buf[256 * 8 * 12];
extern "C" int crk_debug_skb_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(skb_prof_buf), sizeof(unsigned long long) * 256 * 8 * 12) == hipSuccess ? 0 : 2;
}
#define SKB_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pacc_[i] += t_ - plast_; plast_ = t_; }
#else
#define SKB_T(i)
#endif
#define SKB_GS 272  // row stride of the dG tile and of a [64][128] weight chunk: 128 bf16 + 16 B pad

__device__ __forceinline__ bf16x8 skb_frag8(const sk_u32x4 a, const sk_u32x4 b, bool lo_plane) {
  float v[8] = {sk_u2f(a[0]), sk_u2f(a[1]), sk_u2f(a[2]), sk_u2f(a[3]), sk_u2f(b[0]), sk_u2f(b[1]), sk_u2f(b[2]), sk_u2f(b[3])};
  if (lo_plane) {
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = sk_bf_lo(v[j]);
  }
  const sk_u32x4 r = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, r);
}

template <bool PRECISE, bool DROP, int NW, bool FOLD = false>
__global__ __launch_bounds__(NW * 64, PRECISE ? 1 : 2) void stack_bwd_kernel(const StackBP p) {
  constexpr int NT = NW * 64, R = NW * 32, GS = SKB_GS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
#ifdef SKB_PROF
  unsigned long long pacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pstart_ = __builtin_readcyclecounter();
  unsigned long long plast_ = pstart_;
#endif
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;
  const long nbase = (long)b * p.T;

  unsigned char* gs_hi = smem;  // [SK_GUARD + R + SK_GUARD][GS]
  unsigned char* gs_lo = smem + p.o_glo;
  // weight buffer i as an OFFSET from the LDS base: a runtime-indexed array of pointers would decay to
  // generic pointers and every weight-fragment access to a FLAT instruction (vmcnt + lgkmcnt, i.e.
  // serialised with the global weight prefetch) instead of ds_read / ds_write
#define WS_HI(i) (smem + p.o_whi + ((PRECISE || (i) == 0) ? 0 : p.w_bytes))
  unsigned char* ws_lo = smem + p.o_wlo;

  const int row = wave * 32 + l31;
  const int t = t0 - p.hl + row;
  const bool rin = t >= 0 && t < p.T;
  const bool rout = rin && row >= p.hl && row < p.hl + p.tmo;
  const int voff_in = rin ? (int)(((nbase + t) * 64 + 4 * half) * 4) : SK_OOB;   // [N,64] planes
  const int voff_out = rout ? voff_in : SK_OOB;
  const int voff_b = rout ? (int)(((nbase + t) * 64 + 8 * half) * 2) : SK_OOB;    // bf16 [N,64] planes, 8-channel fragments
  const int voff_gb = rout ? (int)(((nbase + t) * 128 + 8 * half) * 2) : SK_OOB;  // bf16 [N,128] dG planes
  const long P = (long)p.B * p.T * 64;

  // weight chunk [64 rows][128 k] bf16 = 1024 16-byte pieces, 16 per row
  int wdst[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int idx = tid + u * NT;
    wdst[u] = idx < 1024 ? (idx >> 4) * GS + (idx & 15) * 16 : -1;  // NT = 384: half of the last round is unused
  }
#define SKB_C1(u, dhi)                                                              \
  if (u * NT < 1024 && (1024 % NT == 0 || wdst[u] >= 0)) {                          \
    *reinterpret_cast<sk_u32x4*>((dhi) + wdst[u]) = wr.h##u;                        \
    if (PRECISE) *reinterpret_cast<sk_u32x4*>(ws_lo + wdst[u]) = wr.l##u;           \
  }
#define SKB_COMMIT(dhi) { SKB_C1(0, dhi) SKB_C1(1, dhi) SKB_C1(2, dhi) SKB_C1(3, dhi) }

  const bool has_aux = p.dc != nullptr && p.aux_ch > 0;
  const int nq = 1 + p.ktaps + (has_aux ? 1 : 0);  // chunks per block: out|skip 1x1, taps, aux

  SkRegs wr;
  {
    const StackBLayer L0 = p.layers[p.L - 1];
    sk_fetch<PRECISE, NT>(wr, p.whi + L0.w_os, p.wlo + L0.w_os, 1024, tid);
  }
  for (int i = tid; i < SK_GUARD * GS / 16; i += NT) {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(gs_hi)[i] = z4;
    reinterpret_cast<uint4*>(gs_hi + (SK_GUARD + R) * GS)[i] = z4;
    if (PRECISE) {
      reinterpret_cast<uint4*>(gs_lo)[i] = z4;
      reinterpret_cast<uint4*>(gs_lo + (SK_GUARD + R) * GS)[i] = z4;
    }
  }
  // dS fragments (B operand, k = 64..127 of the 1x1): 8 consecutive channels per lane and 16-group
  bf16x8 dsf_hi[4], dsf_lo[4];
  f32x16 dxo[2], accc[2], acc[2];
  if constexpr (FOLD) {
    // ---- the head's data gradient first: dy -> (W2^T, x relu'(H1)) -> (W1^T, x relu'(S), x sqrt(1/L)) = dS, kept in
    // registers as the fragments the chain consumes; bf16 dy and the middle gradient are the planes the head's weight
    // gradients read.  Both transposed weight planes are requested up front and pass through weight buffer 1. ----
    const long N = (long)p.B * p.T;
    const int KY = p.kp_y >> 4, ppr2 = p.kp_y >> 3;
    sk_u32x4 w2r[2], w1r;
    {
      const int tot2 = 64 * ppr2;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int idx = tid + u * NT;
        w2r[u] = *reinterpret_cast<const sk_u32x4*>(p.whi + p.w_h2 + (idx < tot2 ? (long)idx * 8 : 0));
      }
      w1r = *reinterpret_cast<const sk_u32x4*>(p.whi + p.w_h1 + (long)tid * 8);
    }
    const __amdgpu_buffer_rsrc_t rdy = sk_rsrc(p.dy, N * p.lddy);
    sk_u32x4 ya[8], yc[8];
#pragma unroll
    for (int kc = 0; kc < 8; kc++)
      if (kc < KY) {
        const int c0 = 16 * kc + 8 * half;
        const int vo = (rin && c0 < p.out_ch) ? (int)(((nbase + t) * p.lddy + c0) * 4) : SK_OOB;
        ya[kc] = __builtin_amdgcn_raw_buffer_load_b128(rdy, vo, 0, 0);
        yc[kc] = __builtin_amdgcn_raw_buffer_load_b128(rdy, vo + 16, 0, 0);
      }
    // both relu' masks of the head (planes H1 | S) are requested with dy: where they are used each would be an exposed
    // round trip between two short MFMA runs
    sk_u32x2 pm1[8], pm0[8];
#define SKB_HEAD_MASK(dst, plane)                                                                     \
    {                                                                                                 \
      const __amdgpu_buffer_rsrc_t r_hm = sk_rsrc16(p.hmask_hi + (plane), P);                         \
      _Pragma("unroll") for (int i = 0; i < 8; i++) { /* i = h2 * 4 + g */                            \
        const int c0 = (i >> 2) * 32 + 8 * (i & 3) + 4 * half;                                        \
        dst[i] = __builtin_amdgcn_raw_buffer_load_b64(r_hm, rin ? (int)(((nbase + t) * 64 + c0) * 2) : SK_OOB, 0, 0); \
      }                                                                                               \
    }
    SKB_HEAD_MASK(pm1, P)
    bf16x8 yg[8];
    {
      const __amdgpu_buffer_rsrc_t r_g2 = sk_rsrc16(p.hb_hi, N * p.kp_y);
#pragma unroll
      for (int kc = 0; kc < 8; kc++)
        if (kc < KY) {
          const sk_u32x4 fb = {pack_bf2(sk_u2f(ya[kc][0]), sk_u2f(ya[kc][1])), pack_bf2(sk_u2f(ya[kc][2]), sk_u2f(ya[kc][3])),
                               pack_bf2(sk_u2f(yc[kc][0]), sk_u2f(yc[kc][1])), pack_bf2(sk_u2f(yc[kc][2]), sk_u2f(yc[kc][3]))};
          yg[kc] = __builtin_bit_cast(bf16x8, fb);
          __builtin_amdgcn_raw_buffer_store_b128(fb, r_g2, rout ? (int)(((nbase + t) * p.kp_y + 16 * kc + 8 * half) * 2) : SK_OOB, 0, 0);
        }
    }
    SKB_HEAD_MASK(pm0, 0)  // (after the dy registers are free)
#undef SKB_HEAD_MASK
    unsigned char* wb1 = smem + p.o_whi + p.w_bytes;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int idx = tid + u * NT;
      if (idx < 64 * ppr2) *reinterpret_cast<sk_u32x4*>(wb1 + (idx / ppr2) * GS + (idx % ppr2) * 16) = w2r[u];
    }
    __syncthreads();
    const unsigned char* wfh = wb1 + l31 * GS + half * 16;
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[h2][i] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 8; kc++)
      if (kc < KY) {
#pragma unroll
        for (int nt = 0; nt < 2; nt++) acc[nt] = mfma_bf16(lds_frag(wfh + nt * 32 * GS + kc * 32), yg[kc], acc[nt]);
      }
    // x relu'(H1) -> G1: fragments for the next 1x1 and the plane for the weight gradient of the head's first conv
    bf16x8 g1[4];
    {
      const __amdgpu_buffer_rsrc_t r_g1 = sk_rsrc16(p.hb_hi + N * p.kp_y, P);
#pragma unroll
      for (int kc = 0; kc < 4; kc++) {
        const int h2 = kc >> 1, g0 = (kc & 1) * 2;
        sk_u32x2 qh[2], ql[2];
#pragma unroll
        for (int gg = 0; gg < 2; gg++) {
          const int g = g0 + gg;
          const sk_u32x2 m = pm1[h2 * 4 + g];
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const unsigned w = j < 2 ? m[0] : m[1];
            const float mv = sk_u2f((j & 1) ? (w & 0xffff0000u) : (w << 16));
            v[j] = rin ? acc[h2][4 * g + j] * (mv > 0.f ? 1.f : 0.f) : 0.f;
          }
          sk_quad<false>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);
        }
        g1[kc] = sk_swap_frag(qh[0], qh[1]);
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(g1[kc]), r_g1, voff_b + (kc * 32), 0, 0);
      }
    }
    __syncthreads();  // every read of W2^T done
    if (tid < 512) *reinterpret_cast<sk_u32x4*>(wb1 + (tid >> 3) * GS + (tid & 7) * 16) = w1r;
    __syncthreads();
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[h2][i] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; kc++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) acc[nt] = mfma_bf16(lds_frag(wfh + nt * 32 * GS + kc * 32), g1[kc], acc[nt]);
    {
      const __amdgpu_buffer_rsrc_t r_sh = sk_rsrc16(p.dsb_hi, P);
#pragma unroll
      for (int kc = 0; kc < 4; kc++) {
        const int h2 = kc >> 1, g0 = (kc & 1) * 2;
        sk_u32x2 qh[2], ql[2];
#pragma unroll
        for (int gg = 0; gg < 2; gg++) {
          const int g = g0 + gg;
          const sk_u32x2 m = pm0[h2 * 4 + g];
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const unsigned w = j < 2 ? m[0] : m[1];
            const float mv = sk_u2f((j & 1) ? (w & 0xffff0000u) : (w << 16));
            v[j] = rin ? acc[h2][4 * g + j] * (mv > 0.f ? 1.f : 0.f) * p.head_scale : 0.f;
          }
          sk_quad<false>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);
        }
        dsf_hi[kc] = sk_swap_frag(qh[0], qh[1]);
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(dsf_hi[kc]), r_sh, voff_b + (kc * 32), 0, 0);
      }
    }
  } else
  {
    const __amdgpu_buffer_rsrc_t rds = sk_rsrc(p.dS, P);
    const int voff_s = rin ? (int)(((nbase + t) * 64 + 8 * half) * 4) : SK_OOB;
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
      const sk_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rds, voff_s + (kc * 64), 0, 0);
      const sk_u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rds, voff_s + (kc * 64 + 16), 0, 0);
      dsf_hi[kc] = skb_frag8(a, c, false);
      if (PRECISE) dsf_lo[kc] = skb_frag8(a, c, true);
    }
    const __amdgpu_buffer_rsrc_t r_sh = sk_rsrc16(p.dsb_hi, P);
    const __amdgpu_buffer_rsrc_t r_sl = sk_rsrc16(PRECISE ? p.dsb_lo : p.dsb_hi, P);
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
      __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(dsf_hi[kc]), r_sh, voff_b + (kc * 32), 0, 0);
      if (PRECISE) __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(dsf_lo[kc]), r_sl, voff_b + (kc * 32), 0, 0);
    }
  }
#pragma unroll
  for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
    for (int i = 0; i < 16; i++) { dxo[h2][i] = 0.f; accc[h2][i] = 0.f; }
  SKB_COMMIT(WS_HI(0))
  SKB_T(0)

  const float rs = 0.70710678118654752440f;
  int cur = 0;
  const unsigned char* wf_lo = ws_lo + l31 * GS + half * 16;
  unsigned char* my_gs_hi = gs_hi + (SK_GUARD + row) * GS + 8 * half * 2;
  unsigned char* my_gs_lo = gs_lo + (SK_GUARD + row) * GS + 8 * half * 2;

#define SKB_MMA(dst, wf_hi, kc, x_hi, x_lo)                                        \
  _Pragma("unroll") for (int nt = 0; nt < 2; nt++) {                               \
    const bf16x8 w_hi = lds_frag((wf_hi) + nt * 32 * GS + (kc) * 32);              \
    dst[nt] = mfma_bf16(w_hi, x_hi, dst[nt]);                                      \
    if (PRECISE) {                                                                 \
      const bf16x8 w_lo = lds_frag(wf_lo + nt * 32 * GS + (kc) * 32);              \
      dst[nt] = mfma_bf16(w_hi, x_lo, dst[nt]);                                    \
      dst[nt] = mfma_bf16(w_lo, x_hi, dst[nt]);                                    \
    }                                                                              \
  }

  // A block's chunks in order: out|skip 1x1 (+ gate backward), the taps, the conditioning 1x1.  Every stage is
  // straight-line code of its own (the last tap peeled off the tap loop): with one loop over chunk kinds and
  // a branch per kind the register allocator copied whole accumulator sets around every chunk.
#define SKB_OPEN(has, off_expr)                                                                       \
  __syncthreads(); /* chunk `cur` committed; every read of the previous chunk's operands done */      \
  SKB_T(1)                                                                                            \
  have_next = (has);                                                                                  \
  if (have_next) {                                                                                    \
    const long long off_ = (off_expr);                                                                \
    sk_fetch<PRECISE, NT>(wr, p.whi + off_, p.wlo + off_, 1024, tid);                                 \
  }                                                                                                   \
  wf_hi = WS_HI(cur) + l31 * GS + half * 16;
#define SKB_CLOSE                                                                                     \
  if (PRECISE) __syncthreads();                                                                       \
  __builtin_amdgcn_sched_barrier(0); /* keep the commit (and its wait for the prefetch) behind the MFMAs */ \
  if (have_next) SKB_COMMIT(WS_HI(PRECISE ? 0 : cur ^ 1))                                             \
  if (!PRECISE) cur ^= 1;                                                                             \
  SKB_T(6)

  StackBLayer LYn = p.layers[p.L - 1];
  for (int l = p.L - 1; l >= 0; l--) {
    const StackBLayer LY = LYn;
    if (l > 0) LYn = p.layers[l - 1];  // the next block's table entry: a scalar load a whole block ahead of its first use
    bool have_next;
    const unsigned char* wf_hi;
    const long long after_taps = has_aux ? LY.w_aux : (l > 0 ? LYn.w_os : 0);
    {
      SKB_OPEN(true, LY.w_conv)  // next: tap 0
      // the block's tanh / sigmoid planes (gate backward, below) are requested before the 1x1's MFMAs: asked for where
      // they are used, each block waited a full HBM / L2 round trip for them with nothing else to do
      const __amdgpu_buffer_rsrc_t r_th = sk_rsrc16(p.tb_hi + (long)l * P, P);
      const __amdgpu_buffer_rsrc_t r_sh = sk_rsrc16(p.sg_hi + (long)l * P, P);
      const int voff_bi = rin ? (int)(((nbase + t) * 64 + 8 * half) * 2) : SK_OOB;  // every in-utterance row of the window
      sk_u32x4 pft[4], pfs[4];
#pragma unroll
      for (int kc = 0; kc < 4; kc++) {
        pft[kc] = __builtin_amdgcn_raw_buffer_load_b128(r_th, voff_bi + (kc * 32), 0, 0);
        pfs[kc] = __builtin_amdgcn_raw_buffer_load_b128(r_sh, voff_bi + (kc * 32), 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler sinks the loads back to their uses)
      SKB_T(8)
      // ---- dz = [sqrt(.5) dX_{l+1} | dS] . [Wout ; Wskip]^T ----
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[h2][i] = 0.f;
      if constexpr (PRECISE) {
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
          const int h2 = kc >> 1, g0 = (kc & 1) * 2;
          sk_u32x2 qh[2], ql[2];
#pragma unroll
          for (int gg = 0; gg < 2; gg++) {
            const int i0 = 4 * (g0 + gg);
            sk_quad<PRECISE>(dxo[h2][i0] * rs, dxo[h2][i0 + 1] * rs, dxo[h2][i0 + 2] * rs, dxo[h2][i0 + 3] * rs, qh[gg], ql[gg]);
          }
          const bf16x8 x_hi = sk_swap_frag(qh[0], qh[1]);
          const bf16x8 x_lo = sk_swap_frag(ql[0], ql[1]);
          SKB_MMA(acc, wf_hi, kc, x_hi, x_lo)
        }
#pragma unroll
        for (int kc = 0; kc < 4; kc++) SKB_MMA(acc, wf_hi, 4 + kc, dsf_hi[kc], dsf_lo[kc])
      } else {
        // fast mode: the dX operand fragments are converted first, then the eight k-steps run as a software pipeline with
        // the weight fragments of three steps in flight (dS half first: its operands are ready-made) - asked for one
        // MFMA at a time, every MFMA of this chunk waited a full LDS round trip
        bf16x8 xq[4];
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
          const int h2 = kc >> 1, g0 = (kc & 1) * 2;
          sk_u32x2 qh[2], ql[2];
#pragma unroll
          for (int gg = 0; gg < 2; gg++) {
            const int i0 = 4 * (g0 + gg);
            sk_quad<false>(dxo[h2][i0] * rs, dxo[h2][i0 + 1] * rs, dxo[h2][i0 + 2] * rs, dxo[h2][i0 + 3] * rs, qh[gg], ql[gg]);
          }
          xq[kc] = sk_swap_frag(qh[0], qh[1]);
        }
        SKB_T(9)
        bf16x8 wq[3][2];
#define SKB_W1(buf, st) { const int kc_ = (st) < 4 ? 4 + (st) : (st) - 4; wq[buf][0] = lds_frag(wf_hi + kc_ * 32); wq[buf][1] = lds_frag(wf_hi + 32 * GS + kc_ * 32); }
        SKB_W1(0, 0) SKB_W1(1, 1) SKB_W1(2, 2)
#pragma unroll
        for (int st = 0; st < 8; st++) {
          const bf16x8 xo = st < 4 ? dsf_hi[st] : xq[st - 4];
          acc[0] = mfma_bf16(wq[st % 3][0], xo, acc[0]);
          acc[1] = mfma_bf16(wq[st % 3][1], xo, acc[1]);
          if (st + 3 < 8) SKB_W1(st % 3, st + 3)
        }
#undef SKB_W1
      }
      SKB_T(2)
      // ---- gate backward -> dG_l (HBM for the weight gradient, LDS for the taps) ----
      const __amdgpu_buffer_rsrc_t r_tl = sk_rsrc16((PRECISE ? p.tb_lo : p.tb_hi) + (long)l * P, P);
      const __amdgpu_buffer_rsrc_t r_sl = sk_rsrc16((PRECISE ? p.sg_lo : p.sg_hi) + (long)l * P, P);
      const __amdgpu_buffer_rsrc_t r_gh = sk_rsrc16(p.gb_hi + (long)l * 2 * P, 2 * P);
      const __amdgpu_buffer_rsrc_t r_gl = sk_rsrc16((PRECISE ? p.gb_lo : p.gb_hi) + (long)l * 2 * P, 2 * P);
#pragma unroll
      for (int kc = 0; kc < 4; kc++) {  // 16 channels of each gate half: quads g0, g0+1 of tile h2
        const int h2 = kc >> 1, g0 = (kc & 1) * 2;
        // tanh / sigmoid come as 8-channel fragments; the lane-pair exchange (its own inverse)
        // returns them to this lane's two accumulator-layout quads
        float tav[8], sbv[8];
        {
          const sk_u32x4 ft = pft[kc], fs = pfs[kc];
          const sk_u32x2 t0 = __builtin_amdgcn_permlane32_swap(ft[0], ft[2], false, false);
          const sk_u32x2 t1 = __builtin_amdgcn_permlane32_swap(ft[1], ft[3], false, false);
          const sk_u32x2 s0 = __builtin_amdgcn_permlane32_swap(fs[0], fs[2], false, false);
          const sk_u32x2 s1 = __builtin_amdgcn_permlane32_swap(fs[1], fs[3], false, false);
          const unsigned tw[4] = {t0[0], t1[0], t0[1], t1[1]}, sw[4] = {s0[0], s1[0], s0[1], s1[1]};  // quad g0 | quad g0+1
#pragma unroll
          for (int j = 0; j < 4; j++) {
            tav[2 * j] = sk_u2f(tw[j] << 16); tav[2 * j + 1] = sk_u2f(tw[j] & 0xffff0000u);
            sbv[2 * j] = sk_u2f(sw[j] << 16); sbv[2 * j + 1] = sk_u2f(sw[j] & 0xffff0000u);
          }
          if (PRECISE) {
            const sk_u32x4 lt = __builtin_amdgcn_raw_buffer_load_b128(r_tl, voff_bi + (kc * 32), 0, 0);
            const sk_u32x4 ls = __builtin_amdgcn_raw_buffer_load_b128(r_sl, voff_bi + (kc * 32), 0, 0);
            const sk_u32x2 a0 = __builtin_amdgcn_permlane32_swap(lt[0], lt[2], false, false);
            const sk_u32x2 a1 = __builtin_amdgcn_permlane32_swap(lt[1], lt[3], false, false);
            const sk_u32x2 b0 = __builtin_amdgcn_permlane32_swap(ls[0], ls[2], false, false);
            const sk_u32x2 b1 = __builtin_amdgcn_permlane32_swap(ls[1], ls[3], false, false);
            const unsigned tl[4] = {a0[0], a1[0], a0[1], a1[1]}, sl[4] = {b0[0], b1[0], b0[1], b1[1]};
#pragma unroll
            for (int j = 0; j < 4; j++) {
              tav[2 * j] += sk_u2f(tl[j] << 16); tav[2 * j + 1] += sk_u2f(tl[j] & 0xffff0000u);
              sbv[2 * j] += sk_u2f(sl[j] << 16); sbv[2 * j + 1] += sk_u2f(sl[j] & 0xffff0000u);
            }
          }
        }
        sk_u32x2 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int gg = 0; gg < 2; gg++) {
          const int g = g0 + gg;
          float da[4], db[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float ta = tav[4 * gg + j], sb = sbv[4 * gg + j], dz = acc[h2][4 * g + j];
            sk_gate_bwd(dz, ta, sb, da[j], db[j]);
          }
          sk_quad<PRECISE>(da[0], da[1], da[2], da[3], ah[gg], al[gg]);
          sk_quad<PRECISE>(db[0], db[1], db[2], db[3], bh[gg], bl[gg]);
        }
        // 8-channel fragments: LDS tile for the taps, bf16 plane for the weight gradient
        const sk_u32x4 fa = sk_frag_bits(sk_swap_frag(ah[0], ah[1]));
        const sk_u32x4 fb = sk_frag_bits(sk_swap_frag(bh[0], bh[1]));
        *reinterpret_cast<sk_u32x4*>(my_gs_hi + kc * 32) = fa;
        *reinterpret_cast<sk_u32x4*>(my_gs_hi + 128 + kc * 32) = fb;
        __builtin_amdgcn_raw_buffer_store_b128(fa, r_gh, voff_gb + (kc * 32), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(fb, r_gh, voff_gb + (128 + kc * 32), 0, 0);
        if (PRECISE) {
          const sk_u32x4 la = sk_frag_bits(sk_swap_frag(al[0], al[1]));
          const sk_u32x4 lb = sk_frag_bits(sk_swap_frag(bl[0], bl[1]));
          *reinterpret_cast<sk_u32x4*>(my_gs_lo + kc * 32) = la;
          *reinterpret_cast<sk_u32x4*>(my_gs_lo + 128 + kc * 32) = lb;
          __builtin_amdgcn_raw_buffer_store_b128(la, r_gl, voff_gb + (kc * 32), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(lb, r_gl, voff_gb + (128 + kc * 32), 0, 0);
        }
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[h2][i] = 0.f;
      SKB_T(3)
      SKB_CLOSE
    }
// fast mode: the eight k-steps of a 64 x 128 chunk as a software pipeline, fragments of three steps in
// flight (left to itself the scheduler keeps one or two loads ahead of each MFMA)
#define SKB_LD3(buf, kc, gf)                                                                          \
  {                                                                                                   \
    xb[buf] = lds_frag((gf) + (kc) * 32);                                                             \
    wa[buf][0] = lds_frag(wf_hi + (kc) * 32);                                                         \
    wa[buf][1] = lds_frag(wf_hi + 32 * GS + (kc) * 32);                                               \
  }
#define SKB_SG(nld) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, nld, 0);
#define SKB_PIPE(dst, gf)                                                                             \
  {                                                                                                   \
    bf16x8 xb[3], wa[3][2];                                                                           \
    SKB_LD3(0, 0, gf) SKB_LD3(1, 1, gf) SKB_LD3(2, 2, gf)                                             \
    _Pragma("unroll") for (int kc = 0; kc < 8; kc++) {                                                \
      dst[0] = mfma_bf16(wa[kc % 3][0], xb[kc % 3], dst[0]);                                          \
      dst[1] = mfma_bf16(wa[kc % 3][1], xb[kc % 3], dst[1]);                                          \
      if (kc + 3 < 8) SKB_LD3(kc % 3, kc + 3, gf)                                                     \
    }                                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);                                                \
    SKB_SG(2) SKB_SG(1) SKB_SG(2) SKB_SG(1) SKB_SG(2) SKB_SG(1) SKB_SG(2) SKB_SG(1) SKB_SG(2) SKB_SG(1) \
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                                \
  }
#define SKB_TAP(tp)                                                                                   \
  {                                                                                                   \
    /* one tap of the transposed dilated conv */                                                      \
    const int arow = SK_GUARD + row + LY.off0 + (tp) * LY.dil;                                        \
    const unsigned char* gf_hi = gs_hi + arow * GS + half * 16;                                       \
    const unsigned char* gf_lo = gs_lo + arow * GS + half * 16;                                       \
    if constexpr (PRECISE) {                                                                          \
      _Pragma("unroll") for (int kc = 0; kc < 8; kc++) {                                              \
        const bf16x8 x_hi = lds_frag(gf_hi + kc * 32);                                                \
        const bf16x8 x_lo = lds_frag(gf_lo + kc * 32);                                                \
        SKB_MMA(acc, wf_hi, kc, x_hi, x_lo)                                                           \
      }                                                                                               \
    } else {                                                                                          \
      SKB_PIPE(acc, gf_hi)                                                                            \
    }                                                                                                 \
  }
    for (int tp = 0; tp + 1 < p.ktaps; tp++) {
      SKB_OPEN(true, LY.w_conv + (long long)(tp + 1) * 64 * 128)
      SKB_TAP(tp)
      SKB_T(4)
      SKB_CLOSE
    }
    {
      SKB_OPEN(has_aux || l > 0, after_taps)
      SKB_TAP(p.ktaps - 1)
      SKB_T(4)
      // ---- dX_l = sqrt(.5) dX_{l+1} + mask * convT(dG_l); kept in registers for block l-1 ----
      const __amdgpu_buffer_rsrc_t r_x = sk_rsrc(p.dX0, P);
      const int voff_x0 = (l == 0 && !FOLD) ? voff_out : SK_OOB;  // fp32 only for the stack input (folded: consumed below)
      const bool lmask = l == 0 && p.mask_l0;
      const __amdgpu_buffer_rsrc_t r_x0 = sk_rsrc(p.saved, P);
      const __amdgpu_buffer_rsrc_t r_dh = sk_rsrc16(p.dxb_hi + (long)l * P, P);
      const __amdgpu_buffer_rsrc_t r_dl = sk_rsrc16((PRECISE ? p.dxb_lo : p.dxb_hi) + (long)l * P, P);
      const unsigned long long dseed = (DROP ? crk_seed(p.drop_seed, p.drop_seed_ptr) : 0ull) + 0x9E3779B97F4A7C15ull * (unsigned long long)(l + 1);
#pragma unroll
      for (int kc = 0; kc < 4; kc++) {
        const int h2 = kc >> 1, g0 = (kc & 1) * 2;
        sk_u32x2 qh[2], ql[2];
#pragma unroll
        for (int gg = 0; gg < 2; gg++) {
          const int g = g0 + gg;
          sk_u32x4 qm = {0u, 0u, 0u, 0u};
          if (lmask) qm = __builtin_amdgcn_raw_buffer_load_b128(r_x0, voff_in + (SK_QOFF(h2, g)), 0, 0);
          sk_u32x4 qx;
          float ov[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int i = 4 * g + j;
            float cv = acc[h2][i];
            if (DROP && p.drop_p > 0.f && rin)
              cv *= dropout_scale(dseed, (unsigned long long)(nbase + t) * 64 + h2 * 32 + 8 * g + 4 * half + j, p.drop_p);
            float o = sk_res_bwd(dxo[h2][i], rs, cv);
            if (lmask) o *= (sk_u2f(qm[j]) > 0.f ? 1.f : p.slope);
            o = rin ? o : 0.f;
            dxo[h2][i] = o;
            ov[j] = o;
            qx[j] = sk_f2u(o);
          }
          if (!FOLD && l == 0) __builtin_amdgcn_raw_buffer_store_b128(qx, r_x, voff_x0 + (SK_QOFF(h2, g)), 0, 0);
          sk_quad<PRECISE>(ov[0], ov[1], ov[2], ov[3], qh[gg], ql[gg]);
        }
        // bf16 dX_l: the out-conv weight gradient of block l-1 (l = 0: of the first conv) reads it
        __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(qh[0], qh[1])), r_dh, voff_b + (kc * 32), 0, 0);
        if (PRECISE)
          __builtin_amdgcn_raw_buffer_store_b128(sk_frag_bits(sk_swap_frag(ql[0], ql[1])), r_dl, voff_b + (kc * 32), 0, 0);
      }
      SKB_T(5)
      SKB_CLOSE
    }
    if (has_aux) {
      SKB_OPEN(l > 0, LYn.w_os)
      // ---- conditioning gradient, accumulated over the blocks ----
      const unsigned char* gf_hi = gs_hi + (SK_GUARD + row) * GS + half * 16;
      const unsigned char* gf_lo = gs_lo + (SK_GUARD + row) * GS + half * 16;
      if constexpr (PRECISE) {
#pragma unroll
        for (int kc = 0; kc < 8; kc++) {
          const bf16x8 x_hi = lds_frag(gf_hi + kc * 32);
          const bf16x8 x_lo = lds_frag(gf_lo + kc * 32);
          SKB_MMA(accc, wf_hi, kc, x_hi, x_lo)
        }
      } else {
        SKB_PIPE(accc, gf_hi)
      }
      SKB_T(4)
      SKB_CLOSE
    }
  }
#undef SKB_TAP
#undef SKB_PIPE
#undef SKB_SG
#undef SKB_LD3

  if (has_aux && rout) {
    float* dcr = p.dc + (nbase + t) * p.lddc;
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ch = h2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
        if (ch < p.aux_ch) dcr[ch] = accc[h2][i];
      }
  }
  if constexpr (FOLD) {
    // ---- the first conv's data gradient last: dx = dx_scale * Wfirst^T . bf16(dX_0), in_rows / 32 output tiles; the
    // transposed weights ([in_rows][64]) fill both weight buffers at once (the chain is done with them) ----
    if (p.dx != nullptr) {
      __syncthreads();  // last chunk consumed by everybody
      unsigned char* wb = smem + p.o_whi;
      for (int idx = tid; idx < p.in_rows * 8; idx += NT)
        *reinterpret_cast<sk_u32x4*>(wb + (idx >> 3) * SK_XS + (idx & 7) * 16) = *reinterpret_cast<const sk_u32x4*>(p.whi + p.w_first + (long)idx * 8);
      bf16x8 xq[4];
#pragma unroll
      for (int kc = 0; kc < 4; kc++) {
        const int h2 = kc >> 1, g0 = (kc & 1) * 2;
        sk_u32x2 qh[2], ql[2];
#pragma unroll
        for (int gg = 0; gg < 2; gg++) {
          const int i0 = 4 * (g0 + gg);
          sk_quad<false>(dxo[h2][i0], dxo[h2][i0 + 1], dxo[h2][i0 + 2], dxo[h2][i0 + 3], qh[gg], ql[gg]);
        }
        xq[kc] = sk_swap_frag(qh[0], qh[1]);
      }
      __syncthreads();
      const unsigned char* wff = wb + l31 * SK_XS + half * 16;
      const __amdgpu_buffer_rsrc_t rdx = sk_rsrc(p.dx, (long)p.B * p.T * p.lddx);
      const int ntile = p.in_rows >> 5;
      for (int nt = 0; nt < ntile; nt++) {
        f32x16 a;
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; kc++) a = mfma_bf16(lds_frag(wff + nt * 32 * SK_XS + kc * 32), xq[kc], a);
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int ch = nt * 32 + 8 * g + 4 * half;
          sk_u32x4 q;
#pragma unroll
          for (int j = 0; j < 4; j++) q[j] = sk_f2u(a[4 * g + j] * p.dx_scale);
          __builtin_amdgcn_raw_buffer_store_b128(q, rdx, (rout && ch < p.in_ch) ? (int)(((nbase + t) * p.lddx + ch) * 4) : SK_OOB, 0, 0);
        }
      }
    }
  }
#ifdef SKB_PROF
  SKB_T(5)
  pacc_[7] = __builtin_readcyclecounter() - pstart_;
  if (blockIdx.x < 256 && lane == 0 && wave < 8) {
#pragma unroll
    for (int i = 0; i < 12; i++) skb_prof_buf[(blockIdx.x * 8 + wave) * 12 + i] = pacc_[i];
  }
#endif
}

// waves per workgroup the data-gradient chain will run with (CRK_SK_NW=FB: forward digit F, data-gradient digit B; debugging)
int stack_bwd_waves(bool precise) {
  static int nw_env = -1;
  if (nw_env < 0) nw_env = crk_sw().sk_nw_bwd;
  return precise ? 4 : (nw_env == 4 || nw_env == 6 || nw_env == 8 ? nw_env : 8);
}

int stack_bwd_plan(StackBP& p, bool precise) {
  p.nw = stack_bwd_waves(precise);
  if (p.nw == 8 && 256 - p.hl - p.hr < 32) return CRK_ERR_UNSUPPORTED;
  const int R = p.nw * 32;
  p.tmo = R - p.hl - p.hr;
  if (p.tmo < 32 || p.max_off > SK_GUARD || p.ktaps > 8 || p.aux_ch > 64) return CRK_ERR_UNSUPPORTED;
  p.tiles_per_utt = ceil_div(p.T, p.tmo);
  p.tmo = ceil_div(p.T, p.tiles_per_utt);
  const int gbytes = (SK_GUARD * 2 + R) * SKB_GS;
  p.w_bytes = 64 * SKB_GS;
  int off = gbytes;
  p.o_glo = off; if (precise) off += gbytes;
  p.o_whi = off; off += precise ? p.w_bytes : 2 * p.w_bytes;
  p.o_wlo = off; if (precise) off += p.w_bytes;
  p.lds_bytes = (off + 15) & ~15;
  return p.lds_bytes <= 160 * 1024 ? CRK_OK : CRK_ERR_UNSUPPORTED;
}

int launch_stack_bwd(const StackBP& p, bool precise, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[8] = {(const void*)stack_bwd_kernel<true, true, 4>,  (const void*)stack_bwd_kernel<true, false, 4>,
                          (const void*)stack_bwd_kernel<false, true, 4>, (const void*)stack_bwd_kernel<false, false, 4>,
                          (const void*)stack_bwd_kernel<false, true, 6>, (const void*)stack_bwd_kernel<false, false, 6>,
                          (const void*)stack_bwd_kernel<false, true, 8>, (const void*)stack_bwd_kernel<false, false, 8>};
    for (int i = 0; i < 8; i++)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return CRK_ERR_HIP;
    attr_set = true;
  }
  dim3 grid(p.B * p.tiles_per_utt);
  const double nfr = (double)p.B * p.T;
  const bool has_aux = p.dc != nullptr && p.aux_ch > 0;
  // algorithmic bytes: dS and the tanh / sigmoid planes read; dG, dX planes, bf16 dS, fp32 dX_0 (and dc) written
  conv_prof_bytes(2, nfr * (256.0 + 256.0 * p.L + (p.mask_l0 ? 256.0 : 0.0) + 256.0 * p.L + 128.0 * p.L + 128.0 + 256.0 +
                            (has_aux ? 4.0 * p.aux_ch : 0.0)));
  conv_prof_begin(2, 2.0 * nfr * p.L * (64.0 * 128.0 * (1 + p.ktaps) + (has_aux ? 128.0 * p.aux_ch : 0.0)), s);
  const bool drop = p.drop_p > 0.f;
#define SKB_LAUNCH(PR, DR, NWV) hipLaunchKernelGGL((stack_bwd_kernel<PR, DR, NWV>), grid, dim3(NWV * 64), p.lds_bytes, s, p)
  if (!precise && !drop && p.nw == 8 && p.dy != nullptr) {
    static bool attr_f = false;
    if (!attr_f) {
      if (hipFuncSetAttribute((const void*)stack_bwd_kernel<false, false, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return CRK_ERR_HIP;
      attr_f = true;
    }
    hipLaunchKernelGGL((stack_bwd_kernel<false, false, 8, true>), grid, dim3(512), p.lds_bytes, s, p);
  }
  else if (precise) { if (drop) SKB_LAUNCH(true, true, 4); else SKB_LAUNCH(true, false, 4); }
  else if (p.nw == 4) { if (drop) SKB_LAUNCH(false, true, 4); else SKB_LAUNCH(false, false, 4); }
  else if (p.nw == 6) { if (drop) SKB_LAUNCH(false, true, 6); else SKB_LAUNCH(false, false, 6); }
  else { if (drop) SKB_LAUNCH(false, true, 8); else SKB_LAUNCH(false, false, 8); }
#undef SKB_LAUNCH
  conv_prof_end(2, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// =====================================================================================
// Weight gradients of ALL gated residual blocks of a stack in one launch, from the bf16 planes
// the two fused kernels above leave behind (block inputs as the conv saw them, z, conditioning;
// dG_l, dX_l, dS).  Workgroup (g, l) owns block l and a group of utterances and produces, per
// 64-frame chunk (32 for bf16x3):
//   dWconv[tap] (128 x 64) += dG^T . X[t + off0 + tap*dil]      dWaux (128 x aux) += dG^T . C
//   dWout|skip  (128 x 64) += [dX_{l+1} | dS]^T . Z              bias sums of dG and [dX_{l+1} | dS]
// Every plane row is read ONCE per block (the per-tap / per-conv re-reads of the generic
// table kernel were its bottleneck), as 16-byte pieces that go through registers (prefetch of
// the next chunk overlaps the MFMAs of the current one) into row-major LDS tiles; both MFMA
// operands need the reduction (frame) axis contiguous per lane and come out of LDS through
// ds_read_b64_tr_b16.  8 waves = 4 output-channel bands x 2 input-channel bands; a wave holds
// its KT tap tiles + one out|skip tile + one aux tile in accumulators.  The per-group partial
// sums use the layout of the table kernel, so the weight-norm backward reduces both alike.
#define SW_RA 320  // row stride of a 128-channel tile (128 bf16 + 64 B: 4 consecutive rows on distinct bank quarters)
#define SW_RB 192  // row stride of a 64-channel tile
#define SW_SPAN 32 // largest tap span (frames) a chunk can carry

struct SwRegs {
  sk_u32x4 g0, g1, dx, ds, x0, x1, z, c;
};

template <bool PRECISE, int KT>
__global__ __launch_bounds__(512, 2) void stack_wgrad_kernel(const StackWP p) {
  constexpr int FR = PRECISE ? 32 : 64, RA = SW_RA, RB = SW_RB, XR = FR + SW_SPAN;
  constexpr int O_GT = 0, O_DT = FR * RA, O_XT = 2 * FR * RA, O_ZT = O_XT + XR * RB, O_CT = O_ZT + FR * RB;
  constexpr int PLANE = O_CT + FR * RB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* t_hi = smem;
  unsigned char* t_lo = smem + PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, l = blockIdx.y;
  const StackWLayer LY = p.layers[l];
  const long N64 = (long)p.B * p.T * 64;
  const int span = (KT - 1) * LY.dil, xrn = FR + span;
  const bool has_aux = p.cb_hi != nullptr;
  const bool has_dx = l + 1 < p.L;
  const int qpa = p.aux_pad >> 3;

  const uint16_t* gh = p.gb_hi + (long)l * 2 * N64;
  const uint16_t* gl = PRECISE ? p.gb_lo + (long)l * 2 * N64 : nullptr;
  const uint16_t* dxh = p.dxb_hi + (long)(l + 1) * N64;
  const uint16_t* dxl = PRECISE ? p.dxb_lo + (long)(l + 1) * N64 : nullptr;
  const uint16_t* xh = p.xb_hi + (long)l * N64;
  const uint16_t* xl = PRECISE ? p.xb_lo + (long)l * N64 : nullptr;
  const uint16_t* zh = p.zb_hi + (long)l * N64;
  const uint16_t* zl = PRECISE ? p.zb_lo + (long)l * N64 : nullptr;

  // ---- fixed per-thread piece geometry (16-byte pieces) ----
  // dG: 16 pieces per row, second piece 32 rows below; 64-channel planes: 8 pieces per row.  Row-major planes: a thread's
  // piece index runs first (a wave reads 4 / 8 whole rows); records: the frame within a 4-frame record first, then the
  // piece - consecutive lanes read consecutive bytes either way (a wave: 1 KB), and the 16 lanes of an LDS write hit 4
  // consecutive rows x 4 pieces = all 64 banks once (row strides of 320 / 192 bytes put 4 rows on 4 bank quarters)
  const int g_r0 = p.rec_g ? ((tid >> 6) << 2) | (tid & 3) : tid >> 4, g_c = p.rec_g ? (tid >> 2) & 15 : tid & 15;
  const int h_r = tid >> 3, h_c = tid & 7;              // the forward's planes (x, z)
  const int hg_r = p.rec_g ? ((tid >> 5) << 2) | (tid & 3) : h_r, hg_c = p.rec_g ? (tid >> 2) & 7 : h_c;  // the chain's (dX, dS)
  const int c_r = qpa ? tid / qpa : 0, c_c = qpa ? tid - c_r * qpa : 0;

  SwRegs Rh, Rl;
  const sk_u32x4 Z4 = {0u, 0u, 0u, 0u};
#define SW_LD(dst, ptr, on, off) dst = (on) ? *reinterpret_cast<const sk_u32x4*>((ptr) + (off)) : Z4;
// element offset of the 16-byte piece (frame n, channels 8c ..) in a plane of 128 / 64 channels: row-major, or 4-frame
// records (StackWP::rec_g: the planes of the data-gradient chain, StackBP::rec)
#define SW_OFF128(n, c, rec) ((rec) ? ((n) >> 2) * 512 + (c) * 32 + ((n) & 3) * 8 : (n) * 128 + (c) * 8)
#define SW_OFF64(n, c, rec) ((rec) ? ((n) >> 2) * 256 + (c) * 32 + ((n) & 3) * 8 : (n) * 64 + (c) * 8)
#define SW_FETCH(nb, f0)                                                                                  \
  {                                                                                                       \
    {                                                                                                     \
      const int t = (f0) + g_r0;                                                                          \
      const bool on = g_r0 < FR && t < p.T;                                                               \
      const long off = SW_OFF128((nb) + t, g_c, p.rec_g);                                                 \
      SW_LD(Rh.g0, gh, on, off) if (PRECISE) SW_LD(Rl.g0, gl, on, off)                                    \
    }                                                                                                     \
    if (FR > 32) {                                                                                        \
      const int t = (f0) + g_r0 + 32;                                                                     \
      const bool on = t < p.T;                                                                            \
      const long off = SW_OFF128((nb) + t, g_c, p.rec_g);                                                 \
      SW_LD(Rh.g1, gh, on, off) if (PRECISE) SW_LD(Rl.g1, gl, on, off)                                    \
    }                                                                                                     \
    {                                                                                                     \
      const int t = (f0) + h_r;                                                                           \
      const bool on = h_r < FR && t < p.T;                                                                \
      const int tg = (f0) + hg_r;                                                                         \
      const bool ong = hg_r < FR && tg < p.T;                                                             \
      const long off = SW_OFF64((nb) + tg, hg_c, p.rec_g), offx = ((nb) + t) * 64 + h_c * 8;              \
      SW_LD(Rh.dx, dxh, ong && has_dx, off) if (PRECISE) SW_LD(Rl.dx, dxl, ong && has_dx, off)            \
      SW_LD(Rh.ds, p.dsb_hi, ong, off) if (PRECISE) SW_LD(Rl.ds, p.dsb_lo, ong, off)                      \
      SW_LD(Rh.z, zh, on, offx) if (PRECISE) SW_LD(Rl.z, zl, on, offx)                                    \
    }                                                                                                     \
    {                                                                                                     \
      const int t = (f0) + LY.off0 + h_r;                                                                 \
      const bool on = h_r < xrn && t >= 0 && t < p.T;                                                     \
      const long off = ((nb) + t) * 64 + h_c * 8;                                                         \
      SW_LD(Rh.x0, xh, on, off) if (PRECISE) SW_LD(Rl.x0, xl, on, off)                                    \
    }                                                                                                     \
    if (XR > 64) {                                                                                        \
      const int r = h_r + 64, t = (f0) + LY.off0 + r;                                                     \
      const bool on = r < xrn && t >= 0 && t < p.T;                                                       \
      const long off = ((nb) + t) * 64 + h_c * 8;                                                         \
      SW_LD(Rh.x1, xh, on, off) if (PRECISE) SW_LD(Rl.x1, xl, on, off)                                    \
    }                                                                                                     \
    if (has_aux) {                                                                                        \
      const int t = (f0) + c_r;                                                                           \
      const bool on = c_r < FR && t < p.T;                                                                \
      const long off = ((nb) + t) * p.aux_pad + c_c * 8;                                                  \
      SW_LD(Rh.c, p.cb_hi, on, off) if (PRECISE) SW_LD(Rl.c, p.cb_lo, on, off)                            \
    }                                                                                                     \
  }
#define SW_ST(tile_off, val_h, val_l)                                                                     \
  {                                                                                                       \
    *reinterpret_cast<sk_u32x4*>(t_hi + (tile_off)) = val_h;                                                 \
    if (PRECISE) *reinterpret_cast<sk_u32x4*>(t_lo + (tile_off)) = val_l;                                    \
  }
#define SW_COMMIT()                                                                                       \
  {                                                                                                       \
    if (g_r0 < FR) SW_ST(O_GT + g_r0 * RA + g_c * 16, Rh.g0, Rl.g0)                                       \
    if (FR > 32) SW_ST(O_GT + (g_r0 + 32) * RA + g_c * 16, Rh.g1, Rl.g1)                                  \
    if (hg_r < FR) {                                                                                      \
      SW_ST(O_DT + hg_r * RA + hg_c * 16, Rh.dx, Rl.dx)                                                   \
      SW_ST(O_DT + hg_r * RA + 128 + hg_c * 16, Rh.ds, Rl.ds)                                             \
    }                                                                                                     \
    if (h_r < FR) SW_ST(O_ZT + h_r * RB + h_c * 16, Rh.z, Rl.z)                                           \
    if (h_r < xrn) SW_ST(O_XT + h_r * RB + h_c * 16, Rh.x0, Rl.x0)                                        \
    if (XR > 64 && h_r + 64 < xrn) SW_ST(O_XT + (h_r + 64) * RB + h_c * 16, Rh.x1, Rl.x1)                 \
    if (has_aux && c_r < FR) SW_ST(O_CT + c_r * RB + c_c * 16, Rh.c, Rl.c)                                \
  }

  // aux tile columns beyond aux_pad are never written: clear the tile once
  if (has_aux && p.aux_pad < 64) {
    for (int i = tid; i < FR * RB / 16; i += 512) {
      reinterpret_cast<sk_u32x4*>(t_hi + O_CT)[i] = Z4;
      if (PRECISE) reinterpret_cast<sk_u32x4*>(t_lo + O_CT)[i] = Z4;
    }
  }

  // ---- MFMA geometry: wave = (output-channel band ct, input-channel band it) ----
  const int ct = wave & 3, it = wave >> 2;
  const int i15 = lane & 15, grp = lane >> 4;
  const int rowoff = (grp >> 1) * 8 + (i15 >> 2);
  const int coloff = (grp & 1) * 16 + (i15 & 3) * 4;
  const int half = lane >> 5, l31 = lane & 31;
  f32x16 accv[KT], acco, acca;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    acco[i] = 0.f; acca[i] = 0.f;
#pragma unroll
    for (int a = 0; a < KT; a++) accv[a][i] = 0.f;
  }
  float bsum = 0.f;  // it == 0: dG column sums of band ct; it == 1: [dX | dS] column sums

  // group g owns the 64-frame chunks [g * cpg, (g + 1) * cpg) of the batch, counted utterance after utterance (an
  // utterance has ceil(T / 64) of them; with FR = 32 a chunk is walked as two halves, the second one empty where it
  // starts behind the utterance's end: every load is masked by t < T)
  const int ncpu = (p.T + 63) / 64, uspan = ncpu * 64;
  const int c_tot = p.B * ncpu;
  const int c_beg = min(c_tot, g * p.cpg), c_end = min(c_tot, c_beg + p.cpg);
  const int nchunks = (c_end - c_beg) * (64 / FR);
  const int u_beg = c_beg / ncpu;
  long nbn = (long)u_beg * p.T;
  int f0n = (c_beg - u_beg * ncpu) * 64;
  if (nchunks > 0) SW_FETCH(nbn, f0n)
  for (int c = 0; c < nchunks; c++) {
    __syncthreads();  // previous chunk's fragments consumed
    SW_COMMIT()
    __syncthreads();
    if (c + 1 < nchunks) {
      f0n += FR;
      if (f0n >= uspan) { f0n = 0; nbn += p.T; }
      SW_FETCH(nbn, f0n)
    }
    const unsigned char* ag_hi = t_hi + O_GT + rowoff * RA + (ct * 32 + coloff) * 2;
    const unsigned char* ad_hi = t_hi + O_DT + rowoff * RA + (ct * 32 + coloff) * 2;
    const unsigned char* bx_hi = t_hi + O_XT + rowoff * RB + (it * 32 + coloff) * 2;
    const unsigned char* bz_hi = t_hi + O_ZT + rowoff * RB + (it * 32 + coloff) * 2;
    const unsigned char* bc_hi = t_hi + O_CT + rowoff * RB + (it * 32 + coloff) * 2;
#pragma unroll
    for (int kc = 0; kc < FR / 16; kc++) {
      const bf16x8 a_hi = sw_tr_frag(ag_hi + kc * 16 * RA, RA);
      const bf16x8 d_hi = sw_tr_frag(ad_hi + kc * 16 * RA, RA);
      bf16x8 a_lo, d_lo;
      if (PRECISE) {
        a_lo = sw_tr_frag(ag_hi + PLANE + kc * 16 * RA, RA);
        d_lo = sw_tr_frag(ad_hi + PLANE + kc * 16 * RA, RA);
      }
      if (it == 0) bsum += sw_sum8(a_hi) + (PRECISE ? sw_sum8(a_lo) : 0.f);
      else bsum += sw_sum8(d_hi) + (PRECISE ? sw_sum8(d_lo) : 0.f);
#pragma unroll
      for (int a = 0; a < KT; a++) {
        const int boff = (kc * 16 + a * LY.dil) * RB;
        const bf16x8 b_hi = sw_tr_frag(bx_hi + boff, RB);
        accv[a] = mfma_bf16(a_hi, b_hi, accv[a]);
        if (PRECISE) {
          const bf16x8 b_lo = sw_tr_frag(bx_hi + PLANE + boff, RB);
          accv[a] = mfma_bf16(a_lo, b_hi, accv[a]);
          accv[a] = mfma_bf16(a_hi, b_lo, accv[a]);
        }
      }
      {
        const bf16x8 b_hi = sw_tr_frag(bz_hi + kc * 16 * RB, RB);
        acco = mfma_bf16(d_hi, b_hi, acco);
        if (PRECISE) {
          const bf16x8 b_lo = sw_tr_frag(bz_hi + PLANE + kc * 16 * RB, RB);
          acco = mfma_bf16(d_lo, b_hi, acco);
          acco = mfma_bf16(d_hi, b_lo, acco);
        }
      }
      if (has_aux) {
        const bf16x8 b_hi = sw_tr_frag(bc_hi + kc * 16 * RB, RB);
        acca = mfma_bf16(a_hi, b_hi, acca);
        if (PRECISE) {
          const bf16x8 b_lo = sw_tr_frag(bc_hi + PLANE + kc * 16 * RB, RB);
          acca = mfma_bf16(a_lo, b_hi, acca);
          acca = mfma_bf16(a_hi, b_lo, acca);
        }
      }
    }
  }

  // ---- this group's partial sums (layout of the table kernel / weight-norm backward) ----
  const int ci = it * 32 + l31;
#pragma unroll
  for (int a = 0; a < KT; a++) {
    float* out = p.partials + LY.pt_conv + ((long)g * KT + a) * 128 * 64;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int co = ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
      out[co * 64 + ci] = accv[a][i];
    }
  }
  {
    float* out = p.partials + LY.pt_os + (long)g * 128 * 64;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int co = ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
      out[co * 64 + ci] = acco[i];
    }
  }
  if (has_aux && ci < p.aux_ch) {
    float* out = p.partials + LY.pt_aux + (long)g * 128 * p.aux_ch;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int co = ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
      out[co * p.aux_ch + ci] = acca[i];
    }
  }
  {
    const float tot = bsum + __shfl_xor(bsum, 32);
    const long long pb = it == 0 ? LY.pb_conv : LY.pb_os;
    if (half == 0 && pb >= 0) p.partials[pb + (long)g * 128 + ct * 32 + l31] = tot;
  }
}

int stack_wgrad_supported(int ktaps, int max_dil, int aux_ch) {
  return (ktaps == 3 || ktaps == 5) && (ktaps - 1) * max_dil <= SW_SPAN && aux_ch <= 64;
}

int launch_stack_wgrad(const StackWP& p, bool precise, hipStream_t s) {
  const int FR = precise ? 32 : 64;
  const int plane = 2 * FR * SW_RA + (FR + SW_SPAN) * SW_RB + 2 * FR * SW_RB;
  const int lds = (precise ? 2 : 1) * plane;
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[4] = {(const void*)stack_wgrad_kernel<true, 3>, (const void*)stack_wgrad_kernel<true, 5>,
                          (const void*)stack_wgrad_kernel<false, 3>, (const void*)stack_wgrad_kernel<false, 5>};
    for (int i = 0; i < 4; i++)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return CRK_ERR_HIP;
    attr_set = true;
  }
  dim3 grid(p.G, p.L);
  const double nfr = (double)p.B * p.T;
  conv_prof_bytes(5, nfr * p.L * (768.0 + (p.cb_hi ? 2.0 * p.aux_pad : 0.0)));  // every plane row once per block
  conv_prof_begin(5, 2.0 * nfr * p.L * (128.0 * 64.0 * (p.ktaps + 1) + 128.0 * p.aux_ch), s);
  if (precise) {
    if (p.ktaps == 3) hipLaunchKernelGGL((stack_wgrad_kernel<true, 3>), grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL((stack_wgrad_kernel<true, 5>), grid, dim3(512), lds, s, p);
  } else {
    if (p.ktaps == 3) hipLaunchKernelGGL((stack_wgrad_kernel<false, 3>), grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL((stack_wgrad_kernel<false, 5>), grid, dim3(512), lds, s, p);
  }
  conv_prof_end(5, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
