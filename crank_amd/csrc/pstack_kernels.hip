// Fused chains of plain Conv1d layers: every conv of the step that is not a gated residual
// block - the speaker classifier C and the speaker-adversarial net (stacks of dilated
// Conv1d + LeakyReLU: parallel_wavegan ParallelWaveGANDiscriminator, SURVEY.md Appendix A.4;
// call sites crank/bin/train.py:78-89, crank/net/module/spkradv.py:49-60), and the 1x1 first
// conv and the "act -> 1x1 -> act -> 1x1" head around every gated stack (Appendix A.1-A.3).
//
// One kernel runs a chain in EITHER direction: a layer is "rows x K weights per tap applied to
// a [frames][K] operand", followed by an epilogue that is an activation (forward) or a multiply
// by the activation derivative read from a saved plane (data gradient, weights in the
// transposed tap-flipped layout).  As in stack_kernels.hip a workgroup owns a window of
// 32*NW frames (+ the chain's halo), MFMA A = weights / B = operand so that a lane owns one
// frame, the operand of the next layer goes accumulator -> bf16 -> v_permlane32_swap ->
// LDS without leaving the CU, and the operand of EVERY layer is also stored as a bf16 plane:
// the forward's planes are the weight gradient's input operands and the activation-derivative
// masks, the backward's planes are its output-gradient operands.
//
// The generic per-layer kernels (conv_kernels.hip) remain as the fallback for shapes these
// kernels do not take; their floor per launch is the instruction-cache fill of ~50 KB of
// code, which is what made ~60 of them per step cost more than the gated stacks themselves.
#include "conv_kernels.h"
#include "stack_common.h"


// Weights reach LDS in CHUNKS of whole taps: as many consecutive taps of a layer as fit the staging registers
// (PS_MAXP 16-byte pieces per thread) and the chunk buffer (PsP::w_bytes).  The next chunk is fetched (global ->
// registers) before the MFMAs of the current one and committed (registers -> LDS) after them, so its L2 round trip
// hides behind tpc taps of MFMAs and the chain takes one barrier per chunk.  (One tap per chunk - the first version -
// left ~1.5 us per tap: a tap's MFMAs are ~0.3 us, the round trip ~1 us.)
#ifndef PS_ABL
#define PS_ABL 0  // ablation builds (tools/ps_ablate.sh): 1 no MFMA, 2 no plane stores, 4 no weight fragment reads, 8 no operand
#endif            // fragment reads, 16 no weight chunk traffic after the first chunk
// Phase cycles (tools/ps_phase_cycles.py, -DPS_PROF): per workgroup and wave [0] prologue [1] tap MFMAs [2] waiting at the
// chunk barrier [3] weight commit (incl. waiting for the fetch) [4] epilogue (incl. its barrier) [5] whole kernel
#ifdef PS_PROF
__device__ unsigned long long ps_prof_buf[256 * 8 * 8];
__device__ unsigned long long ps_prof_res[1024 * 2];
extern "C" int crk_debug_ps_prof(unsigned long long* out, unsigned long long* res) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ps_prof_buf), sizeof(unsigned long long) * 256 * 8 * 8) != hipSuccess) return 2;
  return hipMemcpyFromSymbol(res, HIP_SYMBOL(ps_prof_res), sizeof(unsigned long long) * 1024 * 2) == hipSuccess ? 0 : 2;
}
#define PS_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pacc_[i] += t_ - plast_; plast_ = t_; }
#else
#define PS_T(i)
#endif
#define PS_WFRAG(ptr) ((PS_ABL & 4) ? xb : lds_frag(ptr))
#define PS_XFRAG(ptr) ((PS_ABL & 8) ? x0 : lds_frag(ptr))
#define PS_MAXP(NT) ((NT) == 512 ? 4 : 8)
__host__ __device__ __forceinline__ int ps_tpc(int k, int rows_pad, int kp, int cap_pieces, int w_bytes) {
  int t = cap_pieces / (rows_pad * (kp >> 3));
  const int tl = w_bytes / (rows_pad * (kp * 2 + 16));
  if (tl < t) t = tl;
  if (t > k) t = k;
  return t < 1 ? 1 : t;
}

template <bool PRECISE, int NW>
__global__ __launch_bounds__(NW * 64, PRECISE ? 1 : 2) void pstack_kernel(const PsP p) {
  constexpr int NT = NW * 64, R = NW * 32, MAXP = PS_MAXP(NT);
  const int OS = p.os;  // row stride of the operand tile: widest K of the chain as bf16 + 16 B pad (a layer's weight rows: its own kp * 2 + 16)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
#ifdef PS_PROF
  unsigned long long pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pstart_ = __builtin_readcyclecounter(), preal_ = __builtin_amdgcn_s_memrealtime();
  unsigned long long plast_ = pstart_;
#endif
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;
  const long nbase = (long)b * p.T, N = (long)p.B * p.T;

  unsigned char* os_hi = smem;  // [SK_GUARD + R + SK_GUARD][OS]
  unsigned char* os_lo = smem + p.o_olo;
  // weight buffer i as an OFFSET from the LDS base: a runtime-indexed array of pointers would decay to
  // generic pointers and every weight-fragment access to a FLAT instruction (vmcnt + lgkmcnt, i.e.
  // serialised with the global weight prefetch) instead of ds_read / ds_write
#define WS_HI(i) (smem + p.o_whi + ((PRECISE || (i) == 0) ? 0 : p.w_bytes))
  unsigned char* ws_lo = smem + p.o_wlo;

  const int row = wave * 32 + l31;
  const int t = t0 - p.hl + row;
  const bool rin = t >= 0 && t < p.T;
  const bool rout = rin && row >= p.hl && row < p.hl + p.tmo;
  const long n = nbase + t;

  // ---- weight chunk prefetch (ntaps x rows_pad x kp bf16 = ntaps*rows_pad*kp/8 16-byte pieces, <= MAXP * NT) ----
  sk_u32x4 wr_h[MAXP], wr_l[MAXP];
#define PS_FETCH(LYX, tap, ntaps)                                                            \
  {                                                                                          \
    const int total = (ntaps) * (LYX).rows_pad * ((LYX).kp >> 3);                            \
    const long base = (LYX).w_off + (long)(tap) * (LYX).rows_pad * (LYX).kp;                 \
    _Pragma("unroll") for (int u = 0; u < MAXP; u++) {                                       \
      const int idx = tid + u * NT;                                                          \
      const long off = base + (idx < total ? (long)idx * 8 : 0);                             \
      wr_h[u] = *reinterpret_cast<const sk_u32x4*>(p.whi + off);                             \
      if (PRECISE) wr_l[u] = *reinterpret_cast<const sk_u32x4*>(p.wlo + off);                \
    }                                                                                        \
  }
  /* rows of consecutive taps follow each other in the chunk buffer: row (tap - tap0) * rows_pad + r */          \
#define PS_COMMIT(LYX, ntaps, dhi)                                                           \
  {                                                                                          \
    const int ppr = (LYX).kp >> 3, total = (ntaps) * (LYX).rows_pad * ppr, wst = (LYX).kp * 2 + 16; \
    _Pragma("unroll") for (int u = 0; u < MAXP; u++) {                                       \
      const int idx = tid + u * NT;                                                          \
      if (idx < total) {                                                                     \
        const int r = idx / ppr, c = idx - r * ppr;                                          \
        *reinterpret_cast<sk_u32x4*>((dhi) + r * wst + c * 16) = wr_h[u];                    \
        if (PRECISE) *reinterpret_cast<sk_u32x4*>(ws_lo + r * wst + c * 16) = wr_l[u];       \
      }                                                                                      \
    }                                                                                        \
  }
#define PS_TPC(LYX) ps_tpc((LYX).k, (LYX).rows_pad, (LYX).kp, MAXP * NT, p.w_bytes)

  PsLayer LY = p.layers[0];
  int tpc = PS_TPC(LY);
  PS_FETCH(LY, 0, tpc)

  // ---- layer-0 operand, phase 1: every load of this lane's input row is issued before anything waits
  // (one memory round trip for the whole row instead of one per 16-channel group) ----
  const __amdgpu_buffer_rsrc_t rx = sk_rsrc(p.x, N * p.ldx);
  const bool vec = ((p.ldx & 3) == 0) && ((p.cin & 3) == 0) && ((((uintptr_t)p.x) & 15) == 0);
  const int nk0 = LY.kp >> 4;
  sk_u32x4 xa[8], xc[8];
#pragma unroll
  for (int kc = 0; kc < 8; kc++)
    if (kc < nk0) {
      const int c0 = 16 * kc + 8 * half;
      if (vec) {
        const int vo = rin ? (int)((n * p.ldx + c0) * 4) : SK_OOB;
        xa[kc] = __builtin_amdgcn_raw_buffer_load_b128(rx, c0 + 3 < p.cin ? vo : SK_OOB, 0, 0);
        xc[kc] = __builtin_amdgcn_raw_buffer_load_b128(rx, c0 + 7 < p.cin ? vo + 16 : SK_OOB, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          xa[kc][j] = __builtin_amdgcn_raw_buffer_load_b32(rx, (rin && c0 + j < p.cin) ? (int)((n * p.ldx + c0 + j) * 4) : SK_OOB, 0, 0);
          xc[kc][j] = __builtin_amdgcn_raw_buffer_load_b32(rx, (rin && c0 + 4 + j < p.cin) ? (int)((n * p.ldx + c0 + 4 + j) * 4) : SK_OOB, 0, 0);
        }
      }
    }

  // layer table and biases into LDS once (inside the layer loop they would be global loads - a full
  // memory round trip each - in front of every layer's first MFMA)
  PsLayer* lay_s = reinterpret_cast<PsLayer*>(smem + p.o_tab);  // [L (+1 with tail)]
  float* bias_s = reinterpret_cast<float*>(smem + p.o_bias);     // [L][128]
  {
    const int nl = p.L + (p.tail ? 1 : 0);
    for (int i = tid; i < nl * (int)(sizeof(PsLayer) / 4); i += NT)
      reinterpret_cast<int*>(lay_s)[i] = reinterpret_cast<const int*>(p.layers)[i];
    for (int i = tid; i < p.L * 128; i += NT) {
      const PsLayer Y = p.layers[i >> 7];
      const int c = i & 127;
      bias_s[i] = (Y.b_off >= 0 && c < Y.rows) ? p.params[Y.b_off + c] : 0.f;
    }
  }

  // ---- guard rows ----
  for (int i = tid; i < SK_GUARD * OS / 16; i += NT) {
    const sk_u32x4 z4 = {0u, 0u, 0u, 0u};
    reinterpret_cast<sk_u32x4*>(os_hi)[i] = z4;
    reinterpret_cast<sk_u32x4*>(os_hi + (SK_GUARD + R) * OS)[i] = z4;
    if (PRECISE) {
      reinterpret_cast<sk_u32x4*>(os_lo)[i] = z4;
      reinterpret_cast<sk_u32x4*>(os_lo + (SK_GUARD + R) * OS)[i] = z4;
    }
  }

  unsigned char* my_os_hi = os_hi + (SK_GUARD + row) * OS + 8 * half * 2;
  unsigned char* my_os_lo = os_lo + (SK_GUARD + row) * OS + 8 * half * 2;

  // ---- layer-0 operand, phase 2: fp32 -> act -> bf16 fragments -> LDS tile and saved plane ----
  {
    const float in_sc = p.in_num ? p.in_scale * (p.in_num[0] / p.in_den[1]) : p.in_scale;
    const __amdgpu_buffer_rsrc_t r_sh = sk_rsrc16(p.save_hi ? p.save_hi + LY.save_plane : (const uint16_t*)p.x, N * LY.kp);
    const __amdgpu_buffer_rsrc_t r_sl = sk_rsrc16((p.save_hi && PRECISE) ? p.save_lo + LY.save_plane : (const uint16_t*)p.x, N * LY.kp);
    const int voff_s = (rout && p.save_hi) ? (int)((n * LY.kp + 8 * half) * 2) : SK_OOB;
#pragma unroll
    for (int kc = 0; kc < 8; kc++)
      if (kc < nk0) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; j++) { v[j] = sk_u2f(xa[kc][j]); v[4 + j] = sk_u2f(xc[kc][j]); }
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = apply_act(v[j] * in_sc, p.in_act, p.slope);
        const sk_u32x4 fh = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        *reinterpret_cast<sk_u32x4*>(my_os_hi + kc * 32) = fh;
        if (!(PS_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b128(fh, r_sh, voff_s + kc * 32, 0, 0);
        if (PRECISE) {
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = sk_bf_lo(v[j]);
          const sk_u32x4 fl = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
          *reinterpret_cast<sk_u32x4*>(my_os_lo + kc * 32) = fl;
          __builtin_amdgcn_raw_buffer_store_b128(fl, r_sl, voff_s + kc * 32, 0, 0);
        }
      }
  }
  PS_COMMIT(LY, tpc, WS_HI(0))

  int cur = 0;
  f32x16 acc[4];
  __syncthreads();  // table, biases, layer-0 operand, first weight chunk: staged
  PS_T(0)

  for (int l = 0; l < p.L; l++) {
    const int ntl = LY.rows_pad >> 5, nkc = LY.kp >> 4;
    const bool last = l + 1 == p.L;
    const bool fin = last && !p.tail;  // this layer's output is the chain's fp32 output
    PsLayer LN = LY;
    if (!fin) LN = lay_s[l + 1];
    const int wst = LY.kp * 2 + 16;  // row stride of this layer's weight chunk in LDS
    const int tpcn = PS_TPC(LN);
    // (requesting the activation-derivative masks of the data-gradient chains here, a layer of MFMAs ahead of their use in
    // the epilogue, was measured: no change - the epilogue's time is barrier skew, not the mask round trip)
    // accumulators start from the bias (rows of D = output channels)
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const sk_f32x4 bq = *reinterpret_cast<const sk_f32x4*>(bias_s + l * 128 + nt * 32 + 8 * g + 4 * half);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[nt][4 * g + j] = bq[j];
      }

    // The tap loop exists once per output-tile count (1..4 tiles of 32 channels), chosen per layer: with the
    // count as a run-time guard around each MFMA every MFMA sat in a basic block of its own behind its own
    // LDS round trip.  Inside, the k-steps run as a software pipeline: the fragments of step k+1 are loaded
    // before the MFMAs of step k are issued (the last step reloads its own fragments, branch-free).
#define PS_TAP_LOOP(NTL)                                                                                   \
  for (int c0 = 0; c0 < LY.k; c0 += tpc) {                                                                 \
    const int c1 = c0 + tpc < LY.k ? c0 + tpc : LY.k;                                                      \
    __syncthreads(); /* chunk `cur` committed; previous chunk's reads done; operand tile complete */       \
    PS_T(2)                                                                                                \
    const bool more = c1 < LY.k || !last;                                                                  \
    const int nnext = c1 < LY.k ? (LY.k - c1 < tpc ? LY.k - c1 : tpc) : (tpcn < LN.k ? tpcn : LN.k);       \
    if (!(PS_ABL & 16)) {                                                                                  \
      if (c1 < LY.k) PS_FETCH(LY, c1, nnext)                                                               \
      else if (!last) PS_FETCH(LN, 0, nnext)                                                               \
    }                                                                                                      \
    for (int tap = c0; tap < c1; tap++) {                                                                  \
      const unsigned char* wf_hi = WS_HI(cur) + ((tap - c0) * LY.rows_pad + l31) * wst + half * 16;        \
      const unsigned char* wf_lo = ws_lo + ((tap - c0) * LY.rows_pad + l31) * wst + half * 16;             \
      const int arow = SK_GUARD + row + LY.off0 + tap * LY.dil;                                            \
      const unsigned char* xf_hi = os_hi + arow * OS + half * 16;                                          \
      const unsigned char* xf_lo = os_lo + arow * OS + half * 16;                                          \
      const bf16x8 x0 = lds_frag(os_hi + (SK_GUARD + row) * OS + half * 16);                               \
      bf16x8 xb = PS_XFRAG(xf_hi), xl, wa[NTL], wl[NTL];                                                   \
      if (PRECISE) xl = lds_frag(xf_lo);                                                                   \
      _Pragma("unroll") for (int nt = 0; nt < NTL; nt++) {                                                 \
        wa[nt] = PS_WFRAG(wf_hi + nt * 32 * wst);                                                          \
        if (PRECISE) wl[nt] = lds_frag(wf_lo + nt * 32 * wst);                                             \
      }                                                                                                    \
      for (int kc = 0; kc < nkc; kc++) {                                                                   \
        const int kn = (kc + 1 < nkc ? kc + 1 : kc) * 32;                                                  \
        const bf16x8 nxb = PS_XFRAG(xf_hi + kn);                                                           \
        bf16x8 nxl, nwa[NTL], nwl[NTL];                                                                    \
        if (PRECISE) nxl = lds_frag(xf_lo + kn);                                                           \
        _Pragma("unroll") for (int nt = 0; nt < NTL; nt++) {                                               \
          nwa[nt] = PS_WFRAG(wf_hi + nt * 32 * wst + kn);                                                  \
          if (PRECISE) nwl[nt] = lds_frag(wf_lo + nt * 32 * wst + kn);                                     \
        }                                                                                                  \
        _Pragma("unroll") for (int nt = 0; nt < NTL; nt++) {                                               \
          if (!(PS_ABL & 1)) acc[nt] = mfma_bf16(wa[nt], xb, acc[nt]);                                     \
          else acc[nt][0] += __builtin_bit_cast(float, (unsigned)wa[nt][0] | ((unsigned)xb[0] << 16));     \
          if (PRECISE) {                                                                                   \
            acc[nt] = mfma_bf16(wa[nt], xl, acc[nt]);                                                      \
            acc[nt] = mfma_bf16(wl[nt], xb, acc[nt]);                                                      \
          }                                                                                                \
        }                                                                                                  \
        xb = nxb;                                                                                          \
        if (PRECISE) xl = nxl;                                                                             \
        _Pragma("unroll") for (int nt = 0; nt < NTL; nt++) {                                               \
          wa[nt] = nwa[nt];                                                                                \
          if (PRECISE) wl[nt] = nwl[nt];                                                                   \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
    if (PRECISE) __syncthreads();                                                                          \
    PS_T(1)                                                                                                \
    __builtin_amdgcn_sched_barrier(0); /* keep the commit (and its wait for the prefetch) behind the MFMAs */ \
    if (more && !(PS_ABL & 16)) {                                                                          \
      if (c1 < LY.k) PS_COMMIT(LY, nnext, WS_HI(PRECISE ? 0 : cur ^ 1))                                    \
      else PS_COMMIT(LN, nnext, WS_HI(PRECISE ? 0 : cur ^ 1))                                              \
    }                                                                                                      \
    if (!PRECISE) cur ^= 1;                                                                                \
    PS_T(3)                                                                                                \
  }
    if (ntl == 2) { PS_TAP_LOOP(2) }
    else if (ntl == 1) { PS_TAP_LOOP(1) }
    else if (ntl == 3) { PS_TAP_LOOP(3) }
    else { PS_TAP_LOOP(4) }
#undef PS_TAP_LOOP

    // ---- epilogue: activation (forward) or activation-derivative mask (data gradient) ----
    const bool is_mask = LY.epi >= 3;
    const int ekind = is_mask ? LY.epi - 2 : LY.epi;  // ACT_NONE / ACT_LRELU(1) / ACT_RELU(2)
    const __amdgpu_buffer_rsrc_t r_m = sk_rsrc16(is_mask ? p.mask_hi + LY.mask_plane : (const uint16_t*)p.x, N * LY.mask_w);
    if (!fin) {
      __syncthreads();  // every wave is past this layer's tap reads: the operand tile may be rewritten
      const __amdgpu_buffer_rsrc_t r_sh = sk_rsrc16(p.save_hi ? p.save_hi + LN.save_plane : (const uint16_t*)p.x, N * LN.kp);
      const __amdgpu_buffer_rsrc_t r_sl = sk_rsrc16((p.save_hi && PRECISE) ? p.save_lo + LN.save_plane : (const uint16_t*)p.x, N * LN.kp);
      const int voff_s = (rout && p.save_hi) ? (int)((n * LN.kp + 8 * half) * 2) : SK_OOB;
      const int nk2 = LN.kp >> 4;  // 16-channel groups of the next operand (<= 2 * ntl)
#pragma unroll
      for (int kc = 0; kc < 8; kc++)
        if (kc < nk2) {
          const int nt = kc >> 1, g0 = (kc & 1) * 2;
          sk_u32x2 qh[2], ql[2];
#pragma unroll
          for (int gg = 0; gg < 2; gg++) {
            const int g = g0 + gg, c0 = nt * 32 + 8 * g + 4 * half;
            float v[4];
            if (is_mask) {
              const sk_u32x2 m = __builtin_amdgcn_raw_buffer_load_b64(r_m, rin ? (int)((n * LY.mask_w + c0) * 2) : SK_OOB, 0, 0);
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const unsigned w = j < 2 ? m[0] : m[1];
                const float mv = sk_u2f((j & 1) ? (w & 0xffff0000u) : (w << 16));
                v[j] = acc[nt][4 * g + j] * act_grad(mv, ekind, p.slope);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; j++) v[j] = apply_act(acc[nt][4 * g + j], ekind, p.slope);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = rin ? v[j] : 0.f;
            sk_quad<PRECISE>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);
          }
          const sk_u32x4 fh = sk_frag_bits(sk_swap_frag(qh[0], qh[1]));
          *reinterpret_cast<sk_u32x4*>(my_os_hi + kc * 32) = fh;
          if (!(PS_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b128(fh, r_sh, voff_s + kc * 32, 0, 0);
          if (PRECISE) {
            const sk_u32x4 fl = sk_frag_bits(sk_swap_frag(ql[0], ql[1]));
            *reinterpret_cast<sk_u32x4*>(my_os_lo + kc * 32) = fl;
            __builtin_amdgcn_raw_buffer_store_b128(fl, r_sl, voff_s + kc * 32, 0, 0);
          }
        }
      PS_T(4)
      if (last) break;
      LY = LN;
      tpc = tpcn;
    } else {
      // ---- chain output, fp32 [N, rows] with the caller's row stride ----
      const bool youtp = rout && p.y != nullptr;  // (a chain may be run for its saved planes only)
      const __amdgpu_buffer_rsrc_t ry = sk_rsrc(p.y ? p.y : p.x, N * p.ldy);
      const bool vecy = ((p.ldy & 3) == 0) && ((LY.rows & 3) == 0) && ((((uintptr_t)p.y) & 15) == 0);
#pragma unroll
      for (int nt = 0; nt < 4; nt++)
        if (nt < ntl) {
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int c0 = nt * 32 + 8 * g + 4 * half;
            float v[4];
            if (is_mask) {
              const sk_u32x2 m = __builtin_amdgcn_raw_buffer_load_b64(r_m, rin ? (int)((n * LY.mask_w + c0) * 2) : SK_OOB, 0, 0);
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const unsigned w = j < 2 ? m[0] : m[1];
                const float mv = sk_u2f((j & 1) ? (w & 0xffff0000u) : (w << 16));
                v[j] = acc[nt][4 * g + j] * act_grad(mv, ekind, p.slope) * p.out_scale;
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; j++) v[j] = apply_act(acc[nt][4 * g + j], ekind, p.slope) * p.out_scale;
            }
            if (vecy) {
              const sk_u32x4 q = {sk_f2u(v[0]), sk_f2u(v[1]), sk_f2u(v[2]), sk_f2u(v[3])};
              __builtin_amdgcn_raw_buffer_store_b128(q, ry, (youtp && c0 + 3 < LY.rows) ? (int)((n * p.ldy + c0) * 4) : SK_OOB, 0, 0);
            } else {
#pragma unroll
              for (int j = 0; j < 4; j++)
                __builtin_amdgcn_raw_buffer_store_b32(sk_f2u(v[j]), ry, (youtp && c0 + j < LY.rows) ? (int)((n * p.ldy + c0 + j) * 4) : SK_OOB, 0, 0);
            }
          }
        }
    }
  }
#ifdef PS_PROF
  PS_T(4)
  pacc_[5] = __builtin_readcyclecounter() - pstart_;
  if (blockIdx.x < 256 && lane == 0 && wave < 8) {
#pragma unroll
    for (int i = 0; i < 8; i++) ps_prof_buf[(blockIdx.x * 8 + wave) * 8 + i] = pacc_[i];
  }
  if (blockIdx.x < 1024 && tid == 0) { ps_prof_res[blockIdx.x * 2] = preal_; ps_prof_res[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

int pstack_plan(PsP& p, const PsLayer* host_layers, bool precise) {
  static int nw_env = -1;
  if (nw_env < 0) nw_env = crk_sw().ps_nw;
  p.hl = p.hr = 0;
  int max_kp = 16, max_rows = 32;
  for (int l = 0; l < p.L; l++) {
    const PsLayer& y = host_layers[l];
    const int o0 = y.off0, o1 = y.off0 + (y.k - 1) * y.dil;
    if (-o0 > SK_GUARD || o1 > SK_GUARD || o0 > 0 || o1 < 0) return CRK_ERR_UNSUPPORTED;
    if (y.rows_pad > 128 || y.kp > 128 || (y.rows_pad & 31) || (y.kp & 15) || y.k < 1 || y.k > 8) return CRK_ERR_UNSUPPORTED;
    if ((l + 1 < p.L || p.tail) && host_layers[l + 1].kp > y.rows_pad) return CRK_ERR_UNSUPPORTED;
    p.hl += -o0; p.hr += o1;
    if (y.kp > max_kp) max_kp = y.kp;
    if (y.rows_pad > max_rows) max_rows = y.rows_pad;
  }
  // 256-frame windows (8 waves, two per SIMD) measured ~2 % better than 128-frame ones for every chain
  // of the step, pointwise ones included; bf16x3 keeps 4 waves (LDS)
  p.nw = precise ? 4 : (nw_env == 4 || nw_env == 8 ? nw_env : 8);
  p.os = max_kp * 2 + 16;
  const int R = p.nw * 32;
  p.tmo = R - p.hl - p.hr;
  if (p.tmo < 32) return CRK_ERR_UNSUPPORTED;
  p.tiles_per_utt = ceil_div(p.T, p.tmo);
  p.tmo = ceil_div(p.T, p.tiles_per_utt);
  const int obytes = (SK_GUARD * 2 + R) * p.os;
  // chunk buffer: whole layers where they fit the staging registers and what LDS is left (two buffers; bf16x3 keeps
  // one hi + one lo), never less than one tap of the largest layer
  const int nl = p.L + (p.tail ? 1 : 0);
  const int misc = p.L * 128 * 4 + (p.L + 1) * (int)sizeof(PsLayer) + 64;
  static int lds_cap = -1;
  if (lds_cap < 0) lds_cap = 160;
  const int avail = (lds_cap * 1024 - (precise ? 2 : 1) * obytes - misc) / 2;
  const int cap = PS_MAXP(p.nw * 64) * p.nw * 64;
  int need = 0, one = 0;
  for (int l = 0; l < nl && l < p.L; l++) {
    const PsLayer& y = host_layers[l];
    const int tap_bytes = y.rows_pad * (y.kp * 2 + 16);
    if (y.rows_pad * (y.kp >> 3) > cap) return CRK_ERR_UNSUPPORTED;
    if (tap_bytes > one) one = tap_bytes;
    const int t = ps_tpc(y.k, y.rows_pad, y.kp, cap, avail);
    if (t * tap_bytes > need) need = t * tap_bytes;
  }
  if (one > avail) return CRK_ERR_UNSUPPORTED;
  p.w_bytes = (need + 15) & ~15;
  (void)max_rows;
  int off = obytes;
  p.o_olo = off; if (precise) off += obytes;
  p.o_whi = off; off += precise ? p.w_bytes : 2 * p.w_bytes;
  p.o_wlo = off; if (precise) off += p.w_bytes;
  p.o_bias = off; off += p.L * 128 * 4;
  p.o_tab = off; off += (p.L + 1) * (int)sizeof(PsLayer);
  p.lds_bytes = (off + 15) & ~15;
  {  // algorithmic bytes per frame: fp32 input row, fp32 output row, every kept operand plane, every mask plane read
    double b = 4.0 * p.cin + (p.y ? 4.0 * host_layers[p.L - 1].rows : 0.0);
    for (int l = 0; l < p.L; l++) {
      if (p.save_hi) b += 2.0 * host_layers[l].kp;
      if (host_layers[l].epi >= 3) b += 2.0 * host_layers[l].mask_w;
    }
    if (p.save_hi && p.tail) b += 2.0 * host_layers[p.L].kp;
    p.algo_bytes = b * (double)p.B * p.T;
  }
  return p.lds_bytes <= 160 * 1024 ? CRK_OK : CRK_ERR_UNSUPPORTED;
}

int launch_pstack(const PsP& p, bool precise, double flops, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[3] = {(const void*)pstack_kernel<true, 4>, (const void*)pstack_kernel<false, 4>, (const void*)pstack_kernel<false, 8>};
    for (int i = 0; i < 3; i++)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return CRK_ERR_HIP;
    attr_set = true;
  }
  dim3 grid(p.B * p.tiles_per_utt);
  conv_prof_bytes(4, p.algo_bytes);
  conv_prof_begin(4, flops, s);
  if (precise) hipLaunchKernelGGL((pstack_kernel<true, 4>), grid, dim3(256), p.lds_bytes, s, p);
  else if (p.nw == 4) hipLaunchKernelGGL((pstack_kernel<false, 4>), grid, dim3(256), p.lds_bytes, s, p);
  else hipLaunchKernelGGL((pstack_kernel<false, 8>), grid, dim3(512), p.lds_bytes, s, p);
  conv_prof_end(4, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// =====================================================================================
// Weight gradients of plain convs from bf16 planes: dW[tap][co][ci] = sum_t G[t][co] * O[t + off0 + tap*dil][ci]
// with G the output-gradient plane the backward chain stored and O the input-operand plane the
// forward chain stored.  Workgroup (g, l): conv l of the table, group g = a run of 64-frame
// chunks of the batch.  Rows travel HBM -> registers (next chunk, behind the MFMAs) -> row-major
// LDS tiles -> ds_read_b64_tr_b16 fragments (the reduction axis is the frame axis).  Four
// waves share the (tap, cin-band, cout-band) tiles of the conv, at most 8 per wave.
// Partial sums per group in the layout of the table kernel; wnorm_bwd_kernel reduces them.
#define PW_FR 64
#define PW_SPAN 32
#define PW_MAXT 8
// Phase cycles (tools/ps2_phase_cycles.py, -DPW_PROF): per workgroup (first 512 of a launch) and wave [0] set-up
// [1] barrier + tiles -> LDS + barrier [2] next requests [3] fragments + MFMAs [4] partial sums out [5] whole kernel
#ifdef PW_PROF
__device__ unsigned long long pw_prof_buf[512 * 4 * 8];
extern "C" int crk_debug_pw_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pw_prof_buf), sizeof(unsigned long long) * 512 * 4 * 8) == hipSuccess ? 0 : 2;
}
#define PW_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pacc_[i] += t_ - plast_; plast_ = t_; }
#else
#define PW_T(i)
#endif

// MAXT: (tap, cin band, cout band) tiles a wave may hold (accumulators: 16 VGPRs each).  8 covers the classifier's widest
// conv; the 1x1 convs around a gated stack need 2, and a kernel instantiated for 2 keeps twice the workgroups resident.
template <bool PRECISE, int MAXT = PW_MAXT, bool SA = false>
__device__ __forceinline__ void pstack_wgrad_body(const PwP& p, int g, int layer, unsigned char* smem) {
  const PwLayer LY = p.layers[layer];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef PW_PROF
  unsigned long long pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pstart_ = __builtin_readcyclecounter();
  unsigned long long plast_ = pstart_;
#endif
  const int wa32 = (LY.wa + 31) & ~31, wb32 = (LY.wb + 31) & ~31;
  const int RA = wa32 * 2 + 64, RB = wb32 * 2 + 64;
  const int span = (LY.k - 1) * LY.dil, brn = PW_FR + span;
  const int abytes = PW_FR * RA, bbytes = (PW_FR + PW_SPAN) * RB, plane = abytes + bbytes;
  unsigned char* at_hi = smem;
  unsigned char* bt_hi = smem + abytes;
  const sk_u32x4 Z4 = {0u, 0u, 0u, 0u};
  for (int i = tid; i < (PRECISE ? 2 : 1) * plane / 16; i += 256) reinterpret_cast<sk_u32x4*>(smem)[i] = Z4;

  const int ppa = LY.wa >> 3, ppb = LY.wb >> 3;  // 16-byte pieces per plane row
  const int na = PW_FR * ppa, nb = brn * ppb;    // pieces per chunk: <= 1024, <= 1536
  sk_u32x4 ra_h[4], ra_l[4], rb_h[6], rb_l[6];
  const int ncpu = (p.T + PW_FR - 1) / PW_FR;
  const int c_beg = g * p.cpg, c_end = min(p.B * ncpu, (g + 1) * p.cpg);
#define PW_FETCH(c)                                                                                  \
  {                                                                                                  \
    const int u_ = (c) / ncpu, f0_ = ((c) - u_ * ncpu) * PW_FR;                                       \
    const long nb_ = (long)u_ * p.T;                                                                 \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                  \
      const int idx = tid + u * 256, r = idx / ppa, cc = idx - r * ppa, t = f0_ + r;                  \
      const bool on = idx < na && t < p.T;                                                           \
      const long off = (nb_ + t) * LY.wa + cc * 8;                                                   \
      ra_h[u] = on ? *reinterpret_cast<const sk_u32x4*>(p.abase + LY.a_hi + off) : Z4;                         \
      if (PRECISE) ra_l[u] = on ? *reinterpret_cast<const sk_u32x4*>(p.abase + LY.a_lo + off) : Z4;            \
    }                                                                                                \
    _Pragma("unroll") for (int u = 0; u < 6; u++) {                                                  \
      const int idx = tid + u * 256, r = idx / ppb, cc = idx - r * ppb, t = f0_ + LY.off0 + r;        \
      const bool on = idx < nb && t >= 0 && t < p.T;                                                 \
      const long off = (nb_ + t) * LY.wb + cc * 8;                                                   \
      rb_h[u] = on ? *reinterpret_cast<const sk_u32x4*>(p.bbase + LY.b_hi + off) : Z4;                         \
      if (PRECISE) rb_l[u] = on ? *reinterpret_cast<const sk_u32x4*>(p.bbase + LY.b_lo + off) : Z4;            \
    }                                                                                                \
  }
#define PW_COMMIT()                                                                                  \
  {                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                  \
      const int idx = tid + u * 256, r = idx / ppa, cc = idx - r * ppa;                               \
      if (idx < na) {                                                                                \
        *reinterpret_cast<sk_u32x4*>(at_hi + r * RA + cc * 16) = ra_h[u];                             \
        if (PRECISE) *reinterpret_cast<sk_u32x4*>(at_hi + plane + r * RA + cc * 16) = ra_l[u];        \
      }                                                                                              \
    }                                                                                                \
    _Pragma("unroll") for (int u = 0; u < 6; u++) {                                                  \
      const int idx = tid + u * 256, r = idx / ppb, cc = idx - r * ppb;                               \
      if (idx < nb) {                                                                                \
        *reinterpret_cast<sk_u32x4*>(bt_hi + r * RB + cc * 16) = rb_h[u];                             \
        if (PRECISE) *reinterpret_cast<sk_u32x4*>(bt_hi + plane + r * RB + cc * 16) = rb_l[u];        \
      }                                                                                              \
    }                                                                                                \
  }

  // ---- this wave's tiles: j = wave + 4m -> (cout band ct, cin band it, tap) ----
  const int nct = (LY.ca + 31) >> 5, nit = (LY.cb + 31) >> 5, ntiles = nct * nit * LY.k;
  const int i15 = lane & 15, grp = lane >> 4;
  const int rowoff = (grp >> 1) * 8 + (i15 >> 2);
  const int coloff = (grp & 1) * 16 + (i15 & 3) * 4;
  const int half = lane >> 5, l31 = lane & 31;
  int a_off[MAXT], b_off[MAXT];
  f32x16 acc[MAXT];
#pragma unroll
  for (int m = 0; m < MAXT; m++) {
    const int j = wave + 4 * m < ntiles ? wave + 4 * m : (wave < ntiles ? wave : 0);  // (a tile the wave does not have: its first)
    const int ct = j % nct, it = (j / nct) % nit, tap = j / (nct * nit);
    a_off[m] = rowoff * RA + (ct * 32 + coloff) * 2;
    b_off[m] = (rowoff + tap * LY.dil) * RB + (it * 32 + coloff) * 2;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[m][i] = 0.f;
  }
  float bsum = 0.f;
  const bool bias_wave = wave < nct && LY.pb >= 0;  // tile m = 0 of waves 0..nct-1 is (ct = wave, it 0, tap 0)

  if constexpr (!PRECISE) {
    // ---- plain bf16: chunks requested TWO ahead, nothing of the loop under a branch.  One wave per SIMD is all the
    // accumulators allow (MAXT = 8: 128 of them), so a chunk's HBM round trip was exposed once per chunk behind a
    // one-deep prefetch (12 k cycles per 64-frame chunk for 670 instructions: rocprof 49.5 us for the classifier's eight
    // convs).  Two register sets alternate; every load is a buffer load whose offset is out of range where the piece,
    // its frame or the chunk does not exist (it returns 0: no branch around a load, so the wait counts stay exact -
    // behind a branch the wait-count pass has to drain the queue), and the piece -> (row, column) divisions are done
    // once per thread instead of twice per chunk.  Same products in the same order: bit-identical partial sums. ----
    const __amdgpu_buffer_rsrc_t rA = sk_rsrc16(p.abase + LY.a_hi, (long)p.B * p.T * LY.wa);
    const __amdgpu_buffer_rsrc_t rB = sk_rsrc16(p.bbase + LY.b_hi, (long)p.B * p.T * LY.wb);
    int la[4], ga[4], qa[4], lb[6], gb[6], qb[6];  // LDS byte offset (-1: no piece), plane byte offset from the chunk's row 0, row
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int idx = tid + u * 256, r = idx / ppa, cc = idx - r * ppa;
      const bool ok = idx < na;
      la[u] = ok ? r * RA + cc * 16 : -1; ga[u] = (r * LY.wa + cc * 8) * 2; qa[u] = ok ? r : (1 << 24);
    }
#pragma unroll
    for (int u = 0; u < 6; u++) {
      const int idx = tid + u * 256, r = idx / ppb, cc = idx - r * ppb;
      const bool ok = idx < nb;
      lb[u] = ok ? r * RB + cc * 16 : -1; gb[u] = (r * LY.wb + cc * 8) * 2; qb[u] = ok ? r : (1 << 24);
    }
    int fu = c_beg / ncpu, ff = (c_beg - fu * ncpu) * PW_FR, fc = c_beg;  // the chunk the next request is for
    sk_u32x4 sa0[4], sb0[6], sa1[4], sb1[6];
#define PW_FETCH2(SA, SB)                                                                                    \
  {                                                                                                          \
    const bool live_ = fc < c_end;                                                                           \
    const int basea_ = (int)(((long)fu * p.T + ff) * LY.wa * 2), baseb_ = (int)(((long)fu * p.T + ff + LY.off0) * LY.wb * 2); \
    const int hia_ = live_ ? p.T - ff : 0, lob_ = -(ff + LY.off0), hib_ = live_ ? p.T - ff - LY.off0 : lob_;   \
    _Pragma("unroll") for (int u = 0; u < 4; u++)                                                            \
      SA[u] = __builtin_amdgcn_raw_buffer_load_b128(rA, qa[u] < hia_ ? basea_ + ga[u] : SK_OOB, 0, 0);        \
    _Pragma("unroll") for (int u = 0; u < 6; u++)                                                            \
      SB[u] = __builtin_amdgcn_raw_buffer_load_b128(rB, (qb[u] >= lob_ && qb[u] < hib_) ? baseb_ + gb[u] : SK_OOB, 0, 0); \
    fc++; ff += PW_FR;                                                                                       \
    if (ff >= p.T) { ff = 0; fu++; }                                                                         \
  }
#define PW_COMMIT2(SA, SB)                                                                                   \
  {                                                                                                          \
    _Pragma("unroll") for (int u = 0; u < 4; u++)                                                            \
      if (la[u] >= 0) *reinterpret_cast<sk_u32x4*>(at_hi + la[u]) = SA[u];                                    \
    _Pragma("unroll") for (int u = 0; u < 6; u++)                                                            \
      if (lb[u] >= 0) *reinterpret_cast<sk_u32x4*>(bt_hi + lb[u]) = SB[u];                                    \
  }
    // ONE code path for all MAXT accumulators, no guard around any MFMA: a wave with fewer tiles repeats its first tile
    // into accumulators nobody reads (the kernel is instantiated for the tile count the launch needs).  Fragments of k
    // step kc + 1 are read before the MFMAs of step kc where the registers allow two sets.  (With a run-time guard per
    // tile every MFMA sat in a basic block of its own behind its two transposing LDS reads: ~150 cycles of exposed
    // latency per 34-cycle MFMA.)
// SA: every tile of the wave lies in ONE cout band (tile j = wave + 4 m, band j % nct: true whenever nct divides 4 - 1, 2 or 4
// bands, i.e. every conv but the ones with 65 - 96 output channels): the dY fragment of a k step is then the same for all of
// them and is read once instead of once per tile (two transposing LDS reads per fragment: 12 instead of 20 LDS instructions
// per k step at five tiles - the reads, not the MFMAs, are what a k step takes).
#define PW_COMPUTE(SA)                                                                                       \
  {                                                                                                          \
    constexpr int NB = MAXT <= 4 ? 2 : 1, NA = (SA) ? 1 : MAXT;                                              \
    bf16x8 fa[NB][NA], fb[NB][MAXT];                                                                         \
    _Pragma("unroll") for (int m = 0; m < MAXT; m++) {                                                       \
      if (m < NA) fa[0][m] = sw_tr_frag(at_hi + a_off[m], RA);                                               \
      fb[0][m] = sw_tr_frag(bt_hi + b_off[m], RB);                                                           \
    }                                                                                                        \
    _Pragma("unroll") for (int kc = 0; kc < PW_FR / 16; kc++) {                                              \
      if (NB == 2 && kc + 1 < PW_FR / 16) {                                                                  \
        _Pragma("unroll") for (int m = 0; m < MAXT; m++) {                                                   \
          if (m < NA) fa[(kc + 1) % NB][m] = sw_tr_frag(at_hi + a_off[m] + (kc + 1) * 16 * RA, RA);          \
          fb[(kc + 1) % NB][m] = sw_tr_frag(bt_hi + b_off[m] + (kc + 1) * 16 * RB, RB);                      \
        }                                                                                                    \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      _Pragma("unroll") for (int m = 0; m < MAXT; m++) acc[m] = mfma_bf16(fa[kc % NB][(SA) ? 0 : m], fb[kc % NB][m], acc[m]); \
      if (bias_wave) bsum += sw_sum8(fa[kc % NB][0]);                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      if (NB == 1 && kc + 1 < PW_FR / 16) {                                                                  \
        _Pragma("unroll") for (int m = 0; m < MAXT; m++) {                                                   \
          if (m < NA) fa[0][m] = sw_tr_frag(at_hi + a_off[m] + (kc + 1) * 16 * RA, RA);                      \
          fb[0][m] = sw_tr_frag(bt_hi + b_off[m] + (kc + 1) * 16 * RB, RB);                                  \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
  }
    PW_FETCH2(sa0, sb0)
    PW_FETCH2(sa1, sb1)
    PW_T(0)
    // (an odd run ends with one chunk of zeros: every MFMA of it adds 0)
    for (int c = c_beg; c < c_end; c += 2) {
      __syncthreads();  // previous chunk's fragments consumed (first pass: tiles zeroed)
      PW_COMMIT2(sa0, sb0)
      __syncthreads();
      PW_T(1)
      PW_FETCH2(sa0, sb0)
      PW_T(2)
      PW_COMPUTE(SA)
      PW_T(3)
      __syncthreads();
      PW_COMMIT2(sa1, sb1)
      __syncthreads();
      PW_T(1)
      PW_FETCH2(sa1, sb1)
      PW_T(2)
      PW_COMPUTE(SA)
      PW_T(3)
    }
#undef PW_FETCH2
#undef PW_COMMIT2
#undef PW_COMPUTE
  }
  if (PRECISE && c_end > c_beg) PW_FETCH(c_beg)
  for (int c = c_beg; PRECISE && c < c_end; c++) {
    __syncthreads();  // previous chunk's fragments consumed (first pass: tiles zeroed)
    PW_COMMIT()
    __syncthreads();
    if (c + 1 < c_end) PW_FETCH(c + 1)
#pragma unroll
    for (int kc = 0; kc < PW_FR / 16; kc++) {
#pragma unroll
      for (int m = 0; m < MAXT; m++) {
        if (wave + 4 * m < ntiles) {
          const bf16x8 a_hi = sw_tr_frag(at_hi + a_off[m] + kc * 16 * RA, RA);
          const bf16x8 b_hi = sw_tr_frag(bt_hi + b_off[m] + kc * 16 * RB, RB);
          acc[m] = mfma_bf16(a_hi, b_hi, acc[m]);
          if (m == 0 && bias_wave) bsum += sw_sum8(a_hi);
          if (PRECISE) {
            const bf16x8 a_lo = sw_tr_frag(at_hi + plane + a_off[m] + kc * 16 * RA, RA);
            const bf16x8 b_lo = sw_tr_frag(bt_hi + plane + b_off[m] + kc * 16 * RB, RB);
            acc[m] = mfma_bf16(a_lo, b_hi, acc[m]);
            acc[m] = mfma_bf16(a_hi, b_lo, acc[m]);
            if (m == 0 && bias_wave) bsum += sw_sum8(a_lo);
          }
        }
      }
    }
  }

#pragma unroll
  for (int m = 0; m < MAXT; m++) {
    const int j = wave + 4 * m;
    if (j < ntiles) {
      const int ct = j % nct, it = (j / nct) % nit, tap = j / (nct * nit);
      const int ci = it * 32 + l31;
      float* out = p.partials + LY.pt + ((long)g * LY.k + tap) * LY.ca * LY.cb;
      if (ci < LY.cb) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int co = ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
          if (co < LY.ca) out[(long)co * LY.cb + ci] = acc[m][i];
        }
      }
    }
  }
  if (bias_wave) {
    const float tot = bsum + __shfl_xor(bsum, 32);
    const int co = wave * 32 + l31;
    if (half == 0 && co < LY.ca) p.partials[LY.pb + (long)g * LY.ca + co] = tot;
  }
#ifdef PW_PROF
  PW_T(4)
  pacc_[5] = __builtin_readcyclecounter() - pstart_;
  {
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < 512 && lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) pw_prof_buf[(wg * 4 + wave) * 8 + i] = pacc_[i];
    }
  }
#endif
}

template <bool PRECISE, int MAXT = PW_MAXT>
__global__ __launch_bounds__(256) void pstack_wgrad_kernel(const PwP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (!PRECISE && MAXT > 5) {
    // the body for the tile count of THIS workgroup's conv (the classifier's first conv has 30 tiles, its others 20: with one
    // body for the widest, 3 of every 8 MFMAs of seven of its eight convs were idle repeats)
    const PwLayer& Y = p.layers[blockIdx.y];
    const int per = (((Y.ca + 31) >> 5) * ((Y.cb + 31) >> 5) * Y.k + 3) >> 2;
    const bool sa = (4 % ((Y.ca + 31) >> 5)) == 0;  // one cout band per wave (pstack_wgrad_body, PW_COMPUTE)
    if (per <= 3) { if (sa) pstack_wgrad_body<PRECISE, 3, true>(p, blockIdx.x, blockIdx.y, smem); else pstack_wgrad_body<PRECISE, 3>(p, blockIdx.x, blockIdx.y, smem); return; }
    if (per <= 5) { if (sa) pstack_wgrad_body<PRECISE, 5, true>(p, blockIdx.x, blockIdx.y, smem); else pstack_wgrad_body<PRECISE, 5>(p, blockIdx.x, blockIdx.y, smem); return; }
  }
  if (!PRECISE) {
    const PwLayer& Y = p.layers[blockIdx.y];
    if ((4 % ((Y.ca + 31) >> 5)) == 0) { pstack_wgrad_body<PRECISE, MAXT, !PRECISE>(p, blockIdx.x, blockIdx.y, smem); return; }
  }
  pstack_wgrad_body<PRECISE, MAXT>(p, blockIdx.x, blockIdx.y, smem);
}
// the plain convs of several nets (first conv and head of every generator stack) in one launch: grid row y belongs to the
// net whose layer range holds it
template <int MAXT>
__global__ __launch_bounds__(256) void pstack_wgrad_multi_kernel(const PwMP m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int r = 0;
  while (r + 1 < m.n && (int)blockIdx.y >= m.first[r + 1]) r++;
  const PwP& p = m.q[r];
  if ((int)blockIdx.x >= p.G) return;
  // (the shared dY fragment of pstack_wgrad_kernel is not used here: two tiles per wave at most, 33.9 -> 35.4 us with it)
  pstack_wgrad_body<false, MAXT>(p, blockIdx.x, blockIdx.y - m.first[r], smem);
}
int launch_pstack_wgrad_multi(const PwMP& m, int total_layers, int max_G, int max_wa, int max_wb, int max_tiles, double flops,
                              double bytes, hipStream_t s) {
  const int RA = ((max_wa + 31) & ~31) * 2 + 64, RB = ((max_wb + 31) & ~31) * 2 + 64;
  const int lds = PW_FR * RA + (PW_FR + PW_SPAN) * RB;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)pstack_wgrad_multi_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)pstack_wgrad_multi_kernel<PW_MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return CRK_ERR_HIP;
    attr_set = true;
  }
  conv_prof_bytes(6, bytes);
  conv_prof_begin(6, flops, s);
  if (max_tiles <= 8) hipLaunchKernelGGL(pstack_wgrad_multi_kernel<2>, dim3(max_G, total_layers), dim3(256), lds, s, m);  // <= 2 per wave
  else hipLaunchKernelGGL(pstack_wgrad_multi_kernel<PW_MAXT>, dim3(max_G, total_layers), dim3(256), lds, s, m);
  conv_prof_end(6, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

int pstack_wgrad_supported(int ca, int cb, int wa, int wb, int k, int dil) {
  const int nct = (ca + 31) / 32, nit = (cb + 31) / 32;
  return nct * nit * k <= 4 * PW_MAXT && (k - 1) * dil <= PW_SPAN && wa <= 128 && wb <= 128 && (wa & 15) == 0 &&
         (wb & 15) == 0 && ca <= wa && cb <= wb;
}

// max_tiles: largest (tap, cin band, cout band) tile count of a layer of the table (0: unknown -> the widest instantiation)
int launch_pstack_wgrad(const PwP& p, int nlayers, int max_wa, int max_wb, bool precise, double flops, hipStream_t s, int max_tiles) {
  const int RA = ((max_wa + 31) & ~31) * 2 + 64, RB = ((max_wb + 31) & ~31) * 2 + 64;
  const int lds = (precise ? 2 : 1) * (PW_FR * RA + (PW_FR + PW_SPAN) * RB);
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[6] = {(const void*)pstack_wgrad_kernel<true>, (const void*)pstack_wgrad_kernel<false>, (const void*)pstack_wgrad_kernel<false, 3>,
                          (const void*)pstack_wgrad_kernel<false, 4>, (const void*)pstack_wgrad_kernel<false, 5>, (const void*)pstack_wgrad_kernel<false, 6>};
    for (int i = 0; i < 6; i++)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return CRK_ERR_HIP;
    attr_set = true;
  }
  dim3 grid(p.G, nlayers);
  conv_prof_bytes(6, 2.0 * (max_wa + max_wb) * (double)p.B * p.T * nlayers);  // upper bound: widest planes per conv
  conv_prof_begin(6, flops, s);
  // (plain bf16: every wave issues the MFMAs of `per` tiles unconditionally - the instantiation for the count needed)
  const int per = max_tiles > 0 ? (max_tiles + 3) / 4 : PW_MAXT;
  if (precise) hipLaunchKernelGGL(pstack_wgrad_kernel<true>, grid, dim3(256), lds, s, p);
  else if (per <= 3) hipLaunchKernelGGL((pstack_wgrad_kernel<false, 3>), grid, dim3(256), lds, s, p);
  else if (per == 4) hipLaunchKernelGGL((pstack_wgrad_kernel<false, 4>), grid, dim3(256), lds, s, p);
  else if (per == 5) hipLaunchKernelGGL((pstack_wgrad_kernel<false, 5>), grid, dim3(256), lds, s, p);
  else if (per == 6) hipLaunchKernelGGL((pstack_wgrad_kernel<false, 6>), grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL(pstack_wgrad_kernel<false>, grid, dim3(256), lds, s, p);
  conv_prof_end(6, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
