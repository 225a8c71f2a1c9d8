// Chains of plain Conv1d layers, channel-split (round 3): the speaker classifier C and the speaker-adversarial net
// (parallel_wavegan ParallelWaveGANDiscriminator; call sites crank/bin/train.py:78-89,
// crank/net/module/spkradv.py:49-60) in either direction - the same chains, tables (PsLayer / PsP) and saved planes as
// pstack_kernels.hip, which stays for bf16x3, the 1x1 chains around the discriminator's gated stack and unusual shapes.
//
// pstack_kernel gives a wave 32 frames and every output channel: each MFMA needs its own weight fragment from LDS (three
// LDS reads per two MFMAs, one k-step of prefetch) and a layer costs a barrier per weight chunk - its phase cycles
// (profiles/round2_pstack_phase_cycles.txt) show 10 k cycles of MFMA in a 138 k-cycle workgroup life.  Here, as in
// stack2_kernels.hip, a wave owns ONE 32-channel output tile and two frame tiles of a 128- / 256-row window:
//   * the tile's weights - every (tap, 16-channel k-step) A fragment of the layer, <= 25 x 16 bytes per lane - come
//     straight from L2 in fragment order (weight_prep writes that copy: ConvEntry::fr_mode 6 / bfr_mode 1,
//     [tile][tap][k-step][lane]) into registers, requested a layer ahead behind the MFMAs of the running layer;
//   * the operand tile ping-pongs between two LDS buffers, so a layer is its MFMAs (fully unrolled per (taps, k-steps)
//     shape, B fragments three steps ahead), its epilogue and ONE barrier;
//   * the epilogue - LeakyReLU (forward) or x LeakyReLU'(saved plane) (data gradient), bf16, v_permlane32_swap into the
//     B-fragment layout, the next operand tile and the saved plane - is pstack_kernel's.
// The products are accumulated in the same order (bias, taps ascending, k-steps ascending): bit-identical results
// (tests/test_gpu_properties.py).  A workgroup is 2 x NFH waves (tile parity x window part).
#include "conv_kernels.h"
#include "stack_common.h"

#define PS2_MAXS 25  // A fragments of one (layer, tile): taps x kp / 16
// Phase cycles (tools/ps2_phase_cycles.py, -DPS2_PROF): per workgroup and wave [0] prologue [1] fragment wait + MFMAs
// [2] next fragments + epilogue [3] barrier [5] whole kernel
#ifdef PS2_PROF
__device__ unsigned long long ps2_prof_buf[512 * 4 * 8];
__device__ unsigned long long ps2_prof_res[1024 * 2];
extern "C" int crk_debug_ps2_prof(unsigned long long* out, unsigned long long* res) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ps2_prof_buf), sizeof(unsigned long long) * 512 * 4 * 8) != hipSuccess) return 2;
  return hipMemcpyFromSymbol(res, HIP_SYMBOL(ps2_prof_res), sizeof(unsigned long long) * 1024 * 2) == hipSuccess ? 0 : 2;
}
#define PS2_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pacc_[i] += t_ - plast_; plast_ = t_; }
#else
#define PS2_T(i)
#endif
#ifndef PS2_DEPTH
#define PS2_DEPTH 2  // B fragments are read this many steps ahead of their MFMAs (3 spills)
#endif

// (KT taps, NKC k-steps): the MFMAs of one output tile over FT frame tiles.  xb: this lane's B-fragment address of
// (tap 0, k-step 0, frame tile 0); tstride = dilation x row stride, fstride = 32 rows.
template <int KT, int NKC, int FT>
__device__ __forceinline__ void ps2_mma(f32x16 (&acc)[FT], const bf16x8 (&A)[PS2_MAXS], const unsigned char* xb, int tstride,
                                        int fstride) {
  constexpr int NS = KT * NKC, D = PS2_DEPTH;
  static_assert(NS <= PS2_MAXS, "fragment registers");
  bf16x8 bq[D + 1][FT];
#define PS2_ADDR(s, ft) (xb + ((s) / NKC) * tstride + (ft) * fstride + ((s) % NKC) * 32)
#pragma unroll
  for (int s = 0; s < D && s < NS; s++)
#pragma unroll
    for (int ft = 0; ft < FT; ft++) bq[s][ft] = lds_frag(PS2_ADDR(s, ft));
#pragma unroll
  for (int s = 0; s < NS; s++) {
    if (s + D < NS) {
#pragma unroll
      for (int ft = 0; ft < FT; ft++) bq[(s + D) % (D + 1)][ft] = lds_frag(PS2_ADDR(s + D, ft));
    }
    // (pinned: left alone the scheduler sinks every read to just in front of its MFMA - one fragment buffer, a full LDS
    // round trip per MFMA)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(A[s], bq[s % (D + 1)][ft], acc[ft]);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef PS2_ADDR
}
// a layer record from LDS with every field in scalar registers (uniform branches and addresses, not lane-wise ones; read
// with scalar loads from the global table instead, a record cost a scalar-cache round trip per layer: +15 %)
__device__ __forceinline__ int ps2_rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long ps2_rfl64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ PsLayer ps2_uniform(const PsLayer* src) {
  const PsLayer y = *src;
  PsLayer r;
  r.w_off = ps2_rfl64(y.w_off); r.b_off = ps2_rfl64(y.b_off);
  r.rows = ps2_rfl(y.rows); r.rows_pad = ps2_rfl(y.rows_pad); r.kp = ps2_rfl(y.kp);
  r.k = ps2_rfl(y.k); r.dil = ps2_rfl(y.dil); r.off0 = ps2_rfl(y.off0);
  r.epi = ps2_rfl(y.epi); r.mask_w = ps2_rfl(y.mask_w);
  r.mask_plane = ps2_rfl64(y.mask_plane); r.save_plane = ps2_rfl64(y.save_plane); r.f_off = ps2_rfl64(y.f_off);
  return r;
}
__host__ __device__ __forceinline__ bool ps2_shape_ok(int k, int nkc) {
  return (k == 5 && (nkc == 1 || nkc == 3 || nkc == 4 || nkc == 5)) || (k == 3 && (nkc == 1 || nkc == 4 || nkc == 8));
}

// VEC: the chain's input rows are whole 16-byte pieces (row stride and channel count multiples of 4 floats, 16-byte aligned
// base): one 16-byte request per piece.  A template parameter, not a branch: with both request loops in one function the
// wait-count pass drains the queue in front of whichever runs (the other loop's destination registers are "in flight" on
// the path through it) - the fragment and table requests then make their round trip BEFORE the window's rows are asked for.
template <int NFH, bool VEC>
__global__ __launch_bounds__(NFH * 128, NFH == 2 ? 2 : 1) void pstack2_kernel(const PsP p) {
  // two waves per SIMD either way (8-wave workgroups, or two 4-wave ones per CU): one wave's epilogue - VALU, LDS and
  // plane stores - runs under the other's MFMAs.  (Four waves per CU with four frame tiles each were measured first:
  // every stall exposed, no faster than pstack_kernel.)
  constexpr int FT = 2, R = NFH * FT * 32, NT = NFH * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  // (the wave index as a scalar: the weight-fragment buffer descriptors and the tile loop depend on it - lane-wise they
  // would need a waterfall loop per load)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mtw = wave & 1, fh = wave >> 1;
#ifdef PS2_PROF
  unsigned long long pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pstart_ = __builtin_readcyclecounter(), preal_ = __builtin_amdgcn_s_memrealtime();
  unsigned long long plast_ = pstart_;
#endif
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;
  const long nbase = (long)b * p.T, N = (long)p.B * p.T;
  PsLayer* lay_s = reinterpret_cast<PsLayer*>(smem + p.o_tab);
  float* bias_s = reinterpret_cast<float*>(smem + p.o_bias);
  // operand tiles: the input of layer l lives in buffer l & 1 (row strides os / os_b: the widest operand each ever holds)
  unsigned char* const buf0 = smem;
  unsigned char* const buf1 = smem + p.o_olo;

  // ---- this wave's weight fragments of layer 0 (requested first: the longest latency of the prologue) ----
  bf16x8 A[PS2_MAXS];
#define PS2_LOADA_RANGE(f_off, ns, mt, on, S0, S1)                                                                    \
  {                                                                                                                    \
    const __amdgpu_buffer_rsrc_t ra_ = sk_rsrc16(p.whi + (f_off) + (long)(mt) * (ns) * 512, (on) ? (long)(ns) * 512 : 0); \
    _Pragma("unroll") for (int s_ = (S0); s_ < (S1); s_++)                                                             \
      A[s_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ra_, lane * 16 + s_ * 1024, 0, 0));      \
  }
#define PS2_LOADA(f_off, ns, mt, on) PS2_LOADA_RANGE(f_off, ns, mt, on, 0, PS2_MAXS)
  // (layer 0: the first PS2_A0 fragments here, the rest behind the operand's way into LDS - all 25 next to the window's 16
  // pieces per thread and the bias requests are more than the 256 registers hold: a piece was spilled the moment it was
  // requested, i.e. behind a wait for EVERY request of the prologue)
  constexpr int PS2_A0 = 8;
  PS2_LOADA_RANGE(p.l0_f_off, p.l0_k * (p.l0_kp >> 4), mtw, mtw < (p.l0_rows_pad >> 5), 0, PS2_A0)

  // ---- biases -> LDS: two dependent loads (table, parameter), the first one in front of the operand loads below, so that
  // both round trips pass under the HBM latency of the operand (the memory counter retires in order) ----
  constexpr int PS2_BU = 4;  // biases per thread: PS_MAXL * 128 / NT at most 8 -> two rounds
  long long bo[PS2_BU]; int brow[PS2_BU];
#pragma unroll
  for (int u = 0; u < PS2_BU; u++) {
    const int i = u * NT + tid;
    const PsLayer* Y = p.layers + ((i < p.L * 128) ? (i >> 7) : 0);
    bo[u] = Y->b_off; brow[u] = Y->rows;
  }

  // ---- layer-0 operand: fp32 rows -> act -> bf16, the workgroup walks the window in 16-byte pieces (coalesced rows), every
  // piece requested before anything waits: with the fragments and biases ONE memory round trip in front of the first MFMA
  // (table -> barrier -> biases -> operand in two rounds, as first written, was five) ----
  constexpr int PS2_Q = 16;  // R * 32 pieces (kp = 128) / NT threads: the whole window in one round
  const int kp0 = p.l0_kp, ppr = kp0 >> 2;  // 4-channel pieces per row
  const __amdgpu_buffer_rsrc_t rx = sk_rsrc(p.x, N * p.ldx);
  const int total = R * ppr;
  // piece u of this thread: index tid + u * NT = (row, column) advanced without a division per piece
  const int xr0 = tid / ppr, xc0 = tid - xr0 * ppr, xdr = NT / ppr, xdc = NT - xdr * ppr;
  sk_u32x4 q[PS2_Q];
  if constexpr (VEC) {
    int row = xr0, col = xc0;
#pragma unroll
    for (int u = 0; u < PS2_Q; u++) {
      const int c4 = col * 4, t = t0 - p.hl + row;
      const bool on = row < R && t >= 0 && t < p.T && c4 < p.cin;
      q[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, on ? (int)(((nbase + t) * p.ldx + c4) * 4) : SK_OOB, 0, 0);
      row += xdr; col += xdc;
      if (col >= ppr) { col -= ppr; row++; }
    }
  } else {
    int row = xr0, col = xc0;
#pragma unroll
    for (int u = 0; u < PS2_Q; u++) {
      const int c4 = col * 4, t = t0 - p.hl + row;
      const bool rin = row < R && t >= 0 && t < p.T;
      const long n = nbase + t;
#pragma unroll
      for (int j = 0; j < 4; j++)
        q[u][j] = __builtin_amdgcn_raw_buffer_load_b32(rx, (rin && c4 + j < p.cin) ? (int)((n * p.ldx + c4 + j) * 4) : SK_OOB, 0, 0);
      row += xdr; col += xdc;
      if (col >= ppr) { col -= ppr; row++; }
    }
  }

  PS2_T(4)
  // ---- layer table -> LDS; guard rows of both operand tiles (every other row and every column a layer reads is written
  // by its producer) ----
  {
    const int nl = p.L + (p.tail ? 1 : 0);
    for (int i = tid; i < nl * (int)(sizeof(PsLayer) / 4); i += NT)
      reinterpret_cast<int*>(lay_s)[i] = reinterpret_cast<const int*>(p.layers)[i];
    const sk_u32x4 z4 = {0u, 0u, 0u, 0u};
    const int ga = SK_GUARD * p.os / 16, gb = SK_GUARD * p.os_b / 16;
    for (int i = tid; i < ga; i += NT) {
      reinterpret_cast<sk_u32x4*>(buf0)[i] = z4;
      reinterpret_cast<sk_u32x4*>(buf0 + (SK_GUARD + R) * p.os)[i] = z4;
    }
    for (int i = tid; i < gb; i += NT) {
      reinterpret_cast<sk_u32x4*>(buf1)[i] = z4;
      reinterpret_cast<sk_u32x4*>(buf1 + (SK_GUARD + R) * p.os_b)[i] = z4;
    }
  }
  PS2_T(6)
  {
    float bv[PS2_BU];
#pragma unroll
    for (int u = 0; u < PS2_BU; u++) {
      const int i = u * NT + tid;
      bv[u] = (i < p.L * 128 && bo[u] >= 0 && (i & 127) < brow[u]) ? p.params[bo[u] + (i & 127)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PS2_BU; u++)
      if (u * NT + tid < p.L * 128) bias_s[u * NT + tid] = bv[u];
    for (int i = PS2_BU * NT + tid; i < p.L * 128; i += NT) {  // (more than PS2_BU * NT / 128 layers)
      const PsLayer* Y = p.layers + (i >> 7);
      bias_s[i] = (Y->b_off >= 0 && (i & 127) < Y->rows) ? p.params[Y->b_off + (i & 127)] : 0.f;
    }
  }
  PS2_T(7)
  {
    const __amdgpu_buffer_rsrc_t r_sh0 = sk_rsrc16(p.save_hi ? p.save_hi + p.l0_save_plane : (const uint16_t*)p.x, N * kp0);
    // (a cross-entropy's  upstream gradient / count  folded into the chain's input: crk_net_backward_scaled)
    const float in_sc = p.in_num ? p.in_scale * (p.in_num[0] / p.in_den[1]) : p.in_scale;
    const bool in_relu = p.in_act == ACT_RELU;
    const float in_neg = p.in_act == ACT_LRELU ? p.slope : 1.f;
    int row = xr0, col = xc0;
#pragma unroll
    for (int u = 0; u < PS2_Q; u++) {
      const int c4 = col * 4, t = t0 - p.hl + row;
      const bool rin = row < R && t >= 0 && t < p.T;
      if (row < R) {
        const bool rout = rin && row >= p.hl && row < p.hl + p.tmo;
        // (the input activation as one select - the kind decided once, outside the 64 elements: as a three-way branch per
        // element it was 128 scalar branches in this loop.  Same values: v > 0 ? v : v * {1, slope} or 0)
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float a_ = sk_u2f(q[u][j]) * in_sc;
          const float neg_ = in_relu ? 0.f : a_ * in_neg;
          v[j] = rin ? (a_ > 0.f ? a_ : neg_) : 0.f;
        }
        const sk_u32x2 h = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<sk_u32x2*>(buf0 + (SK_GUARD + row) * p.os + c4 * 2) = h;
        __builtin_amdgcn_raw_buffer_store_b64(h, r_sh0, (rout && p.save_hi) ? (int)(((nbase + t) * kp0 + c4) * 2) : SK_OOB, 0, 0);
      }
      row += xdr; col += xdc;
      if (col >= ppr) { col -= ppr; row++; }
    }
  }
  PS2_LOADA_RANGE(p.l0_f_off, p.l0_k * (p.l0_kp >> 4), mtw, mtw < (p.l0_rows_pad >> 5), PS2_A0, PS2_MAXS)
  __syncthreads();
  PS2_T(0)

  const int row0 = fh * FT * 32 + l31;  // this lane's frame in the wave's first frame tile
  for (int l = 0; l < p.L; l++) {
    const PsLayer LY = ps2_uniform(lay_s + l);
    const int ntile = LY.rows_pad >> 5, nkc = LY.kp >> 4;
    const bool last = l + 1 == p.L;
    const bool fin = last && !p.tail;  // this layer's output is the chain's fp32 output
    const PsLayer LN = fin ? LY : ps2_uniform(lay_s + l + 1);
    const unsigned char* oc = (l & 1) ? buf1 : buf0;
    unsigned char* on = (l & 1) ? buf0 : buf1;
    const int osc = (l & 1) ? p.os_b : p.os, osn = (l & 1) ? p.os : p.os_b;
    const bool is_mask = LY.epi >= 3;
    const int ekind = is_mask ? LY.epi - 2 : LY.epi;  // ACT_NONE / ACT_RELU / ACT_LRELU
    // act(v) = v > 0 ? v : v * eneg, act'(side) = side > 0 ? 1 : eneg: apply_act / act_grad with the kind folded into a factor
    const float eneg = ekind == ACT_LRELU ? p.slope : (ekind == ACT_RELU ? 0.f : 1.f);
    const __amdgpu_buffer_rsrc_t r_m = sk_rsrc16(is_mask ? p.mask_hi + LY.mask_plane : (const uint16_t*)p.x, N * LY.mask_w);
    const unsigned char* xb = oc + (SK_GUARD + row0 + LY.off0) * osc + half * 16;
    const int nsn = LN.k * (LN.kp >> 4);
    bool loaded_next = false;

    for (int mt = mtw; mt < ntile; mt += 2) {
      // the activation-derivative planes of this tile's channels (data gradient), requested in front of the MFMAs
      sk_u32x2 mk[FT][4];
      if (is_mask) {
#pragma unroll
        for (int ft = 0; ft < FT; ft++) {
          const int t = t0 - p.hl + row0 + ft * 32;
          const bool rin = t >= 0 && t < p.T;
#pragma unroll
          for (int g = 0; g < 4; g++)
            mk[ft][g] = __builtin_amdgcn_raw_buffer_load_b64(r_m, rin ? (int)(((nbase + t) * LY.mask_w + mt * 32 + 8 * g + 4 * half) * 2) : SK_OOB, 0, 0);
        }
      }
      f32x16 acc[FT];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const sk_f32x4 bq = *reinterpret_cast<const sk_f32x4*>(bias_s + l * 128 + mt * 32 + 8 * g + 4 * half);
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[ft][4 * g + j] = bq[j];
      }
      const int ts = LY.dil * osc, fs = 32 * osc;
      if (LY.k == 5) {
        if (nkc == 4) ps2_mma<5, 4, FT>(acc, A, xb, ts, fs);
        else if (nkc == 5) ps2_mma<5, 5, FT>(acc, A, xb, ts, fs);
        else if (nkc == 3) ps2_mma<5, 3, FT>(acc, A, xb, ts, fs);
        else ps2_mma<5, 1, FT>(acc, A, xb, ts, fs);
      } else {
        if (nkc == 4) ps2_mma<3, 4, FT>(acc, A, xb, ts, fs);
        else if (nkc == 8) ps2_mma<3, 8, FT>(acc, A, xb, ts, fs);
        else ps2_mma<3, 1, FT>(acc, A, xb, ts, fs);
      }
      // the fragments of this wave's next tile - of this layer, or of the next one - behind the MFMAs that read A
      __builtin_amdgcn_sched_barrier(0);
      PS2_T(1)
      if (mt + 2 < ntile) PS2_LOADA(LY.f_off, LY.k * nkc, mt + 2, true)
      else if (!last) { PS2_LOADA(LN.f_off, nsn, mtw, mtw < (LN.rows_pad >> 5)) loaded_next = true; }

      if (!fin) {
        // ---- epilogue: the tile's 32 channels of the next operand -> the other LDS buffer and the saved plane ----
        const __amdgpu_buffer_rsrc_t r_sh = sk_rsrc16(p.save_hi ? p.save_hi + LN.save_plane : (const uint16_t*)p.x, N * LN.kp);
        const int nk2 = LN.kp >> 4;
#pragma unroll
        for (int ft = 0; ft < FT; ft++) {
          const int row = row0 + ft * 32, t = t0 - p.hl + row;
          const bool rin = t >= 0 && t < p.T;
          const bool rout = rin && row >= p.hl && row < p.hl + p.tmo;
          const int voff_s = (rout && p.save_hi) ? (int)(((nbase + t) * LN.kp + 8 * half) * 2) : SK_OOB;
#pragma unroll
          for (int kk = 0; kk < 2; kk++) {
            const int kc = 2 * mt + kk;
            if (kc < nk2) {
              sk_u32x2 qh[2], ql[2];
#pragma unroll
              for (int gg = 0; gg < 2; gg++) {
                const int g = 2 * kk + gg;
                float v[4];
                if (is_mask) {
#pragma unroll
                  for (int j = 0; j < 4; j++) {
                    const unsigned w = j < 2 ? mk[ft][g][0] : mk[ft][g][1];
                    const float mv = sk_u2f((j & 1) ? (w & 0xffff0000u) : (w << 16));
                    v[j] = acc[ft][4 * g + j] * (mv > 0.f ? 1.f : eneg);
                  }
                } else {
#pragma unroll
                  for (int j = 0; j < 4; j++) { const float a_ = acc[ft][4 * g + j]; v[j] = a_ > 0.f ? a_ : a_ * eneg; }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = rin ? v[j] : 0.f;
                sk_quad<false>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);
              }
              const sk_u32x4 fhb = sk_frag_bits(sk_swap_frag(qh[0], qh[1]));
              *reinterpret_cast<sk_u32x4*>(on + (SK_GUARD + row) * osn + 8 * half * 2 + kc * 32) = fhb;
              __builtin_amdgcn_raw_buffer_store_b128(fhb, r_sh, voff_s + kc * 32, 0, 0);
            }
          }
        }
      } else {
        // ---- chain output, fp32 [N, rows] with the caller's row stride ----
        const __amdgpu_buffer_rsrc_t ry = sk_rsrc(p.y ? p.y : p.x, N * p.ldy);
        const bool vecy = ((p.ldy & 3) == 0) && ((LY.rows & 3) == 0) && ((((uintptr_t)p.y) & 15) == 0);
#pragma unroll
        for (int ft = 0; ft < FT; ft++) {
          const int row = row0 + ft * 32, t = t0 - p.hl + row;
          const bool rin = t >= 0 && t < p.T;
          const bool youtp = rin && row >= p.hl && row < p.hl + p.tmo && p.y != nullptr;
          const long n = nbase + t;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int c0 = mt * 32 + 8 * g + 4 * half;
            float v[4];
            if (is_mask) {
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const unsigned w = j < 2 ? mk[ft][g][0] : mk[ft][g][1];
                const float mv = sk_u2f((j & 1) ? (w & 0xffff0000u) : (w << 16));
                v[j] = acc[ft][4 * g + j] * (mv > 0.f ? 1.f : eneg) * p.out_scale;
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; j++) { const float a_ = acc[ft][4 * g + j]; v[j] = (a_ > 0.f ? a_ : a_ * eneg) * p.out_scale; }
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
    // a wave without a tile in this layer (32-channel layers: the odd waves) still needs its fragments of the next one
    if (!loaded_next && !last) PS2_LOADA(LN.f_off, nsn, mtw, mtw < (LN.rows_pad >> 5))
    PS2_T(2)
    if (last) break;
    __syncthreads();  // the next operand tile is complete; this layer's reads of the other buffer are done
    PS2_T(3)
  }
#undef PS2_LOADA
#undef PS2_LOADA_RANGE
#ifdef PS2_PROF
  pacc_[5] = __builtin_readcyclecounter() - pstart_;
  if (blockIdx.x < 512 && lane == 0 && wave < 4) {
#pragma unroll
    for (int i = 0; i < 8; i++) ps2_prof_buf[(blockIdx.x * 4 + wave) * 8 + i] = pacc_[i];
  }
  if (blockIdx.x < 1024 && tid == 0) { ps2_prof_res[blockIdx.x * 2] = preal_; ps2_prof_res[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

int pstack2_plan(PsP& p, const PsLayer* host_layers) {
  p.hl = p.hr = 0;
  const int nl = p.L + (p.tail ? 1 : 0);
  if (p.L < 1 || nl > 17) return CRK_ERR_UNSUPPORTED;  // (PS_MAXL + 1 table entries)
  int kp_a = 16, kp_b = 16;
  for (int l = 0; l < nl; l++) {
    const PsLayer& y = host_layers[l];
    if (y.kp > 128 || (y.kp & 15)) return CRK_ERR_UNSUPPORTED;
    if ((l & 1) ? y.kp > kp_b : y.kp > kp_a) ((l & 1) ? kp_b : kp_a) = y.kp;
    if (l >= p.L) break;
    const int o0 = y.off0, o1 = y.off0 + (y.k - 1) * y.dil;
    if (-o0 > SK_GUARD || o1 > SK_GUARD || o0 > 0 || o1 < 0) return CRK_ERR_UNSUPPORTED;
    if (y.f_off < 0 || y.rows_pad > 128 || (y.rows_pad & 31) || !ps2_shape_ok(y.k, y.kp >> 4)) return CRK_ERR_UNSUPPORTED;
    if ((l + 1 < p.L || p.tail) && host_layers[l + 1].kp > y.rows_pad) return CRK_ERR_UNSUPPORTED;
    p.hl += -o0; p.hr += o1;
  }
  if ((p.cin + 3) / 4 * 4 > host_layers[0].kp) return CRK_ERR_UNSUPPORTED;
  p.l0_f_off = host_layers[0].f_off; p.l0_save_plane = host_layers[0].save_plane;
  p.l0_k = host_layers[0].k; p.l0_kp = host_layers[0].kp; p.l0_rows_pad = host_layers[0].rows_pad;
  // 128-row windows (4 waves, two workgroups per CU) where the halo is small, 256-row ones (8 waves) otherwise
  const int R = (p.hl + p.hr <= 32) ? 128 : 256;
  p.nw = R / 32;
  p.tmo = R - p.hl - p.hr;
  if (p.tmo < 32) return CRK_ERR_UNSUPPORTED;
  p.tiles_per_utt = ceil_div(p.T, p.tmo);
  p.tmo = ceil_div(p.T, p.tiles_per_utt);
  p.os = kp_a * 2 + 16; p.os_b = kp_b * 2 + 16;
  int off = (SK_GUARD * 2 + R) * p.os;
  p.o_olo = off; off += (SK_GUARD * 2 + R) * p.os_b;
  off = (off + 15) & ~15;
  p.o_bias = off; off += p.L * 128 * 4;
  p.o_tab = off; off += (p.L + 1) * (int)sizeof(PsLayer);
  p.lds_bytes = (off + 15) & ~15;
  {  // algorithmic bytes per frame (as pstack_plan)
    double bb = 4.0 * p.cin + (p.y ? 4.0 * host_layers[p.L - 1].rows : 0.0);
    for (int l = 0; l < p.L; l++) {
      if (p.save_hi) bb += 2.0 * host_layers[l].kp;
      if (host_layers[l].epi >= 3) bb += 2.0 * host_layers[l].mask_w;
    }
    if (p.save_hi && p.tail) bb += 2.0 * host_layers[p.L].kp;
    p.algo_bytes = bb * (double)p.B * p.T;
  }
  return p.lds_bytes <= (R == 128 ? 80 : 160) * 1024 ? CRK_OK : CRK_ERR_UNSUPPORTED;
}

int launch_pstack2(const PsP& p, double flops, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[4] = {(const void*)pstack2_kernel<2, true>, (const void*)pstack2_kernel<2, false>, (const void*)pstack2_kernel<4, true>,
                          (const void*)pstack2_kernel<4, false>};
    for (int i = 0; i < 4; i++)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return CRK_ERR_HIP;
    attr_set = true;
  }
  dim3 grid(p.B * p.tiles_per_utt);
  conv_prof_bytes(4, p.algo_bytes);
  conv_prof_begin(4, flops, s);
  // whole 16-byte pieces per input row (pstack2_kernel<.., VEC>)
  const bool vec = ((p.ldx & 3) == 0) && ((p.cin & 3) == 0) && ((((uintptr_t)p.x) & 15) == 0);
  if (p.nw == 4) {
    if (vec) hipLaunchKernelGGL((pstack2_kernel<2, true>), grid, dim3(256), p.lds_bytes, s, p);
    else hipLaunchKernelGGL((pstack2_kernel<2, false>), grid, dim3(256), p.lds_bytes, s, p);
  } else {
    if (vec) hipLaunchKernelGGL((pstack2_kernel<4, true>), grid, dim3(512), p.lds_bytes, s, p);
    else hipLaunchKernelGGL((pstack2_kernel<4, false>), grid, dim3(512), p.lds_bytes, s, p);
  }
  conv_prof_end(4, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
