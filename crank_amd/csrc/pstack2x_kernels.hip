// Forward chains of plain Conv1d layers, channel-split, in SPLIT-OPERAND arithmetic ("bf16x3": w_hi x_hi + w_hi x_lo +
// w_lo x_hi per product, ~fp32 accuracy): the speaker classifier C and the speaker-adversarial net (parallel_wavegan
// ParallelWaveGANDiscriminator; call sites crank/bin/train.py:78-89, crank/net/module/spkradv.py:49-60)
// in the `bf16x3f` mode - forward losses within 1e-3 of the fp32 reference, backward in plain bf16 (pstack2_kernels.hip reads
// the hi planes written here: same tables, same layout).
//
// pstack2_kernel with three differences:
//   * both operand tiles exist as a hi and a lo tile (the ping-pong pair becomes four LDS tiles): windows of 192 rows (twelve
//     waves) where that fits 160 KB, 128 rows (eight waves) otherwise; a wave owns ONE frame tile of 32 rows and the output
//     tiles of its parity;
//   * a layer's weight fragments do not fit the register file twice (25 x (hi, lo) x 16 B per lane): they stream from L2
//     through a ring of PS2X_RING (hi, lo) pairs, requested PS2X_RING k-steps (3 MFMAs each) ahead, across tile and layer
//     boundaries;
//   * forward chains only (activation epilogues; the data-gradient chains of this mode are plain bf16).
// Accumulation order per output element: bias, taps ascending, k-steps ascending, within a k-step hi.hi, hi.lo, lo.hi - the
// order of pstack_kernel<PRECISE = true> (pstack_kernels.hip).
#include "conv_kernels.h"
#include "stack_common.h"

#define PS2X_RING 4
#define PS2X_MAXS 25

__device__ __forceinline__ int ps2x_rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long ps2x_rfl64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ PsLayer ps2x_uniform(const PsLayer* src) {
  const PsLayer y = *src;
  PsLayer r;
  r.w_off = ps2x_rfl64(y.w_off); r.b_off = ps2x_rfl64(y.b_off);
  r.rows = ps2x_rfl(y.rows); r.rows_pad = ps2x_rfl(y.rows_pad); r.kp = ps2x_rfl(y.kp);
  r.k = ps2x_rfl(y.k); r.dil = ps2x_rfl(y.dil); r.off0 = ps2x_rfl(y.off0);
  r.epi = ps2x_rfl(y.epi); r.mask_w = ps2x_rfl(y.mask_w);
  r.mask_plane = ps2x_rfl64(y.mask_plane); r.save_plane = ps2x_rfl64(y.save_plane); r.f_off = ps2x_rfl64(y.f_off);
  return r;
}

// The (KT taps, NKC k-steps) MFMAs of one output tile over the wave's frame tile.  The ring holds k-steps 0 .. PS2X_RING - 1
// on entry; a slot is refilled with the pair PS2X_RING steps ahead as soon as its MFMAs are issued.  xb: this lane's hi
// B-fragment address of (tap 0, k-step 0); lo at xb + lo_delta; tstride = dilation x row stride.
template <int KT, int NKC>
__device__ __forceinline__ void ps2x_mma(f32x16& acc, sk_u32x4 (&rh)[PS2X_RING], sk_u32x4 (&rl)[PS2X_RING],
                                         const __amdgpu_buffer_rsrc_t ra_h, const __amdgpu_buffer_rsrc_t ra_l, int lane16,
                                         const unsigned char* xb, int lo_delta, int tstride) {
  constexpr int NS = KT * NKC, D = 2;
  static_assert(NS <= PS2X_MAXS, "k-steps of a layer");
  bf16x8 bh[D + 1], bl[D + 1];
#define PS2X_ADDR(s) (xb + ((s) / NKC) * tstride + ((s) % NKC) * 32)
#pragma unroll
  for (int s = 0; s < D && s < NS; s++) { bh[s] = lds_frag(PS2X_ADDR(s)); bl[s] = lds_frag(PS2X_ADDR(s) + lo_delta); }
#pragma unroll
  for (int s = 0; s < NS; s++) {
    if (s + D < NS) { bh[(s + D) % (D + 1)] = lds_frag(PS2X_ADDR(s + D)); bl[(s + D) % (D + 1)] = lds_frag(PS2X_ADDR(s + D) + lo_delta); }
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 ah = __builtin_bit_cast(bf16x8, rh[s % PS2X_RING]), al = __builtin_bit_cast(bf16x8, rl[s % PS2X_RING]);
    acc = mfma_bf16(ah, bh[s % (D + 1)], acc);
    acc = mfma_bf16(ah, bl[s % (D + 1)], acc);
    acc = mfma_bf16(al, bh[s % (D + 1)], acc);
    if (s + PS2X_RING < NS) {
      rh[s % PS2X_RING] = __builtin_amdgcn_raw_buffer_load_b128(ra_h, lane16 + (s + PS2X_RING) * 1024, 0, 0);
      rl[s % PS2X_RING] = __builtin_amdgcn_raw_buffer_load_b128(ra_l, lane16 + (s + PS2X_RING) * 1024, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef PS2X_ADDR
}

template <int NP>  // frame parts of 32 rows: R = 32 NP rows, 2 NP waves (tile parity x frame part)
__global__ __launch_bounds__(NP * 128, 1) void pstack2x_kernel(const PsP p) {
  constexpr int R = NP * 32, NT = NP * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mtw = wave & 1, fp = wave >> 1;
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;
  const long nbase = (long)b * p.T, N = (long)p.B * p.T;
  PsLayer* lay_s = reinterpret_cast<PsLayer*>(smem + p.o_tab);
  float* bias_s = reinterpret_cast<float*>(smem + p.o_bias);
  // operand tiles: the input of layer l lives in buffer l & 1 (row strides os / os_b); each buffer = hi tile, then lo tile
  const int tileA = (SK_GUARD * 2 + R) * p.os, tileB = (SK_GUARD * 2 + R) * p.os_b;
  unsigned char* const buf0 = smem;
  unsigned char* const buf1 = smem + p.o_olo;

  // ---- the first PS2X_RING (hi, lo) weight-fragment pairs of this wave's first tile of layer 0 ----
  sk_u32x4 rh[PS2X_RING], rl[PS2X_RING];
  const int lane16 = lane * 16;
// the ring's first pairs of tile mt of a layer whose fragment copy starts at f_off (ns k-steps per tile); a tile the layer does
// not have reads zeros (range 0)
#define PS2X_PRIME(f_off, ns, mt, on)                                                                                  \
  {                                                                                                                    \
    const __amdgpu_buffer_rsrc_t h_ = sk_rsrc16(p.whi + (f_off) + (long)(mt) * (ns) * 512, (on) ? (long)(ns) * 512 : 0); \
    const __amdgpu_buffer_rsrc_t l_ = sk_rsrc16(p.wlo + (f_off) + (long)(mt) * (ns) * 512, (on) ? (long)(ns) * 512 : 0); \
    _Pragma("unroll") for (int s_ = 0; s_ < PS2X_RING; s_++) {                                                         \
      rh[s_] = __builtin_amdgcn_raw_buffer_load_b128(h_, lane16 + s_ * 1024, 0, 0);                                    \
      rl[s_] = __builtin_amdgcn_raw_buffer_load_b128(l_, lane16 + s_ * 1024, 0, 0);                                    \
    }                                                                                                                  \
  }
  PS2X_PRIME(p.l0_f_off, p.l0_k * (p.l0_kp >> 4), mtw, mtw < (p.l0_rows_pad >> 5))

  // ---- biases: table entry and parameter, two dependent loads, the first in front of the operand loads ----
  constexpr int BU = 4;
  long long bo[BU]; int brow[BU];
#pragma unroll
  for (int u = 0; u < BU; u++) {
    const int i = u * NT + tid;
    const PsLayer* Y = p.layers + ((i < p.L * 128) ? (i >> 7) : 0);
    bo[u] = Y->b_off; brow[u] = Y->rows;
  }

  // ---- layer-0 operand: fp32 rows -> act -> (hi, lo) bf16; every piece requested before anything waits ----
  constexpr int Q = 8;  // R * 32 pieces (kp = 128) / NT threads
  const int kp0 = p.l0_kp, ppr = kp0 >> 2;
  const bool vec = ((p.ldx & 3) == 0) && ((((uintptr_t)p.x) & 15) == 0);
  const __amdgpu_buffer_rsrc_t rx = sk_rsrc(p.x, N * p.ldx);
  const int xr0 = tid / ppr, xc0 = tid - xr0 * ppr, xdr = NT / ppr, xdc = NT - xdr * ppr;
  sk_u32x4 q[Q];
  {
    int row = xr0, col = xc0;
#pragma unroll
    for (int u = 0; u < Q; u++) {
      const int c4 = col * 4, t = t0 - p.hl + row;
      const bool rin = row < R && t >= 0 && t < p.T;
      const long n = nbase + t;
      if (vec && c4 + 3 < p.cin) {
        q[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, rin ? (int)((n * p.ldx + c4) * 4) : SK_OOB, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
          q[u][j] = __builtin_amdgcn_raw_buffer_load_b32(rx, (rin && c4 + j < p.cin) ? (int)((n * p.ldx + c4 + j) * 4) : SK_OOB, 0, 0);
      }
      row += xdr; col += xdc;
      if (col >= ppr) { col -= ppr; row++; }
    }
  }
  // ---- layer table -> LDS; guard rows of the four operand tiles ----
  {
    const int nl = p.L + (p.tail ? 1 : 0);
    for (int i = tid; i < nl * (int)(sizeof(PsLayer) / 4); i += NT)
      reinterpret_cast<int*>(lay_s)[i] = reinterpret_cast<const int*>(p.layers)[i];
    const sk_u32x4 z4 = {0u, 0u, 0u, 0u};
    const int ga = SK_GUARD * p.os / 16, gb = SK_GUARD * p.os_b / 16;
    for (int i = tid; i < ga; i += NT) {
      reinterpret_cast<sk_u32x4*>(buf0)[i] = z4;
      reinterpret_cast<sk_u32x4*>(buf0 + (SK_GUARD + R) * p.os)[i] = z4;
      reinterpret_cast<sk_u32x4*>(buf0 + tileA)[i] = z4;
      reinterpret_cast<sk_u32x4*>(buf0 + tileA + (SK_GUARD + R) * p.os)[i] = z4;
    }
    for (int i = tid; i < gb; i += NT) {
      reinterpret_cast<sk_u32x4*>(buf1)[i] = z4;
      reinterpret_cast<sk_u32x4*>(buf1 + (SK_GUARD + R) * p.os_b)[i] = z4;
      reinterpret_cast<sk_u32x4*>(buf1 + tileB)[i] = z4;
      reinterpret_cast<sk_u32x4*>(buf1 + tileB + (SK_GUARD + R) * p.os_b)[i] = z4;
    }
  }
  {
    float bv[BU];
#pragma unroll
    for (int u = 0; u < BU; u++) {
      const int i = u * NT + tid;
      bv[u] = (i < p.L * 128 && bo[u] >= 0 && (i & 127) < brow[u]) ? p.params[bo[u] + (i & 127)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < BU; u++)
      if (u * NT + tid < p.L * 128) bias_s[u * NT + tid] = bv[u];
    for (int i = BU * NT + tid; i < p.L * 128; i += NT) {
      const PsLayer* Y = p.layers + (i >> 7);
      bias_s[i] = (Y->b_off >= 0 && (i & 127) < Y->rows) ? p.params[Y->b_off + (i & 127)] : 0.f;
    }
  }
  {
    const __amdgpu_buffer_rsrc_t r_sh0 = sk_rsrc16(p.save_hi ? p.save_hi + p.l0_save_plane : (const uint16_t*)p.x, N * kp0);
    const float in_sc = p.in_num ? p.in_scale * (p.in_num[0] / p.in_den[1]) : p.in_scale;
    int row = xr0, col = xc0;
#pragma unroll
    for (int u = 0; u < Q; u++) {
      const int c4 = col * 4, t = t0 - p.hl + row;
      const bool rin = row < R && t >= 0 && t < p.T;
      if (row < R) {
        const bool rout = rin && row >= p.hl && row < p.hl + p.tmo;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = rin ? apply_act(sk_u2f(q[u][j]) * in_sc, p.in_act, p.slope) : 0.f;
        sk_u32x2 h, lo;
        sk_quad<true>(v[0], v[1], v[2], v[3], h, lo);
        *reinterpret_cast<sk_u32x2*>(buf0 + (SK_GUARD + row) * p.os + c4 * 2) = h;
        *reinterpret_cast<sk_u32x2*>(buf0 + tileA + (SK_GUARD + row) * p.os + c4 * 2) = lo;
        __builtin_amdgcn_raw_buffer_store_b64(h, r_sh0, (rout && p.save_hi) ? (int)(((nbase + t) * kp0 + c4) * 2) : SK_OOB, 0, 0);
      }
      row += xdr; col += xdc;
      if (col >= ppr) { col -= ppr; row++; }
    }
  }
  __syncthreads();

  const int row = fp * 32 + l31;  // this lane's frame
  const int tfr = t0 - p.hl + row;
  const bool rin = tfr >= 0 && tfr < p.T;
  const bool rout = rin && row >= p.hl && row < p.hl + p.tmo;
  for (int l = 0; l < p.L; l++) {
    const PsLayer LY = ps2x_uniform(lay_s + l);
    const int ntile = LY.rows_pad >> 5, nkc = LY.kp >> 4;
    const bool last = l + 1 == p.L;
    const bool fin = last && !p.tail;
    const PsLayer LN = fin ? LY : ps2x_uniform(lay_s + l + 1);
    const unsigned char* oc = (l & 1) ? buf1 : buf0;
    unsigned char* on = (l & 1) ? buf0 : buf1;
    const int osc = (l & 1) ? p.os_b : p.os, osn = (l & 1) ? p.os : p.os_b;
    const int lod_c = (l & 1) ? tileB : tileA, lod_n = (l & 1) ? tileA : tileB;
    const float eneg = LY.epi == ACT_LRELU ? p.slope : (LY.epi == ACT_RELU ? 0.f : 1.f);
    const unsigned char* xb = oc + (SK_GUARD + row + LY.off0) * osc + half * 16;
    const int ns = LY.k * nkc, nsn = LN.k * (LN.kp >> 4);
    bool primed_next = false;

    for (int mt = mtw; mt < ntile; mt += 2) {
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const sk_f32x4 bq = *reinterpret_cast<const sk_f32x4*>(bias_s + l * 128 + mt * 32 + 8 * g + 4 * half);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[4 * g + j] = bq[j];
      }
      {
        const __amdgpu_buffer_rsrc_t ra_h = sk_rsrc16(p.whi + LY.f_off + (long)mt * ns * 512, (long)ns * 512);
        const __amdgpu_buffer_rsrc_t ra_l = sk_rsrc16(p.wlo + LY.f_off + (long)mt * ns * 512, (long)ns * 512);
        const int ts = LY.dil * osc;
        if (LY.k == 5) {
          if (nkc == 4) ps2x_mma<5, 4>(acc, rh, rl, ra_h, ra_l, lane16, xb, lod_c, ts);
          else if (nkc == 5) ps2x_mma<5, 5>(acc, rh, rl, ra_h, ra_l, lane16, xb, lod_c, ts);
          else if (nkc == 3) ps2x_mma<5, 3>(acc, rh, rl, ra_h, ra_l, lane16, xb, lod_c, ts);
          else ps2x_mma<5, 1>(acc, rh, rl, ra_h, ra_l, lane16, xb, lod_c, ts);
        } else {
          if (nkc == 4) ps2x_mma<3, 4>(acc, rh, rl, ra_h, ra_l, lane16, xb, lod_c, ts);
          else if (nkc == 8) ps2x_mma<3, 8>(acc, rh, rl, ra_h, ra_l, lane16, xb, lod_c, ts);
          else ps2x_mma<3, 1>(acc, rh, rl, ra_h, ra_l, lane16, xb, lod_c, ts);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the ring's first pairs of this wave's next tile - of this layer, or of the next one - behind the MFMAs
      if (mt + 2 < ntile) PS2X_PRIME(LY.f_off, ns, mt + 2, true)
      else if (!last) { PS2X_PRIME(LN.f_off, nsn, mtw, mtw < (LN.rows_pad >> 5)) primed_next = true; }

      if (!fin) {
        // ---- epilogue: the tile's 32 channels of the next operand -> the other buffer's hi and lo tiles, hi plane ----
        const __amdgpu_buffer_rsrc_t r_sh = sk_rsrc16(p.save_hi ? p.save_hi + LN.save_plane : (const uint16_t*)p.x, N * LN.kp);
        const int nk2 = LN.kp >> 4;
        const int voff_s = (rout && p.save_hi) ? (int)(((nbase + tfr) * LN.kp + 8 * half) * 2) : SK_OOB;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
          const int kc = 2 * mt + kk;
          if (kc < nk2) {
            sk_u32x2 qh[2], ql[2];
#pragma unroll
            for (int gg = 0; gg < 2; gg++) {
              const int g = 2 * kk + gg;
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; j++) { const float a_ = acc[4 * g + j]; v[j] = a_ > 0.f ? a_ : a_ * eneg; }
#pragma unroll
              for (int j = 0; j < 4; j++) v[j] = rin ? v[j] : 0.f;
              sk_quad<true>(v[0], v[1], v[2], v[3], qh[gg], ql[gg]);
            }
            const sk_u32x4 fhb = sk_frag_bits(sk_swap_frag(qh[0], qh[1])), flb = sk_frag_bits(sk_swap_frag(ql[0], ql[1]));
            *reinterpret_cast<sk_u32x4*>(on + (SK_GUARD + row) * osn + 8 * half * 2 + kc * 32) = fhb;
            *reinterpret_cast<sk_u32x4*>(on + lod_n + (SK_GUARD + row) * osn + 8 * half * 2 + kc * 32) = flb;
            __builtin_amdgcn_raw_buffer_store_b128(fhb, r_sh, voff_s + kc * 32, 0, 0);
          }
        }
      } else {
        // ---- chain output, fp32 [N, rows] with the caller's row stride ----
        const __amdgpu_buffer_rsrc_t ry = sk_rsrc(p.y ? p.y : p.x, N * p.ldy);
        const bool vecy = ((p.ldy & 3) == 0) && ((LY.rows & 3) == 0) && ((((uintptr_t)p.y) & 15) == 0);
        const bool youtp = rout && p.y != nullptr;
        const long n = nbase + tfr;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int c0 = mt * 32 + 8 * g + 4 * half;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) { const float a_ = acc[4 * g + j]; v[j] = (a_ > 0.f ? a_ : a_ * eneg) * p.out_scale; }
          if (vecy) {
            const sk_u32x4 qv = {sk_f2u(v[0]), sk_f2u(v[1]), sk_f2u(v[2]), sk_f2u(v[3])};
            __builtin_amdgcn_raw_buffer_store_b128(qv, ry, (youtp && c0 + 3 < LY.rows) ? (int)((n * p.ldy + c0) * 4) : SK_OOB, 0, 0);
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
              __builtin_amdgcn_raw_buffer_store_b32(sk_f2u(v[j]), ry, (youtp && c0 + j < LY.rows) ? (int)((n * p.ldy + c0 + j) * 4) : SK_OOB, 0, 0);
          }
        }
      }
    }
    // a wave without a tile in this layer (32-channel layers: the odd waves) still needs its first pairs of the next one
    if (!primed_next && !last) PS2X_PRIME(LN.f_off, nsn, mtw, mtw < (LN.rows_pad >> 5))
    if (last) break;
    __syncthreads();  // the next operand tiles are complete; this layer's reads of the other buffer are done
  }
#undef PS2X_PRIME
}

int pstack2x_plan(PsP& p, const PsLayer* host_layers) {
  p.hl = p.hr = 0;
  const int nl = p.L + (p.tail ? 1 : 0);
  if (p.L < 1 || nl > 17) return CRK_ERR_UNSUPPORTED;
  int kp_a = 16, kp_b = 16;
  for (int l = 0; l < nl; l++) {
    const PsLayer& y = host_layers[l];
    if (y.kp > 128 || (y.kp & 15)) return CRK_ERR_UNSUPPORTED;
    if ((l & 1) ? y.kp > kp_b : y.kp > kp_a) ((l & 1) ? kp_b : kp_a) = y.kp;
    if (l >= p.L) break;
    const int o0 = y.off0, o1 = y.off0 + (y.k - 1) * y.dil;
    if (-o0 > SK_GUARD || o1 > SK_GUARD || o0 > 0 || o1 < 0) return CRK_ERR_UNSUPPORTED;
    const int nkc = y.kp >> 4;
    const bool shape_ok = (y.k == 5 && (nkc == 1 || nkc == 3 || nkc == 4 || nkc == 5)) || (y.k == 3 && (nkc == 1 || nkc == 4 || nkc == 8));
    if (y.f_off < 0 || y.rows_pad > 128 || (y.rows_pad & 31) || !shape_ok) return CRK_ERR_UNSUPPORTED;
    if (y.epi > 2) return CRK_ERR_UNSUPPORTED;  // forward chains only (no mask epilogues)
    if ((l + 1 < p.L || p.tail) && host_layers[l + 1].kp > y.rows_pad) return CRK_ERR_UNSUPPORTED;
    p.hl += -o0; p.hr += o1;
  }
  if ((p.cin + 3) / 4 * 4 > host_layers[0].kp) return CRK_ERR_UNSUPPORTED;
  p.l0_f_off = host_layers[0].f_off; p.l0_save_plane = host_layers[0].save_plane;
  p.l0_k = host_layers[0].k; p.l0_kp = host_layers[0].kp; p.l0_rows_pad = host_layers[0].rows_pad;
  p.os = kp_a * 2 + 16; p.os_b = kp_b * 2 + 16;
  // 192-row windows (twelve waves) where the four operand tiles fit the CU's LDS, 128-row ones (eight waves) otherwise
  for (int R = 192; R >= 128; R -= 64) {
    int off = 2 * (SK_GUARD * 2 + R) * p.os;
    const int o_olo = off;
    off += 2 * (SK_GUARD * 2 + R) * p.os_b;
    off = (off + 15) & ~15;
    const int o_bias = off; off += p.L * 128 * 4;
    const int o_tab = off; off += (p.L + 1) * (int)sizeof(PsLayer);
    const int tmo = R - p.hl - p.hr;
    if (off > 160 * 1024 || tmo < 32) continue;
    p.o_olo = o_olo; p.o_bias = o_bias; p.o_tab = o_tab; p.lds_bytes = (off + 15) & ~15;
    p.nw = R / 32 * 2;
    p.tiles_per_utt = ceil_div(p.T, tmo);
    p.tmo = ceil_div(p.T, p.tiles_per_utt);
    double bb = 4.0 * p.cin + (p.y ? 4.0 * host_layers[p.L - 1].rows : 0.0);
    for (int l = 0; l < p.L; l++)
      if (p.save_hi) bb += 2.0 * host_layers[l].kp;
    if (p.save_hi && p.tail) bb += 2.0 * host_layers[p.L].kp;
    p.algo_bytes = bb * (double)p.B * p.T;
    return CRK_OK;
  }
  return CRK_ERR_UNSUPPORTED;
}

int launch_pstack2x(const PsP& p, double flops, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)pstack2x_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)pstack2x_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return CRK_ERR_HIP;
    attr_set = true;
  }
  dim3 grid(p.B * p.tiles_per_utt);
  conv_prof_bytes(4, p.algo_bytes);
  conv_prof_begin(4, flops, s);
  if (p.nw == 8) hipLaunchKernelGGL((pstack2x_kernel<4>), grid, dim3(512), p.lds_bytes, s, p);
  else hipLaunchKernelGGL((pstack2x_kernel<6>), grid, dim3(768), p.lds_bytes, s, p);
  conv_prof_end(4, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
