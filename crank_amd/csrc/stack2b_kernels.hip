// Channel-split data-gradient chain of ALL gated residual blocks of a generator stack (plain-bf16 arithmetic), with the
// head's data gradient in front and the first conv's behind it - the backward counterpart of stack2_fwd_kernel.
//
// Same arithmetic, same planes and the same summation order per output element as stack_bwd_kernel<.., FOLD = true>
// (stack_kernels.hip; reference: autograd of parallel_wavegan's ResidualBlock chain, call sites
// crank/net/module/vqvae2.py:237-273): the results are bit-identical (tests/test_gpu_properties.py).  What differs is
// who computes what:
//
//   stack_bwd_kernel  : a wave owns 32 FRAMES and all channels; every weight chunk (64 x 128) is staged through LDS for
//                       the whole workgroup behind a barrier of its own (7 per block for k = 5), every wave reads every
//                       weight fragment from LDS, windows are 256 rows (T = 500: 3 windows per utterance = 192
//                       workgroups for 256 CUs, 167 useful rows of 256).
//   here              : a wave owns 32 output CHANNELS (tile mt = wave & 1 of the 64) of half the window (fh = wave >> 1),
//                       FT tiles of 32 frames each, one wave per SIMD.  Its weights - one A fragment per (chunk, k step) -
//                       go from L2 straight into registers in fragment order (weight_prep writes that layout,
//                       ConvEntry::bfr_off), feed FT MFMAs each and never touch LDS; a register ring keeps S2B_RING
//                       fragments in flight across chunk and block boundaries.  Two barriers per block (dG tile complete /
//                       next 1x1 operand complete).  Windows are 64 * FT rows: T = 500 runs as 4 windows of 192 rows on
//                       256 workgroups - every CU busy, a quarter fewer rows per CU.
//
// Per block l (last first), wave (mt, fh), frame tile ft:
//   P1  dz[32 mt..] = Wos^T[mt] . [sqrt(.5) dX_{l+1} | dS]      8 k steps: dS half from registers, dX half from the LDS tile
//       gate backward on its 32 z channels (tanh / sigmoid planes requested a phase ahead) -> dG (tanh-side and
//       sigmoid-side halves) as bf16: LDS tile [rows][128] for the taps, HBM plane for the weight gradient
//   P2  dXc[32 mt..] = sum_tap Wconv^T[tap][mt] . dG[t + off]   8 k steps per tap; (+ conditioning gradient, 8 k steps)
//       dX_l = sqrt(.5) dX_{l+1} + dXc -> fp32 registers (next block), bf16 plane (weight gradient), bf16 x sqrt(.5) LDS tile
#include "conv_kernels.h"

#include "stack_common.h"

// Phase-cycle instrumentation (tools/s2b_phase_cycles.py builds a second library with -DS2B_PROF): per workgroup and wave the
// shader cycles in [0] prologue (head) [1] phase 1 (1x1 + gate backward) [2] wait at barrier A [3] taps [4] dX epilogue
// [5] wait at barrier B [6] first-conv epilogue [7] whole kernel
#ifdef S2B_PROF
__device__ unsigned long long s2b_prof_buf[256 * 4 * 12];
extern "C" int crk_debug_s2b_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(s2b_prof_buf), sizeof(unsigned long long) * 256 * 4 * 12) == hipSuccess ? 0 : 2;
}
#define S2B_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); pacc_[i] += now_ - plast_; plast_ = now_; }
#else
#define S2B_T(i)
#endif
#define S2B_GS 272   // row stride of the dG tile: 128 bf16 + 16 B pad (conflict-free ds_read_b128)
// Ablation builds (tools/s2b_ablate.sh; timing only, results are wrong): S2B_ABL bit 0 no gate arithmetic, bit 1 no MFMAs,
// bit 2 no LDS fragment reads, bit 3 no weight loads, bit 4 no gate-plane loads, bit 5 no plane stores (= CRK_S2B_DBG=1)
#ifndef S2B_ABL
#define S2B_ABL 0
#endif
#if S2B_ABL & 2
#define mfma_bf16(a, b, c) s2b_fake_mfma(a, b, c)
__device__ __forceinline__ f32x16 s2b_fake_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  asm volatile("" ::"v"(a), "v"(b));
  return c;
}
#endif
#if S2B_ABL & 4
#define lds_frag(p) s2b_fake_frag(p)
__device__ __forceinline__ bf16x8 s2b_fake_frag(const unsigned char* p) {
  const unsigned v = (unsigned)(size_t)p;
  const sk_u32x4 q = {v, v, v, v};
  return __builtin_bit_cast(bf16x8, q);
}
#endif

// FOLD (generator stacks): the head's data gradient in front of the chain (dy -> dS) and the first conv's behind it.
// !FOLD (the discriminator; round 4): dS arrives as an fp32 plane from the head's own backward launch, dX_0 leaves as an
// fp32 plane for the first conv's (x LeakyReLU'(X_0), StackBP::mask_l0), and the conv input of every block went through a
// dropout mask that is regenerated here (StackBP::drop_p; the same hash as the forward).
// The body of one wave: FT = the frame tiles THIS wave owns, R = the rows of the window, NW = the waves of the workgroup, rb =
// the first row of the wave's frame part.  Round 4 tried eight waves, two per SIMD, with frame parts of (2, 2, 1, 1) tiles
// (waves w and w + 4 share a SIMD: three tiles per SIMD as before) because the SQ counters of the four-wave chain
// (profiles/round4_pmc_sq_stacks.txt) show a wave issue-stalled 51 % of its life with the matrix pipe 21 % busy - it is
// slower (see stack2_bwd_plan) and off by default.
template <int KT, bool AUX, int FT, int R, int NW, bool FOLD>
__device__ __forceinline__ void s2b_wave(const StackBP& p, unsigned char* smem, const int rb) {
  constexpr int GS = S2B_GS, XS = SK_XS, NT = NW * 64;
  // Weight fragments of the tap phase in flight: half a block (k = 5: 20 of 40 k steps, k = 3: 12 of 24; with conditioning a
  // third, 16 of 48).  The ring must divide the k steps of a block (slot = step % S2B_RING holds across blocks).  Deep, because
  // every CU of an XCD asks L2 for the same lines at the same time (~250 cycles per fragment when they all do): the stream
  // has to spread over the 1x1 / gate / epilogue time as well.  Not deeper, because the memory counter holds 63 operations:
  // ring + 1x1 fragments + gate planes + the block's 18 plane stores must fit, or every wait of the phase degenerates into
  // "wait for the oldest" (a whole-block ring, 40 fragments, measured slower than 8).
  // (eight waves, 256 registers each: a one-tile wave issues an MFMA per fragment and needs the deep ring, a two-tile wave has
  // the registers for a shorter one)
  constexpr int S2B_NS = KT * 8 + (AUX ? 8 : 0);  // k steps of a block: 24, 32, 40, 48
  constexpr int S2B_RING = (NW == 8 && FT == 2) ? (S2B_NS == 24 ? 12 : S2B_NS == 32 ? 8 : S2B_NS == 40 ? 10 : 12)
                                                : (S2B_NS == 24 ? 12 : S2B_NS == 40 ? 20 : 16);
  static_assert((KT * 8 + (AUX ? 8 : 0)) % S2B_RING == 0, "the ring must divide the k steps of a block");
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 1;
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * p.tmo;
  const long nbase = (long)b * p.T;
  const long N = (long)p.B * p.T, P = N * 64;

#ifdef S2B_PROF
  unsigned long long pacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, plast_ = __builtin_readcyclecounter();
  const unsigned long long pstart_ = plast_;
#endif
  unsigned char* gs = smem;            // [SK_GUARD + R + SK_GUARD][GS] dG_l (prologue: scratch for the dS exchange)
  unsigned char* xt = smem + p.o_dx;   // [R][XS] sqrt(.5) dX_{l+1} as the 1x1's operand (prologue: G1; epilogue: dX_0)
  unsigned char* dst = xt + R * XS;    // [R][XS] bf16 dS: the other half of the 1x1's operand, the same for every block

  int row[FT], voff_in[FT], voff_b[FT], voff_gb[FT], voff_r[FT];  // voff_r: the [N,64] planes the weight gradient reads (dX_l, dS)
  bool rin[FT], rout[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ft++) {
    row[ft] = rb + ft * 32 + l31;
    const int t = t0 - p.hl + row[ft];
    rin[ft] = t >= 0 && t < p.T;
    rout[ft] = rin[ft] && row[ft] >= p.hl && row[ft] < p.hl + p.tmo;
    voff_in[ft] = rin[ft] ? (int)(((nbase + t) * 64) * 2) : SK_OOB;    // bf16 [N,64] planes: byte offset of channel 0
    voff_b[ft] = (rout[ft] && !(p.dbg & 1) && !(S2B_ABL & 32)) ? voff_in[ft] : SK_OOB;
    voff_gb[ft] = (rout[ft] && !(p.dbg & 1) && !(S2B_ABL & 32)) ? (int)(((nbase + t) * 128) * 2) : SK_OOB;  // bf16 [N,128] dG planes
    if (p.rec) {  // the planes this chain writes as 4-frame records (StackBP::rec): the lane's 16 bytes of piece 0
      const long nn = nbase + t;
      if (voff_gb[ft] != SK_OOB) { voff_gb[ft] = (int)((nn >> 2) * 1024 + (nn & 3) * 16); voff_r[ft] = (int)((nn >> 2) * 512 + (nn & 3) * 16); }
      else voff_r[ft] = SK_OOB;
    } else voff_r[ft] = voff_b[ft];
  }
  const int ch0 = 32 * mt + 4 * half;  // first channel of quad 0 of this lane's accumulator tile

  const uint16_t* wl = p.whi + lane * 8;
#if S2B_ABL & 8
#define S2B_WLOAD(off) (sk_u32x4{(unsigned)(off), (unsigned)(off) + 1u, (unsigned)lane, 0x3f803f80u})
#else
#define S2B_WLOAD(off) (*reinterpret_cast<const sk_u32x4*>(wl + (off)))
#endif

  // accumulator-layout quads (q = 0..3: channels ch0 + 8q .. + 3) of a bf16 [N,64] plane, 8 bytes each
#define S2B_LOADQ(dst, rsrc, ft)                                                                                  \
  _Pragma("unroll") for (int q = 0; q < 4; q++)                                                                   \
    dst[q] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_in[ft] + (ch0 + 8 * q) * 2, 0, 0);
  // four floats of one 8-byte piece
#define S2B_UNPACK(dst, w2)                                                                                       \
  { dst[0] = sk_u2f((w2)[0] << 16); dst[1] = sk_u2f((w2)[0] & 0xffff0000u); dst[2] = sk_u2f((w2)[1] << 16); dst[3] = sk_u2f((w2)[1] & 0xffff0000u); }
  // 16 accumulator-layout values (4 quads) -> two 16-byte pieces (channels 32 mt + 16 g + 8 half .. + 7, g = 0, 1)
#define S2B_PIECES(f0, f1, vals)                                                                                  \
  {                                                                                                               \
    sk_u32x2 qh_[4], ql_;                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; q++) sk_quad<false>(vals[4 * q], vals[4 * q + 1], vals[4 * q + 2], vals[4 * q + 3], qh_[q], ql_); \
    f0 = sk_frag_bits(sk_swap_frag(qh_[0], qh_[1]));                                                              \
    f1 = sk_frag_bits(sk_swap_frag(qh_[2], qh_[3]));                                                              \
  }
  const int colb = (32 * mt + 8 * half) * 2;  // byte column of piece g = 0 in a 64-channel row; g = 1: + 32
  // the same for the planes of the weight gradient: row-major as above, or records (piece c = channels / 8 at c * 64)
  const int colr = p.rec ? (4 * mt + half) * 64 : colb;
  const int pg1 = p.rec ? 128 : 32, pg8 = p.rec ? 512 : 128;  // two pieces on (16 channels), eight pieces on (64 channels)

  f32x16 acc[FT], dxo[FT], accc[AUX ? FT : 1];

  // ================= the head's data gradient: dy -> (W2^T, x relu'(H1)) = G1 -> (W1^T, x relu'(S), x sqrt(1/L)) = dS =================
  if constexpr (!FOLD) {
    // dS from the head's backward launch: this lane's two 8-channel pieces per frame tile, fp32 -> bf16 (the LDS operand of
    // every block's 1x1 and the plane the skip convs' weight gradients read)
    const __amdgpu_buffer_rsrc_t rds = sk_rsrc(p.dS, P);
    const __amdgpu_buffer_rsrc_t r_sh = sk_rsrc16(p.dsb_hi, P);
    sk_u32x4 sa[FT][2], sc[FT][2];
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int g = 0; g < 2; g++) {
        const int vo = rin[ft] ? (int)(((nbase + t0 - p.hl + row[ft]) * 64 + 32 * mt + 16 * g + 8 * half) * 4) : SK_OOB;
        sa[ft][g] = __builtin_amdgcn_raw_buffer_load_b128(rds, vo, 0, 0);
        sc[ft][g] = __builtin_amdgcn_raw_buffer_load_b128(rds, vo + 16, 0, 0);
      }
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int g = 0; g < 2; g++) {
        const sk_u32x4 f = {pack_bf2(sk_u2f(sa[ft][g][0]), sk_u2f(sa[ft][g][1])), pack_bf2(sk_u2f(sa[ft][g][2]), sk_u2f(sa[ft][g][3])),
                            pack_bf2(sk_u2f(sc[ft][g][0]), sk_u2f(sc[ft][g][1])), pack_bf2(sk_u2f(sc[ft][g][2]), sk_u2f(sc[ft][g][3]))};
        *reinterpret_cast<sk_u32x4*>(dst + row[ft] * XS + colb + 32 * g) = f;
        __builtin_amdgcn_raw_buffer_store_b128(f, r_sh, voff_r[ft] + colr + pg1 * g, 0, 0);
      }
    for (int i = tid; i < R * XS / 16; i += NT) reinterpret_cast<uint4*>(xt)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < SK_GUARD * GS / 16; i += NT) {
      reinterpret_cast<uint4*>(gs)[i] = make_uint4(0, 0, 0, 0);
      reinterpret_cast<uint4*>(gs + (SK_GUARD + R) * GS)[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();  // dS tile, zeroed operand tile and guard rows visible
  } else {
    const int KY = p.kp_y >> 4;
    const __amdgpu_buffer_rsrc_t rdy = sk_rsrc(p.dy, N * p.lddy);
    const __amdgpu_buffer_rsrc_t r_g2 = sk_rsrc16(p.hb_hi, N * p.kp_y);
    const __amdgpu_buffer_rsrc_t r_m1 = sk_rsrc16(p.hmask_hi + P, P);
    const __amdgpu_buffer_rsrc_t r_m0 = sk_rsrc16(p.hmask_hi, P);
    sk_u32x2 pm1[FT][4], pm0[FT][4];
#pragma unroll
    for (int ft = 0; ft < FT; ft++) { S2B_LOADQ(pm1[ft], r_m1, ft) }
#pragma unroll
    for (int ft = 0; ft < FT; ft++) { S2B_LOADQ(pm0[ft], r_m0, ft) }
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[ft][i] = 0.f;
    // every dy load of the lane in flight before the first is consumed (a load per k step inside the MFMA loop is a full
    // HBM round trip per step); KY <= 8: out_ch <= 128
    sk_u32x4 ya[8][FT], yc[8][FT], w2f[8];
    const bool dy_vec = ((p.lddy & 3) == 0) && ((((uintptr_t)p.dy) & 15) == 0);
#pragma unroll
    for (int kc = 0; kc < 8; kc++)
      if (kc < KY) {
        w2f[kc] = S2B_WLOAD(p.f_h2 + (mt * KY + kc) * 512);
        const int c0 = 16 * kc + 8 * half;
#pragma unroll
        for (int ft = 0; ft < FT; ft++) {
          const long nn = nbase + t0 - p.hl + row[ft];
          const int vo = (rin[ft] && c0 < p.out_ch) ? (int)((nn * p.lddy + c0) * 4) : SK_OOB;
          if (dy_vec) {
            ya[kc][ft] = __builtin_amdgcn_raw_buffer_load_b128(rdy, vo, 0, 0);
            yc[kc][ft] = __builtin_amdgcn_raw_buffer_load_b128(rdy, vo + 16, 0, 0);
          } else {  // dy is a column slice of a wider gradient (the discriminator's input gradient): rows only 4-byte aligned
#pragma unroll
            for (int j = 0; j < 4; j++) {
              ya[kc][ft][j] = __builtin_amdgcn_raw_buffer_load_b32(rdy, vo + 4 * j, 0, 0);
              yc[kc][ft][j] = __builtin_amdgcn_raw_buffer_load_b32(rdy, vo + 16 + 4 * j, 0, 0);
            }
          }
        }
      }
#pragma unroll
    for (int kc = 0; kc < 8; kc++)
      if (kc < KY) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, w2f[kc]);
        const int c0 = 16 * kc + 8 * half;
#pragma unroll
        for (int ft = 0; ft < FT; ft++) {
          const sk_u32x4 fb = {pack_bf2(sk_u2f(ya[kc][ft][0]), sk_u2f(ya[kc][ft][1])), pack_bf2(sk_u2f(ya[kc][ft][2]), sk_u2f(ya[kc][ft][3])),
                               pack_bf2(sk_u2f(yc[kc][ft][0]), sk_u2f(yc[kc][ft][1])), pack_bf2(sk_u2f(yc[kc][ft][2]), sk_u2f(yc[kc][ft][3]))};
          if (mt == 0) {  // bf16 dy: the plane the weight gradient of the head's last conv reads
            const long nn = nbase + t0 - p.hl + row[ft];
            __builtin_amdgcn_raw_buffer_store_b128(fb, r_g2, rout[ft] ? (int)((nn * p.kp_y + c0) * 2) : SK_OOB, 0, 0);
          }
          acc[ft] = mfma_bf16(a, __builtin_bit_cast(bf16x8, fb), acc[ft]);
        }
      }
    // x relu'(H1) -> G1: the plane of the middle conv's weight gradient and (LDS) the operand of the next 1x1
    {
      const __amdgpu_buffer_rsrc_t r_g1 = sk_rsrc16(p.hb_hi + N * p.kp_y, P);
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          float mv[4];
          S2B_UNPACK(mv, pm1[ft][q])
#pragma unroll
          for (int j = 0; j < 4; j++) v[4 * q + j] = rin[ft] ? acc[ft][4 * q + j] * (mv[j] > 0.f ? 1.f : 0.f) : 0.f;
        }
        sk_u32x4 f0, f1;
        S2B_PIECES(f0, f1, v)
        *reinterpret_cast<sk_u32x4*>(xt + row[ft] * XS + colb) = f0;
        *reinterpret_cast<sk_u32x4*>(xt + row[ft] * XS + colb + 32) = f1;
        __builtin_amdgcn_raw_buffer_store_b128(f0, r_g1, voff_b[ft] + colb, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(f1, r_g1, voff_b[ft] + colb + 32, 0, 0);
      }
    }
    __syncthreads();  // G1 tile complete
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[ft][i] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, S2B_WLOAD(p.f_h1 + (mt * 4 + kc) * 512));
#pragma unroll
      for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(a, lds_frag(xt + row[ft] * XS + kc * 32 + half * 16), acc[ft]);
    }
    {
      const __amdgpu_buffer_rsrc_t r_sh = sk_rsrc16(p.dsb_hi, P);
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          float mv[4];
          S2B_UNPACK(mv, pm0[ft][q])
#pragma unroll
          for (int j = 0; j < 4; j++) v[4 * q + j] = rin[ft] ? acc[ft][4 * q + j] * (mv[j] > 0.f ? 1.f : 0.f) * p.head_scale : 0.f;
        }
        sk_u32x4 f0, f1;
        S2B_PIECES(f0, f1, v)
        *reinterpret_cast<sk_u32x4*>(dst + row[ft] * XS + colb) = f0;
        *reinterpret_cast<sk_u32x4*>(dst + row[ft] * XS + colb + 32) = f1;
        __builtin_amdgcn_raw_buffer_store_b128(f0, r_sh, voff_r[ft] + colr, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(f1, r_sh, voff_r[ft] + colr + pg1, 0, 0);
      }
      __syncthreads();  // dS tile complete; every read of the G1 tile done
    }
    // dX_L = 0: the first block's 1x1 operand tile is zero; guard rows of the dG tile are zero for good
    for (int i = tid; i < R * XS / 16; i += NT) reinterpret_cast<uint4*>(xt)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < SK_GUARD * GS / 16; i += NT) {
      reinterpret_cast<uint4*>(gs)[i] = make_uint4(0, 0, 0, 0);
      reinterpret_cast<uint4*>(gs + (SK_GUARD + R) * GS)[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();  // zeroed operand tile visible to its readers
  }
#pragma unroll
  for (int ft = 0; ft < FT; ft++)
#pragma unroll
    for (int i = 0; i < 16; i++) { dxo[ft][i] = 0.f; if (AUX) accc[ft][i] = 0.f; }

  S2B_T(0)
  const float rs = 0.70710678118654752440f;
  constexpr int NS2 = KT * 8 + (AUX ? 8 : 0);  // weight fragments of phase 2: taps, then the conditioning 1x1

  // fragment s of a block's phase 2
#define S2B_FRAG2(fconv, faux, s) S2B_WLOAD(((s) < KT * 8 ? (fconv) + ((((s) >> 3) * 2 + mt) * 8 + ((s) & 7)) * 512 \
                                                          : (faux) + (mt * 8 + ((s) - KT * 8)) * 512))
  sk_u32x4 wos[8], ring[S2B_RING];
  // the layer-table entry as scalars (a struct copy lands in scratch memory); the next block's a whole block ahead of use
  long long c_conv = p.layers[p.L - 1].f_conv, c_aux = p.layers[p.L - 1].f_aux;
  int c_dil = p.layers[p.L - 1].dil, c_off0 = p.layers[p.L - 1].off0;
  {
    const long long f_os0 = p.layers[p.L - 1].f_os;
#pragma unroll
    for (int kc = 0; kc < 8; kc++) wos[kc] = S2B_WLOAD(f_os0 + (mt * 8 + kc) * 512);
  }
#pragma unroll
  for (int s = 0; s < S2B_RING; s++) ring[s] = S2B_FRAG2(c_conv, c_aux, s);

  // tanh / sigmoid planes of a block in the accumulator layout: requested a whole phase ahead of the gate backward
  // tanh / sigmoid planes of a block: the forward (stack2_fwd_kernel) wrote them in the lane-record layout (StackP::ts_stride):
  // piece g of this wave's channel group = the lane's accumulator-layout quads 2g, 2g + 1, 1 KB contiguous per 32 frames.
  // (As 8-byte quads of the [N,64] row layout the same data cost 3 k cache-line requests per block and CU, every wave
  // instruction touching 32 lines: the loads of a block took ~9 k cycles to drain.)
  int voff_ts[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ft++) {
    const long nn_ = nbase + t0 - p.hl + row[ft];
    voff_ts[ft] = rin[ft] ? (int)((nn_ >> 5) * 4096 + (nn_ & 31) * 16 + half * 512 + mt * 2048) : SK_OOB;
  }
  sk_u32x4 tp[FT][2], sp[FT][2];
#define S2B_REQ_PLANES(lb)                                                                                        \
  {                                                                                                               \
    const __amdgpu_buffer_rsrc_t r_th_ = sk_rsrc16(p.tb_hi + (long)(lb) * p.ts_stride, p.ts_stride);              \
    const __amdgpu_buffer_rsrc_t r_sh_ = sk_rsrc16(p.sg_hi + (long)(lb) * p.ts_stride, p.ts_stride);              \
    _Pragma("unroll") for (int ft = 0; ft < FT; ft++)                                                             \
      _Pragma("unroll") for (int g = 0; g < 2; g++) {                                                             \
        tp[ft][g] = (S2B_ABL & 16) ? sk_u32x4{0x3f003e80u, 0x3e003f00u, (unsigned)lane, 0x3f003f00u}              \
                                   : __builtin_amdgcn_raw_buffer_load_b128(r_th_, voff_ts[ft] + 1024 * g, 0, 0);  \
        sp[ft][g] = (S2B_ABL & 16) ? sk_u32x4{0x3f003e80u, 0x3e003f00u, (unsigned)lane, 0x3f003f00u}              \
                                   : __builtin_amdgcn_raw_buffer_load_b128(r_sh_, voff_ts[ft] + 1024 * g, 0, 0);  \
      }                                                                                                           \
  }
  // piece g -> quads 2g, 2g + 1 (8 floats: quad 2g first)
#define S2B_PIECE_TO_QUADS(dst, pc)                                                                               \
  { _Pragma("unroll") for (int j = 0; j < 4; j++) { dst[2 * j] = sk_u2f((pc)[j] << 16); dst[2 * j + 1] = sk_u2f((pc)[j] & 0xffff0000u); } }
  S2B_REQ_PLANES(p.L - 1)

  for (int l = p.L - 1; l >= 0; l--) {
    const int ln = __builtin_amdgcn_readfirstlane(l > 0 ? l - 1 : 0);  // (uniform: the table is read with scalar loads)
    const long long n_os = p.layers[ln].f_os, n_conv = p.layers[ln].f_conv, n_aux = p.layers[ln].f_aux;
    const int n_dil = p.layers[ln].dil, n_off0 = p.layers[ln].off0;
    // ---------------- phase 1: out|skip 1x1 transposed + gate backward ----------------
    {
      // eight k steps of tile ft: the dS half first (operands ready-made), then the dX half from the LDS tile
#define S2B_DZ(ft)                                                                                                \
  {                                                                                                               \
    bf16x8 xb_[4], sb_[4];                                                                                        \
    _Pragma("unroll") for (int kc = 0; kc < 4; kc++) sb_[kc] = lds_frag(dst + row[ft] * XS + kc * 32 + half * 16); \
    _Pragma("unroll") for (int kc = 0; kc < 4; kc++) xb_[kc] = lds_frag(xt + row[ft] * XS + kc * 32 + half * 16); \
    _Pragma("unroll") for (int i = 0; i < 16; i++) acc[ft][i] = 0.f;                                              \
    _Pragma("unroll") for (int st = 0; st < 4; st++) acc[ft] = mfma_bf16(__builtin_bit_cast(bf16x8, wos[4 + st]), sb_[st], acc[ft]); \
    _Pragma("unroll") for (int st = 0; st < 4; st++) acc[ft] = mfma_bf16(__builtin_bit_cast(bf16x8, wos[st]), xb_[st], acc[ft]); \
  }
#define S2B_GATE(ft)                                                                                              \
  {                                                                                                               \
    float da[16], db[16];                                                                                         \
    _Pragma("unroll") for (int g = 0; g < 2; g++) {                                                               \
      float ta[8], sb[8];                                                                                         \
      S2B_PIECE_TO_QUADS(ta, tp[ft][g])                                                                           \
      S2B_PIECE_TO_QUADS(sb, sp[ft][g])                                                                           \
      _Pragma("unroll") for (int j = 0; j < 8; j++) {                                                             \
        if (S2B_ABL & 1) { da[8 * g + j] = acc[ft][8 * g + j] + ta[j]; db[8 * g + j] = acc[ft][8 * g + j] + sb[j]; } \
        else sk_gate_bwd(acc[ft][8 * g + j], ta[j], sb[j], da[8 * g + j], db[8 * g + j]);                         \
      }                                                                                                           \
    }                                                                                                             \
    sk_u32x4 a0, a1, b0, b1;                                                                                      \
    S2B_PIECES(a0, a1, da)                                                                                        \
    S2B_PIECES(b0, b1, db)                                                                                        \
    unsigned char* dst = gs + (SK_GUARD + row[ft]) * GS + colb;                                                   \
    *reinterpret_cast<sk_u32x4*>(dst) = a0;                                                                       \
    *reinterpret_cast<sk_u32x4*>(dst + 32) = a1;                                                                  \
    *reinterpret_cast<sk_u32x4*>(dst + 128) = b0;                                                                 \
    *reinterpret_cast<sk_u32x4*>(dst + 160) = b1;                                                                 \
  }
      // software pipeline over the frame tiles: the MFMAs of tile ft + 1 are in flight under the gate arithmetic of tile ft
      S2B_DZ(0)
#pragma unroll
      for (int ft = 1; ft < FT; ft++) {
        S2B_DZ(ft)
        S2B_GATE(ft - 1)
      }
      S2B_GATE(FT - 1)
#undef S2B_DZ
#undef S2B_GATE
    }
    S2B_T(1)
    __syncthreads();  // dG tile complete; every read of the 1x1 operand tile done
    S2B_T(2)
    // ---------------- phase 2: transposed dilated conv (+ conditioning gradient) ----------------
    {
#pragma unroll
      for (int ft = 0; ft < FT; ft++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[ft][i] = 0.f;
      const unsigned char* gb0 = gs + (SK_GUARD + rb + l31 + c_off0) * GS + half * 16;  // tap 0, tile 0
      const unsigned char* gc0 = gs + (SK_GUARD + rb + l31) * GS + half * 16;            // conditioning: no shift
      bf16x8 bq[3][FT];  // B fragments two k steps (~200 cycles) ahead of their MFMAs
#define S2B_BREAD(s)                                                                                              \
  {                                                                                                               \
    const unsigned char* src_ = ((s) < KT * 8 ? gb0 + ((s) >> 3) * c_dil * GS + ((s) & 7) * 32 : gc0 + ((s) - KT * 8) * 32); \
    _Pragma("unroll") for (int ft = 0; ft < FT; ft++) bq[(s) % 3][ft] = lds_frag(src_ + ft * 32 * GS);            \
  }
      S2B_BREAD(0)
      S2B_BREAD(1)
#pragma unroll
      for (int s = 0; s < NS2; s++) {
        if (s + 2 < NS2) S2B_BREAD(s + 2)
        const bf16x8 a = __builtin_bit_cast(bf16x8, ring[s % S2B_RING]);
        __builtin_amdgcn_sched_barrier(0);
        if (s < KT * 8) {
#pragma unroll
          for (int ft = 0; ft < FT; ft++) acc[ft] = mfma_bf16(a, bq[s % 3][ft], acc[ft]);
        } else if (AUX) {
#pragma unroll
          for (int ft = 0; ft < FT; ft++) accc[ft] = mfma_bf16(a, bq[s % 3][ft], accc[ft]);
        }
        // the slot is free: the fragment S2B_RING steps ahead (the next block's first ones at the end).  The memory counter
        // retires in order: whatever is requested in front of a fragment delays the wait for it.  The next block's 1x1
        // fragments and gate planes (HBM, ~1.5 k cache-line requests per CU) therefore go BEHIND this block's last own
        // fragment - nothing of this phase waits for them - and in front of the next block's first ones, which are not
        // needed before its tap phase.
        // (Unconditionally, also in the last block, whose "next" entry is its own: a request inside `if (l > 0)` makes the
        // wait-count pass assume the shorter queue of the other path - every later wait of the phase then waits for the HBM
        // loads in front of it: 9 k cycles per block.)
        if (s + S2B_RING < NS2) ring[s % S2B_RING] = S2B_FRAG2(c_conv, c_aux, s + S2B_RING);
        else {
          if (s + S2B_RING == NS2) {
#pragma unroll
            for (int kc = 0; kc < 8; kc++) wos[kc] = S2B_WLOAD(n_os + (mt * 8 + kc) * 512);
            S2B_REQ_PLANES(ln)
          }
          ring[s % S2B_RING] = S2B_FRAG2(n_conv, n_aux, s + S2B_RING - NS2);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef S2B_PROF
        if (s == 0) S2B_T(8)
        if (s == 8) S2B_T(9)
        if (s == 16) S2B_T(10)
#endif
      }
#undef S2B_BREAD
      S2B_T(3)
      // Every HBM store of the block is issued here, at the very end: the memory counter retires in order, so a load that is
      // waited for (weight fragments, gate planes) must not have a store in front of it - acknowledged stores take thousands
      // of cycles when all CUs write their planes at once.  The wave's own dG pieces come back from the LDS tile.
      sk_u32x4 gpc[FT][4];
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        const unsigned char* src = gs + (SK_GUARD + row[ft]) * GS + colb;
        gpc[ft][0] = *reinterpret_cast<const sk_u32x4*>(src);
        gpc[ft][1] = *reinterpret_cast<const sk_u32x4*>(src + 32);
        gpc[ft][2] = *reinterpret_cast<const sk_u32x4*>(src + 128);
        gpc[ft][3] = *reinterpret_cast<const sk_u32x4*>(src + 160);
      }
      // dX_l = sqrt(.5) dX_{l+1} + convT(dG_l); kept in registers for block l - 1, bf16 plane for the weight gradient of
      // the out conv of block l - 1 (l = 0: of the first conv), bf16 x sqrt(.5) tile for the next 1x1 (l = 0: unscaled,
      // the first conv's data gradient consumes it)
      const __amdgpu_buffer_rsrc_t r_dh = sk_rsrc16(p.dxb_hi + (long)l * P, P);
      const bool drop = !FOLD && p.drop_p > 0.f;
      const unsigned long long dseed = (drop ? crk_seed(p.drop_seed, p.drop_seed_ptr) : 0ull) + 0x9E3779B97F4A7C15ull * (unsigned long long)(l + 1);
      const bool lmask = !FOLD && l == 0 && p.mask_l0;
      const __amdgpu_buffer_rsrc_t r_x0 = sk_rsrc(lmask ? p.saved : (const float*)p.dxb_hi, P);
      const __amdgpu_buffer_rsrc_t r_xo = sk_rsrc((!FOLD && p.dX0) ? p.dX0 : (float*)p.dxb_hi, P);
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        float ov[16], os[16];
        sk_u32x4 qm[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        const int voff_f = rin[ft] ? (int)(((nbase + t0 - p.hl + row[ft]) * 64 + ch0) * 4) : SK_OOB;  // fp32 [N,64], quad 0
        if (!FOLD && lmask) {
#pragma unroll
          for (int q = 0; q < 4; q++) qm[q] = __builtin_amdgcn_raw_buffer_load_b128(r_x0, voff_f + q * 32, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
          float cv = acc[ft][i];
          if (!FOLD && drop && rin[ft])
            cv *= dropout_scale(dseed, (unsigned long long)(nbase + t0 - p.hl + row[ft]) * 64 + ch0 + 8 * (i >> 2) + (i & 3), p.drop_p);
          float o = sk_res_bwd(dxo[ft][i], rs, cv);
          if (!FOLD && lmask) o *= (sk_u2f(qm[i >> 2][i & 3]) > 0.f ? 1.f : p.slope);
          o = rin[ft] ? o : 0.f;
          dxo[ft][i] = o;
          ov[i] = o;
          os[i] = sk_mul_nc(o, rs);
        }
        if (!FOLD && l == 0 && p.dX0) {  // fp32 dX_0 of the window's own frames: the first conv's backward launch reads it
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const sk_u32x4 v = {sk_f2u(ov[4 * q]), sk_f2u(ov[4 * q + 1]), sk_f2u(ov[4 * q + 2]), sk_f2u(ov[4 * q + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, r_xo, rout[ft] ? voff_f + q * 32 : SK_OOB, 0, 0);
          }
        }
        sk_u32x4 f0, f1;
        S2B_PIECES(f0, f1, ov)
        // (dX_0 row-major: the first conv's weight gradient reads it, see StackBP::rec)
        __builtin_amdgcn_raw_buffer_store_b128(f0, r_dh, l > 0 ? voff_r[ft] + colr : voff_b[ft] + colb, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(f1, r_dh, l > 0 ? voff_r[ft] + colr + pg1 : voff_b[ft] + colb + 32, 0, 0);
        if (l > 0) { S2B_PIECES(f0, f1, os) }
        *reinterpret_cast<sk_u32x4*>(xt + row[ft] * XS + colb) = f0;
        *reinterpret_cast<sk_u32x4*>(xt + row[ft] * XS + colb + 32) = f1;
      }
      const __amdgpu_buffer_rsrc_t r_gh = sk_rsrc16(p.gb_hi + (long)l * 2 * P, 2 * P);
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        __builtin_amdgcn_raw_buffer_store_b128(gpc[ft][0], r_gh, voff_gb[ft] + colr, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(gpc[ft][1], r_gh, voff_gb[ft] + colr + pg1, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(gpc[ft][2], r_gh, voff_gb[ft] + colr + pg8, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(gpc[ft][3], r_gh, voff_gb[ft] + colr + pg8 + pg1, 0, 0);
      }
    }
    S2B_T(4)
    __syncthreads();  // next 1x1 operand tile complete; every tap read of the dG tile done
    S2B_T(5)
    c_conv = n_conv; c_aux = n_aux; c_dil = n_dil; c_off0 = n_off0;
  }

  if (AUX) {
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
      if (rout[ft]) {
        float* dcr = p.dc + (nbase + t0 - p.hl + row[ft]) * p.lddc;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int ch = ch0 + (i & 3) + 8 * (i >> 2);
          if (ch < p.aux_ch) dcr[ch] = accc[ft][i];
        }
      }
  }
  // ================= the first conv's data gradient: dx = dx_scale * Wfirst^T . bf16(dX_0) =================
  if (FOLD && p.dx != nullptr) {
    const __amdgpu_buffer_rsrc_t rdx = sk_rsrc(p.dx, N * p.lddx);
    const int ntile = p.in_rows >> 5;
    bf16x8 xq[FT][4];
#pragma unroll
    for (int ft = 0; ft < FT; ft++)
#pragma unroll
      for (int kc = 0; kc < 4; kc++) xq[ft][kc] = lds_frag(xt + row[ft] * XS + kc * 32 + half * 16);
    for (int nt = mt; nt < ntile; nt += 2) {
      sk_u32x4 wf[4];
#pragma unroll
      for (int kc = 0; kc < 4; kc++) wf[kc] = S2B_WLOAD(p.f_first + (nt * 4 + kc) * 512);
#pragma unroll
      for (int ft = 0; ft < FT; ft++) {
        f32x16 a;
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; kc++) a = mfma_bf16(__builtin_bit_cast(bf16x8, wf[kc]), xq[ft][kc], a);
        const long nn = nbase + t0 - p.hl + row[ft];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int ch = nt * 32 + 8 * q + 4 * half;
          sk_u32x4 v;
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = sk_f2u(a[4 * q + j] * p.dx_scale);
          __builtin_amdgcn_raw_buffer_store_b128(v, rdx, (rout[ft] && ch < p.in_ch) ? (int)((nn * p.lddx + ch) * 4) : SK_OOB, 0, 0);
        }
      }
    }
  }
#ifdef S2B_PROF
  S2B_T(6)
  pacc_[7] = __builtin_readcyclecounter() - pstart_;
  if (blockIdx.x < 256 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 12; i++) s2b_prof_buf[(blockIdx.x * 4 + wave) * 12 + i] = pacc_[i];
  }
#endif
}

// four waves, one per SIMD: two frame halves of FT tiles each
template <int KT, bool AUX, int FT, bool FOLD = true>
__global__ __launch_bounds__(256, 1) void stack2_bwd_kernel(const StackBP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int fh = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 7);
  s2b_wave<KT, AUX, FT, 64 * FT, 4, FOLD>(p, smem, fh * 32 * FT);
}
// Window shapes: FT tiles of 32 frames per wave, two frame halves -> 64 FT rows: 192 (FT = 3) or 128 (FT = 2, short inputs).
int stack2_bwd_plan(StackBP& p) {
  if ((p.ktaps != 3 && p.ktaps != 5) || p.max_off > SK_GUARD || p.aux_ch > 64 || p.L > 16) return CRK_ERR_UNSUPPORTED;
  int best = 0; long best_cost = 0;
  for (int ft = 2; ft <= 3; ft++) {
    const int tmo = 64 * ft - p.hl - p.hr;
    if (tmo < 16) continue;
    const long wgs = (long)p.B * ceil_div(p.T, tmo);
    const long cost = ((wgs + 255) / 256) * ft;  // rounds of 256 workgroups x MFMAs per wave
    if (!best || cost < best_cost || (cost == best_cost && ft > best)) { best = ft; best_cost = cost; }
  }
  if (!best) return CRK_ERR_UNSUPPORTED;
  p.ft = best;
  p.dbg = 0;
  const int R = 64 * p.ft;
  p.tmo = R - p.hl - p.hr;
  p.tiles_per_utt = ceil_div(p.T, p.tmo);
  p.tmo = ceil_div(p.T, p.tiles_per_utt);
  int off = (SK_GUARD * 2 + R) * S2B_GS;
  p.o_dx = off; off += 2 * R * SK_XS;  // 1x1 operand tiles: dX (rewritten per block), dS
  p.lds_bytes = (off + 15) & ~15;
  return p.lds_bytes <= 160 * 1024 ? CRK_OK : CRK_ERR_UNSUPPORTED;
}

template <int KT, bool AUX, bool FOLD = true>
static int s2b_launch(const StackBP& p, dim3 grid, hipStream_t s) {
#define S2B_GO(FTV)                                                                                                  \
  {                                                                                                                  \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      if (hipFuncSetAttribute((const void*)stack2_bwd_kernel<KT, AUX, FTV, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != \
          hipSuccess) return CRK_ERR_HIP;                                                                            \
      attr = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((stack2_bwd_kernel<KT, AUX, FTV, FOLD>), grid, dim3(256), p.lds_bytes, s, p);                 \
  }
  if (p.ft == 2) S2B_GO(2)
  else S2B_GO(3)
#undef S2B_GO
  return CRK_OK;
}

int launch_stack2_bwd(const StackBP& p, hipStream_t s) {
  dim3 grid(p.B * p.tiles_per_utt);
  const double nfr = (double)p.B * p.T;
  const bool has_aux = p.dc != nullptr && p.aux_ch > 0;
  conv_prof_bytes(2, nfr * (256.0 + 256.0 * p.L + 256.0 * p.L + 128.0 * p.L + 128.0 + 256.0 + (has_aux ? 4.0 * p.aux_ch : 0.0)));
  conv_prof_begin(2, 2.0 * nfr * p.L * (64.0 * 128.0 * (1 + p.ktaps) + (has_aux ? 128.0 * p.aux_ch : 0.0)), s);
  int rc;
  if (p.dy == nullptr) {  // not folded: the discriminator (no conditioning)
    if (has_aux || !p.dS) return CRK_ERR_UNSUPPORTED;
    rc = p.ktaps == 3 ? s2b_launch<3, false, false>(p, grid, s) : s2b_launch<5, false, false>(p, grid, s);
  } else if (p.ktaps == 3) rc = has_aux ? s2b_launch<3, true>(p, grid, s) : s2b_launch<3, false>(p, grid, s);
  else rc = has_aux ? s2b_launch<5, true>(p, grid, s) : s2b_launch<5, false>(p, grid, s);
  conv_prof_end(2, s);
  if (rc != CRK_OK) return rc;
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
