// Channel-last Conv1d kernels for gfx950: implicit-GEMM tile conv on
// v_mfma_f32_32x32x16_bf16 with LDS-staged operands, the fused gated residual block
// forward, the gate backward, the weight-gradient GEMM and the weight-norm kernels.
//
// Replaces the torch op clusters K1-K4/K12 of SURVEY.md section 2.2 (reference call
// sites: the parallel_wavegan stacks built at crank/net/module/vqvae2.py:237-273,
// crank/net/module/spkradv.py:49-60, crank/bin/train.py:78-128).
#include "conv_kernels.h"

#include <vector>

static inline int al16(int x) { return (x + 15) & ~15; }

static int sw_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
const CrkSwitches& crk_sw() {
  static const CrkSwitches s = [] {
    CrkSwitches v;
    v.sk_v = sw_int("CRK_SK_V", 2); v.skb_v = sw_int("CRK_SKB_V", 2); v.ps_v = sw_int("CRK_PS_V", 2);
    v.no_fuse = sw_int("CRK_NO_FUSE", 0); v.s2x = sw_int("CRK_S2X", 1); v.disc_split = sw_int("CRK_DISC_SPLIT", 1);
    v.s2_cfg = sw_int("CRK_S2_CFG", 0);
    const int nw = sw_int("CRK_SK_NW", 0);
    v.sk_nw_fwd = nw > 10 ? nw / 10 : nw; v.sk_nw_bwd = nw > 10 ? nw % 10 : nw;
    v.ps_nw = sw_int("CRK_PS_NW", 0);
    v.wg_groups = sw_int("CRK_WG_GROUPS", 32); if (v.wg_groups < 1) v.wg_groups = 32;
    v.wg_cpg = sw_int("CRK_WG_CPG", 0); v.wg_fill = sw_int("CRK_WG_FILL", 1) != 0;
    v.vq_f16 = sw_int("CRK_VQ_F16", 1) != 0; v.vq_lc = sw_int("CRK_VQ_LC", 2); v.logmel_wave = sw_int("CRK_LOGMEL_WAVE", 1);
    return v;
  }();
  return s;
}

// ------------------------------------------------------------------------------
// optional per-kernel-class timing with HIP events on the launch stream (bench.py's
// roofline leg).  One class per kernel: 0 conv_tile_kernel (generic per-layer conv, all modes),
// 1 stack_fwd_kernel, 2 stack_bwd_kernel, 3 wgrad_kernel (table), 4 pstack_kernel,
// 5 stack_wgrad_kernel, 6 pstack_wgrad_kernel.
// ------------------------------------------------------------------------------
struct ProfClass {
  std::vector<hipEvent_t> ev;  // start/stop pairs
  size_t used = 0;
  double flops = 0.0;
  double bytes = 0.0;  // algorithmic HBM bytes (what the launch must move: DESIGN.md section 3)
};
static bool g_prof = false;
static ProfClass g_pc[CRK_PROF_CLASSES];

static void prof_begin(int cls, double flops, hipStream_t s) {
  if (!g_prof) return;
  ProfClass& c = g_pc[cls];
  if (c.used + 2 > c.ev.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    c.ev.push_back(a); c.ev.push_back(b);
  }
  (void)hipEventRecord(c.ev[c.used], s);
  c.flops += flops;
}
static void prof_end(int cls, hipStream_t s) {
  if (!g_prof) return;
  ProfClass& c = g_pc[cls];
  if (c.used + 2 > c.ev.size()) return;
  (void)hipEventRecord(c.ev[c.used + 1], s);
  c.used += 2;
}
// Measurement aid (crk_debug_flush_before): a read-modify-write pass over a private buffer larger than the 256 MiB Infinity
// Cache in front of every profiled conv kernel, so that nothing its producer left in a cache is still there - the A/B that
// separates what a kernel reads from HBM from what it reads out of the last-level cache.
static long long g_flush_bytes = 0;
static float* g_flush_buf = nullptr;
static long long g_flush_cap = 0;
__global__ void cache_flush_kernel(float4* buf, long long n16) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
    float4 v = buf[i];
    v.x += 1.f;
    buf[i] = v;
  }
}
extern "C" int crk_debug_flush_before(long long bytes) {
  if (bytes < 0) return CRK_ERR_ARG;
  if (bytes > g_flush_cap) {
    if (g_flush_buf) (void)hipFree(g_flush_buf);
    g_flush_buf = nullptr; g_flush_cap = 0;
    if (hipMalloc(&g_flush_buf, (size_t)bytes) != hipSuccess) return CRK_ERR_HIP;
    if (hipMemset(g_flush_buf, 0, (size_t)bytes) != hipSuccess) return CRK_ERR_HIP;
    g_flush_cap = bytes;
  }
  g_flush_bytes = bytes;
  return CRK_OK;
}
void conv_prof_begin(int cls, double flops, hipStream_t s) {
  if (g_flush_bytes > 0) hipLaunchKernelGGL(cache_flush_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<float4*>(g_flush_buf), g_flush_bytes / 16);
  prof_begin(cls, flops, s);
}
void conv_prof_end(int cls, hipStream_t s) { prof_end(cls, s); }
void conv_prof_bytes(int cls, double bytes) { if (g_prof) g_pc[cls].bytes += bytes; }
extern "C" int crk_prof_enable(int on) {
  g_prof = on != 0;
  if (on) for (auto& c : g_pc) { c.used = 0; c.flops = 0.0; c.bytes = 0.0; }
  return CRK_OK;
}
// summed algorithmic HBM bytes of one class since crk_prof_enable(1)
extern "C" int crk_prof_report_bytes(int cls, double* total_bytes) {
  if (cls < 0 || cls >= CRK_PROF_CLASSES || !total_bytes) return CRK_ERR_ARG;
  *total_bytes = g_pc[cls].bytes;
  return CRK_OK;
}
// synchronises on the recorded events; returns launches, summed kernel time and the
// summed algorithmic FLOPs of one class since crk_prof_enable(1)
extern "C" int crk_prof_report(int cls, long long* count, double* total_ms, double* total_flops) {
  if (cls < 0 || cls >= CRK_PROF_CLASSES || !count || !total_ms || !total_flops) return CRK_ERR_ARG;
  ProfClass& c = g_pc[cls];
  double ms = 0.0;
  for (size_t i = 0; i + 1 < c.used; i += 2) {
    if (hipEventSynchronize(c.ev[i + 1]) != hipSuccess) return CRK_ERR_HIP;
    float t = 0.f;
    if (hipEventElapsedTime(&t, c.ev[i], c.ev[i + 1]) != hipSuccess) return CRK_ERR_HIP;
    ms += t;
  }
  *count = (long long)(c.used / 2); *total_ms = ms; *total_flops = c.flops;
  return CRK_OK;
}

// MFMA fragment maps (32x32x16 bf16): A lane l holds A[i=l&31][k=8*(l>>5)..+8];
// B lane l holds B[k=8*(l>>5)..+8][j=l&31]; C/D lane l, reg r holds
// C[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
__device__ __forceinline__ int cd_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ------------------------------------------------------------------------------
// staging helpers
// ------------------------------------------------------------------------------
__device__ __forceinline__ float load_src(const ConvP& p, long n, int c) {
  float v;
  if (c < p.cinA) {
    v = p.xa ? p.xa[n * p.lda + c] * p.scaleA : 0.f;
  } else {
    v = p.xb ? p.xb[n * p.ldb + (c - p.cinA)] * p.scaleB : 0.f;
  }
  v = apply_act(v, p.act_in, p.slope);
  if (p.drop_p > 0.f) v *= dropout_scale(crk_seed(p.drop_seed, p.drop_seed_ptr), (unsigned long long)n * p.cin + c, p.drop_p);
  return v;
}

template <bool PRECISE>
__device__ __forceinline__ void store4(unsigned char* hi, unsigned char* lo, const float v[4]) {
  *reinterpret_cast<uint2*>(hi) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
  if (PRECISE) {
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = v[j] - bf2f(f2bf(v[j]));
    *reinterpret_cast<uint2*>(lo) = make_uint2(pack_bf2(r[0], r[1]), pack_bf2(r[2], r[3]));
  }
}

// copy one prepared weight chunk [nrows][kp] (bf16) into LDS rows of stride ws
template <bool PRECISE>
__device__ __forceinline__ void stage_w(unsigned char* dhi, unsigned char* dlo, const uint16_t* shi, const uint16_t* slo,
                                        int nrows, int kp, int ws, int tid) {
  const int q_per_row = kp >> 3;
  const int total = nrows * q_per_row;
  for (int idx = tid; idx < total; idx += 256) {
    int r = idx / q_per_row, q = idx - r * q_per_row;
    *reinterpret_cast<uint4*>(dhi + r * ws + q * 16) = *reinterpret_cast<const uint4*>(shi + (long)r * kp + q * 8);
    if (PRECISE)
      *reinterpret_cast<uint4*>(dlo + r * ws + q * 16) = *reinterpret_cast<const uint4*>(slo + (long)r * kp + q * 8);
  }
}

// ------------------------------------------------------------------------------
// the tile kernel
// ------------------------------------------------------------------------------
// Global loads are issued in batches into registers before anything consumes them
// (one workgroup per CU at the benchmark shape: there is no second workgroup to hide a
// dependent load -> store chain behind), and the next weight chunk is fetched into
// registers while the current one feeds the MFMAs.
#define WREG_MAX 8  // 16-byte pieces per thread per plane for one weight chunk: 128 rows x 128 k / 8 / 256
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gate_tanh(float x, bool precise) {
  const float t = precise ? expf(2.f * x) : __expf(2.f * x);
  return 1.f - 2.f / (1.f + t);  // +-1 at the extremes (t = inf / 0)
}
__device__ __forceinline__ float gate_sigmoid(float x, bool precise) {
  const float t = precise ? expf(-x) : __expf(-x);
  return 1.f / (1.f + t);
}

// named fields instead of an array: the prefetch registers live across the (runtime)
// chunk loop and an indexed local array ends up in scratch
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct WRegs {
  u32x4 h0, h1, h2, h3, h4, h5, h6, h7;
  u32x4 l0, l1, l2, l3, l4, l5, l6, l7;
};
#define WR_ALL(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <bool PRECISE>
__device__ __forceinline__ void wfetch(WRegs& w, const uint16_t* shi, const uint16_t* slo, int nrows, int kp, int tid) {
  const int total = nrows * (kp >> 3);
#define WR_F(u)                                                                  \
  {                                                                              \
    const int idx = tid + u * 256;                                               \
    const long off = idx < total ? (long)idx * 8 : 0;                            \
    w.h##u = *reinterpret_cast<const u32x4*>(shi + off);                         \
    if (PRECISE) w.l##u = *reinterpret_cast<const u32x4*>(slo + off);            \
  }
  WR_ALL(WR_F)
#undef WR_F
}
template <bool PRECISE>
__device__ __forceinline__ void wcommit(const WRegs& w, unsigned char* dhi, unsigned char* dlo, int nrows, int kp, int ws,
                                        int tid) {
  const int qpr = kp >> 3, total = nrows * qpr;
#define WR_C(u)                                                                  \
  {                                                                              \
    const int idx = tid + u * 256;                                               \
    if (idx < total) {                                                           \
      const int r = idx / qpr, q = idx - r * qpr;                                \
      *reinterpret_cast<u32x4*>(dhi + r * ws + q * 16) = w.h##u;                 \
      if (PRECISE) *reinterpret_cast<u32x4*>(dlo + r * ws + q * 16) = w.l##u;    \
    }                                                                            \
  }
  WR_ALL(WR_C)
#undef WR_C
}

template <int MODE, int NT, bool PRECISE>
__global__ __launch_bounds__(256) void conv_tile_kernel(const ConvP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / p.tiles_per_utt, tile = blockIdx.x - b * p.tiles_per_utt;
  const int t0 = tile * CRK_TM;
  const long nbase = (long)b * p.T;
  const int HL = -p.off0;
  const int HR = p.off0 + (p.ktaps - 1) * p.dil;
  const int rows = CRK_TM + HL + HR;

  unsigned char* xs_hi = smem;
  unsigned char* xs_lo = smem + p.o_xlo;
  unsigned char* cs_hi = smem + p.o_chi;
  unsigned char* cs_lo = smem + p.o_clo;
  unsigned char* ws_hi = smem + p.o_whi;
  unsigned char* ws_lo = smem + p.o_wlo;
  unsigned char* zs_hi = smem + p.o_zhi;
  unsigned char* zs_lo = smem + p.o_zlo;
  const int XS = p.xs_stride, CS = p.cs_stride, WS = p.ws_stride, ZS = p.zs_stride;

  if (p.dbg & 128) return;
  const int nchunks = p.ktaps + ((MODE == MODE_RESFWD && p.cinC > 0) ? 1 : 0);
  const long wchunk_elems = (long)p.cout_pad * p.cin_pad;

  // first weight chunk is on its way while the activation tile is staged
  WRegs wr;
  wfetch<PRECISE>(wr, p.w_hi, p.w_lo, (p.dbg & 64) ? 0 : p.cout_pad, p.cin_pad, tid);

  // ---- stage the activation tile (with halo), fp32 -> bf16 (hi[/lo]) ----
  {
    const int qpr = p.cin_pad >> 2;  // channel quads per row
    int lg = 2;
    while ((1 << lg) < qpr) lg++;
    const int q = tid & ((1 << lg) - 1), r0 = tid >> lg, rstep = 256 >> lg;
    const int c4 = q << 2;
    const bool vecA = p.xa && ((p.lda & 3) == 0) && ((p.cinA & 3) == 0) && ((((uintptr_t)p.xa) & 15) == 0) &&
                      p.drop_p == 0.f;
    if (q < qpr) {
      const bool vec = vecA && (c4 + 3 < p.cinA);
      for (int rb = r0; rb < rows; rb += 4 * rstep) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {  // loads first
          const int r = rb + u * rstep;
          const int t = t0 - HL + r;
          v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
          if (r < rows && t >= 0 && t < p.T && !(p.dbg & 32)) {
            const long n = nbase + t;
            if (vec) {
              const float4 f = *reinterpret_cast<const float4*>(p.xa + n * p.lda + c4);
              v[u][0] = f.x; v[u][1] = f.y; v[u][2] = f.z; v[u][3] = f.w;
            } else {
#pragma unroll
              for (int j = 0; j < 4; j++)
                if (c4 + j < p.cin) v[u][j] = load_src(p, n, c4 + j);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {  // then transform + LDS stores
          const int r = rb + u * rstep;
          if (r < rows) {
            if (vec) {
#pragma unroll
              for (int j = 0; j < 4; j++) v[u][j] = apply_act(v[u][j] * p.scaleA, p.act_in, p.slope);
            }
            if (!(p.dbg & 256)) store4<PRECISE>(xs_hi + r * XS + c4 * 2, xs_lo + r * XS + c4 * 2, v[u]);
          }
        }
      }
    }
    if constexpr (MODE == MODE_RESFWD) if (p.cinC > 0) {
      const int qc = p.cinC_pad >> 2;
      int lgc = 2;
      while ((1 << lgc) < qc) lgc++;
      const int qq = tid & ((1 << lgc) - 1), rr0 = tid >> lgc, rs2 = 256 >> lgc;
      const int cc4 = qq << 2;
      if (qq < qc) {
        for (int rb = rr0; rb < CRK_TM; rb += 4 * rs2) {
          float v[4][4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int r = rb + u * rs2, t = t0 + r;
            v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
            if (r < CRK_TM && t < p.T) {
              const long n = nbase + t;
#pragma unroll
              for (int j = 0; j < 4; j++)
                if (cc4 + j < p.cinC) v[u][j] = p.xc[n * p.ldc + cc4 + j];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int r = rb + u * rs2;
            if (r < CRK_TM) store4<PRECISE>(cs_hi + r * CS + cc4 * 2, cs_lo + r * CS + cc4 * 2, v[u]);
          }
        }
      }
    }
  }

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[nt][i] = 0.f;

  for (int ch = 0; ch < nchunks; ch++) {
    const bool is_aux = ch >= p.ktaps;
    const int kp = is_aux ? p.cinC_pad : p.cin_pad;
    __syncthreads();  // previous chunk's fragments consumed (first pass: nothing pending)
    wcommit<PRECISE>(wr, ws_hi, ws_lo, (p.dbg & 64) ? 0 : p.cout_pad, kp, WS, tid);
    __syncthreads();  // weights (and, first pass, the activation tile) visible
    // prefetch the next chunk (or the 1x1 out|skip weights) behind this chunk's MFMAs
    if (ch + 1 < nchunks) {
      if (ch + 1 < p.ktaps)
        wfetch<PRECISE>(wr, p.w_hi + (ch + 1) * wchunk_elems, p.w_lo + (ch + 1) * wchunk_elems, (p.dbg & 64) ? 0 : p.cout_pad,
                        p.cin_pad, tid);
      else
        wfetch<PRECISE>(wr, p.wc_hi, p.wc_lo, (p.dbg & 64) ? 0 : p.cout_pad, p.cinC_pad, tid);
    } else if (MODE == MODE_RESFWD) {
      wfetch<PRECISE>(wr, p.w2_hi, p.w2_lo, (p.dbg & 64) ? 0 : 128, 64, tid);
    }
    const unsigned char* ab_hi = is_aux ? cs_hi : xs_hi;
    const unsigned char* ab_lo = is_aux ? cs_lo : xs_lo;
    const int AS = is_aux ? CS : XS;
    const int arow = wave * 32 + l31 + (is_aux ? 0 : ch * p.dil);
    const unsigned char* ap_hi = ab_hi + arow * AS + half * 16;
    const unsigned char* ap_lo = ab_lo + arow * AS + half * 16;
    const unsigned char* bp_hi = ws_hi + l31 * WS + half * 16;
    const unsigned char* bp_lo = ws_lo + l31 * WS + half * 16;
    const int nkc = (p.dbg & 8) ? 0 : (kp >> 4);
    for (int kc = 0; kc < nkc; kc++) {
      bf16x8 a_hi = lds_frag(ap_hi + kc * 32);
      bf16x8 a_lo;
      if (PRECISE) a_lo = lds_frag(ap_lo + kc * 32);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        bf16x8 b_hi = lds_frag(bp_hi + nt * 32 * WS + kc * 32);
        acc[nt] = mfma_bf16(a_hi, b_hi, acc[nt]);
        if (PRECISE) {
          bf16x8 b_lo = lds_frag(bp_lo + nt * 32 * WS + kc * 32);
          acc[nt] = mfma_bf16(a_lo, b_hi, acc[nt]);
          acc[nt] = mfma_bf16(a_hi, b_lo, acc[nt]);
        }
      }
    }
  }

  const int wrow0 = wave * 32;
  // this lane's 16 output rows
  bool rv[16];
  int rn[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int t = t0 + wrow0 + cd_row(i, half);
    rv[i] = t < p.T;
    rn[i] = (int)(nbase + t);
  }

  if constexpr (MODE == MODE_PLAIN) {
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int col = nt * 32 + l31;
      if (col >= p.cout) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
      float rsd[16], msk[16], old[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {  // all side loads first
        rsd[i] = (p.res && rv[i]) ? p.res[(long)rn[i] * p.ldr + col] : 0.f;
        msk[i] = (p.dmask && rv[i]) ? p.dmask[(long)rn[i] * p.ldm + col] : 1.f;
        old[i] = (p.accumulate && rv[i]) ? p.y[(long)rn[i] * p.ldy + col] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (!rv[i]) continue;
        float v = acc[nt][i] * p.out_scale + bv;
        v = apply_act(v, p.act_out, p.slope);
        if (p.epi_drop_p > 0.f)
          v *= dropout_scale(crk_seed(p.epi_drop_seed, p.drop_seed_ptr), (unsigned long long)rn[i] * p.cout + col, p.epi_drop_p);
        v += rsd[i] * p.res_scale;
        if (p.dmask) v *= act_grad(msk[i], p.dmask_act, p.slope);
        p.y[(long)rn[i] * p.ldy + col] = v + old[i];
      }
    }
  }

  if constexpr (MODE == MODE_RESFWD) {
    // NT == 4: acc[0..1] = tanh branch (cols 0..63), acc[2..3] = sigmoid branch (64..127)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const int col = nt * 32 + l31;
      const float ba = p.bias ? p.bias[col] : 0.f;
      const float bb = p.bias ? p.bias[64 + col] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int r = wrow0 + cd_row(i, half);
        const float ta = (p.dbg & 4) ? acc[nt][i] + ba : gate_tanh(acc[nt][i] + ba, PRECISE);
        const float sb = (p.dbg & 4) ? acc[nt + 2][i] + bb : gate_sigmoid(acc[nt + 2][i] + bb, PRECISE);
        const float z = ta * sb;
        if (rv[i] && !(p.dbg & 1)) {
          p.sv_ta[(long)rn[i] * 64 + col] = ta;
          p.sv_sb[(long)rn[i] * 64 + col] = sb;
          p.sv_z[(long)rn[i] * 64 + col] = z;
        }
        uint16_t zh, zl;
        if (PRECISE) split_bf(z, zh, zl);
        else zh = f2bf(z);
        *reinterpret_cast<uint16_t*>(zs_hi + r * ZS + col * 2) = zh;
        if (PRECISE) *reinterpret_cast<uint16_t*>(zs_lo + r * ZS + col * 2) = zl;
      }
    }
    // residual / running skip sums are fetched now and consumed after the second GEMM
    float xres[2][16], sk[2][16];
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
      const int col = h2 * 32 + l31;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        xres[h2][i] = (p.y && rv[i] && !(p.dbg & 16)) ? p.xa[(long)rn[i] * p.lda + col] : 0.f;
        sk[h2][i] = (!p.skip_init && rv[i] && !(p.dbg & 16)) ? p.skip[(long)rn[i] * 64 + col] : 0.f;
      }
    }
    __syncthreads();  // conv weights consumed, z tile complete
    wcommit<PRECISE>(wr, ws_hi, ws_lo, (p.dbg & 64) ? 0 : 128, 64, WS, tid);
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[nt][i] = 0.f;
    {
      const unsigned char* ap_hi = zs_hi + (wrow0 + l31) * ZS + half * 16;
      const unsigned char* ap_lo = zs_lo + (wrow0 + l31) * ZS + half * 16;
      const unsigned char* bp_hi = ws_hi + l31 * WS + half * 16;
      const unsigned char* bp_lo = ws_lo + l31 * WS + half * 16;
#pragma unroll
      for (int kc = 0; kc < 4; kc++) {
        bf16x8 a_hi = lds_frag(ap_hi + kc * 32);
        bf16x8 a_lo;
        if (PRECISE) a_lo = lds_frag(ap_lo + kc * 32);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          bf16x8 b_hi = lds_frag(bp_hi + nt * 32 * WS + kc * 32);
          acc[nt] = mfma_bf16(a_hi, b_hi, acc[nt]);
          if (PRECISE) {
            bf16x8 b_lo = lds_frag(bp_lo + nt * 32 * WS + kc * 32);
            acc[nt] = mfma_bf16(a_lo, b_hi, acc[nt]);
            acc[nt] = mfma_bf16(a_hi, b_lo, acc[nt]);
          }
        }
      }
    }
    const float rs = 0.70710678118654752440f;  // sqrt(0.5) as the reference's math.sqrt(0.5) rounds to fp32
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int h2 = nt & 1;
      const int col = h2 * 32 + l31;
      const bool is_out = nt < 2;
      const float bv = is_out ? (p.bias2a ? p.bias2a[col] : 0.f) : (p.bias2b ? p.bias2b[col] : 0.f);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (!rv[i] || (p.dbg & 2)) continue;
        if (is_out) {
          if (p.y) p.y[(long)rn[i] * p.ldy + col] = (acc[nt][i] + bv + xres[h2][i]) * rs;
        } else {
          const float s = acc[nt][i] + bv;
          p.skip[(long)rn[i] * 64 + col] = p.skip_init ? s : (sk[h2][i] + s);
        }
      }
    }
  }

  if constexpr (MODE == MODE_BWDA) {
    // NT == 2: acc = dz (cols 0..63); dG = [dz*sb*(1-ta^2) | dz*ta*sb*(1-sb)]
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int col = nt * 32 + l31;
      float tav[16], sbv[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        tav[i] = rv[i] ? p.ta[(long)rn[i] * 64 + col] : 0.f;
        sbv[i] = rv[i] ? p.sb[(long)rn[i] * 64 + col] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (!rv[i]) continue;
        const float dz = acc[nt][i];
        p.y[(long)rn[i] * p.ldy + col] = dz * sbv[i] * (1.f - tav[i] * tav[i]);
        p.y[(long)rn[i] * p.ldy + 64 + col] = dz * tav[i] * sbv[i] * (1.f - sbv[i]);
      }
    }
  }
}

// ------------------------------------------------------------------------------
// host side: LDS carve-up and dispatch
// ------------------------------------------------------------------------------

void conv_fill_lds(ConvP& p, int mode, bool precise) {
  const int HL = -p.off0, HR = p.off0 + (p.ktaps - 1) * p.dil;
  const int rows = CRK_TM + HL + HR;
  p.xs_stride = p.cin_pad * 2 + 16;
  p.cs_stride = (p.cinC_pad > 0 ? p.cinC_pad : 16) * 2 + 16;
  int kmax = p.cin_pad;
  if (mode == MODE_RESFWD) {
    if (p.cinC_pad > kmax) kmax = p.cinC_pad;
    if (64 > kmax) kmax = 64;
  }
  p.ws_stride = kmax * 2 + 16;
  p.zs_stride = 64 * 2 + 16;
  int off = 0;
  const int xbytes = al16(rows * p.xs_stride);
  off += xbytes;
  p.o_xlo = off; if (precise) off += xbytes;
  const int cbytes = (mode == MODE_RESFWD && p.cinC > 0) ? al16(CRK_TM * p.cs_stride) : 0;
  p.o_chi = off; off += cbytes;
  p.o_clo = off; if (precise) off += cbytes;
  const int wrows = (mode == MODE_RESFWD) ? 128 : p.cout_pad;
  const int wbytes = al16(wrows * p.ws_stride);
  p.o_whi = off; off += wbytes;
  p.o_wlo = off; if (precise) off += wbytes;
  const int zbytes = (mode == MODE_RESFWD) ? al16(CRK_TM * p.zs_stride) : 0;
  p.o_zhi = off; off += zbytes;
  p.o_zlo = off; if (precise) off += zbytes;
  p.lds_bytes = off;
}

typedef void (*conv_fn)(const ConvP);
template <int MODE, int NT, bool PR>
static conv_fn cf() { return conv_tile_kernel<MODE, NT, PR>; }

static conv_fn pick_conv(int mode, int nt, bool precise) {
  if (mode == MODE_RESFWD) return precise ? cf<MODE_RESFWD, 4, true>() : cf<MODE_RESFWD, 4, false>();
  if (mode == MODE_BWDA) return precise ? cf<MODE_BWDA, 2, true>() : cf<MODE_BWDA, 2, false>();
  switch (nt) {
    case 1: return precise ? cf<MODE_PLAIN, 1, true>() : cf<MODE_PLAIN, 1, false>();
    case 2: return precise ? cf<MODE_PLAIN, 2, true>() : cf<MODE_PLAIN, 2, false>();
    case 3: return precise ? cf<MODE_PLAIN, 3, true>() : cf<MODE_PLAIN, 3, false>();
    case 4: return precise ? cf<MODE_PLAIN, 4, true>() : cf<MODE_PLAIN, 4, false>();
  }
  return nullptr;
}

int conv_kernels_init() {
  const int maxlds = 160 * 1024;
  for (int pr = 0; pr < 2; pr++) {
    for (int nt = 1; nt <= 4; nt++) {
      conv_fn f = pick_conv(MODE_PLAIN, nt, pr);
      if (hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess) return CRK_ERR_HIP;
    }
    if (hipFuncSetAttribute((const void*)pick_conv(MODE_RESFWD, 4, pr), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess) return CRK_ERR_HIP;
    if (hipFuncSetAttribute((const void*)pick_conv(MODE_BWDA, 2, pr), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess) return CRK_ERR_HIP;
  }
  return CRK_OK;
}

int launch_conv(const ConvP& p, int mode, bool precise, hipStream_t s) {
  const int nt = p.cout_pad / 32;
  if (p.cout_pad % 32 || nt < 1 || nt > 4 || (p.cin_pad % 16) || p.off0 > 0 ||
      p.off0 + (p.ktaps - 1) * p.dil < 0 || p.lds_bytes > 160 * 1024) {
    fprintf(stderr, "[crank_hip] launch_conv: unsupported shape cout_pad=%d cin_pad=%d off0=%d k=%d dil=%d lds=%d\n",
            p.cout_pad, p.cin_pad, p.off0, p.ktaps, p.dil, p.lds_bytes);
    return CRK_ERR_UNSUPPORTED;
  }
  if (mode == MODE_RESFWD && nt != 4) return CRK_ERR_UNSUPPORTED;
  if (mode == MODE_BWDA && nt != 2) return CRK_ERR_UNSUPPORTED;
  conv_fn f = pick_conv(mode, nt, precise);
  dim3 grid(p.B * p.tiles_per_utt), block(256);
  const double nfr = (double)p.B * p.T;
  double fl = 2.0 * nfr * p.cout * p.cin * p.ktaps;
  if (mode == MODE_RESFWD) fl += 2.0 * nfr * 128.0 * (p.cinC + 64);
  prof_begin(0, fl, s);
  hipLaunchKernelGGL(f, grid, block, p.lds_bytes, s, p);
  prof_end(0, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

#include "wgrad_kernel.inc"

// split one problem into table entries (tap groups + optional aux entry) that respect
// the 10-tiles-per-wave and LDS limits
int wgrad_expand(const WgradP& job, bool precise, std::vector<WgradP>& out) {
  if ((job.ca_pad % 32) || (job.cx_pad % 32) || job.ca > 256 || job.ca_pad > 128 || job.cx_pad > 128) {
    fprintf(stderr, "[crank_hip] wgrad: unsupported shape ca_pad=%d cx_pad=%d\n", job.ca_pad, job.cx_pad);
    return CRK_ERR_UNSUPPORTED;
  }
  const int nct = job.ca_pad / 32, nit = job.cx_pad / 32;
  const int nctp = nct <= 1 ? 1 : (nct <= 2 ? 2 : 4);
  const int per_wave_cap = WG_MAXJ * (4 / nctp);  // (tap, cin-band) tiles one dY band can take
  int tg = job.ktaps;
  while (tg > 1) {
    WgradP t = job; t.grp_tap0 = 0; t.grp_ntap = tg; t.grp_aux = 0;
    const int rbs = 256 / (job.cx_pad / 4);  // input rows one pass of the workgroup covers
    const int nfs = 64 + (tg - 1) * job.dil;
    if (tg * nit <= per_wave_cap && wgrad_entry_lds(t, precise) <= 150 * 1024 && (nfs + rbs - 1) / rbs <= 11) break;
    tg--;
  }
  for (int t0 = 0; t0 < job.ktaps; t0 += tg) {
    WgradP e = job; e.grp_tap0 = t0; e.grp_ntap = (job.ktaps - t0 < tg) ? job.ktaps - t0 : tg; e.grp_aux = 0;
    if (wgrad_entry_lds(e, precise) > 150 * 1024) return CRK_ERR_UNSUPPORTED;
    out.push_back(e);
  }
  if (job.has_aux) {
    WgradP e = job; e.grp_tap0 = 0; e.grp_ntap = 0; e.grp_aux = 1;
    out.push_back(e);
  }
  return CRK_OK;
}

int launch_wgrad_table(const WgradP* d_jobs, const std::vector<WgradP>& h_jobs, int B, int T, int max_groups, bool precise,
                       hipStream_t s) {
  if (h_jobs.empty()) return CRK_OK;
  int lds = 0;
  double fl = 0.0;
  for (const auto& e : h_jobs) {
    const int l = wgrad_entry_lds(e, precise);
    if (l > lds) lds = l;
    fl += 2.0 * (double)B * T * e.ca * (e.grp_aux ? (double)e.cc : (double)e.cx * e.grp_ntap);
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
      return CRK_ERR_HIP;
    attr_set = true;
  }
  dim3 grid(max_groups, (unsigned)h_jobs.size()), block(256);
  prof_begin(3, fl, s);
  if (precise) hipLaunchKernelGGL(wgrad_kernel<true>, grid, block, lds, s, d_jobs, B, T);
  else hipLaunchKernelGGL(wgrad_kernel<false>, grid, block, lds, s, d_jobs, B, T);
  prof_end(3, s);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// ------------------------------------------------------------------------------
// weight norm: w = g * v / ||v||  (torch.nn.utils.weight_norm, dim 0), materialised
// as bf16 hi/lo planes in the forward and data-gradient layouts.
// grid (n_entries, 128): one 64-thread block per output channel.
// ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Weight preparation of a BAND of 16 output channels of one conv: w = g * v / ||v|| (weight norm, K1), split into bf16 hi /
// lo, written in every layout the kernels read - forward [tap][cout][cin], data-gradient [tap'][cin][cout] and the
// MFMA-fragment-ordered copies.  The band's rows of v are one contiguous run: read once, coalesced, into LDS; norms per
// row by one wave in the lane-strided order of the first version (one wave per output channel: same sums, same bits);
// then every layout leaves as 16-byte pieces - 8 consecutive input channels of a row, or 8 consecutive rows of an input
// channel.  (One workgroup per output channel wrote single bf16 values: the data-gradient and fragment layouts have
// consecutive OUTPUT channels adjacent, so every store was a 2-byte partial write - 35 us for the generator's 1.3 M weights.)
#define WP_BAND 16
typedef unsigned wp_u32x4 __attribute__((ext_vector_type(4)));
template <int BAND = WP_BAND>  // (8 in the fused update: the weight-norm backward's band; any multiple of 8 writes the same planes)
__device__ __forceinline__ void weight_prep_band(const ConvEntry& e, int band, const float* params, uint16_t* whi,
                                                 uint16_t* wlo, float* norms, unsigned* ws) {
  const int co0 = band * BAND;
  if (co0 >= e.cout) return;
  const int nrow = e.cout - co0 < BAND ? e.cout - co0 : BAND;
  const int k = e.k, cin = e.cin, n = cin * k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const float* v0 = params + e.off_v + (long long)co0 * n;
    const int tot = nrow * n;
    constexpr int WPQ = 28;  // loads in flight per thread: a band of rows of up to 448 weights in ONE memory round trip
    for (int i0 = 0; i0 < tot; i0 += 256 * WPQ) {
      float t[WPQ];
#pragma unroll
      for (int u = 0; u < WPQ; u++) { const int i = i0 + u * 256 + tid; t[u] = i < tot ? v0[i] : 0.f; }
#pragma unroll
      for (int u = 0; u < WPQ; u++) { const int i = i0 + u * 256 + tid; if (i < tot) ws[i] = __builtin_bit_cast(unsigned, t[u]); }
    }
  }
  float gv[BAND / 4];
#pragma unroll
  for (int q = 0; q < BAND / 4; q++) { const int r = wave + 4 * q; gv[q] = r < nrow ? params[e.off_g + co0 + r] : 0.f; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < BAND / 4; q++) {
    const int r = wave + 4 * q;
    if (r < nrow) {
      unsigned* row = ws + r * n;
      float ss = 0.f;
      for (int i = lane; i < n; i += 64) { const float x = __builtin_bit_cast(float, row[i]); ss += x * x; }
      ss = wave_sum(ss);
      const float nrm = sqrtf(ss);
      if (lane == 0) norms[e.norm_off + co0 + r] = nrm;
      const float sc = gv[q] / nrm;
      for (int i = lane; i < n; i += 64) {
        const float w = __builtin_bit_cast(float, row[i]) * sc;
        uint16_t h, l;
        split_bf(w, h, l);
        row[i] = (unsigned)h | ((unsigned)l << 16);
      }
    }
  }
  __syncthreads();
  // ---- runs of 8 input channels of (tap, row): forward layout (hi, lo) and the forward fragment copy ----
  {
    const int nc8 = e.fw_kp >> 3, per_tap = nrow * nc8, total = k * per_tap;
    const float i_pt = 1.f / (float)per_tap, i_nc = 1.f / (float)nc8;  // (index / small divisor through a reciprocal: exact here)
    for (int pidx = tid; pidx < total; pidx += 256) {
      const int tap = (int)(((float)pidx + 0.5f) * i_pt), rem = pidx - tap * per_tap;
      const int r = (int)(((float)rem + 0.5f) * i_nc), c8 = rem - r * nc8;
      const int co = co0 + r, ci0 = c8 * 8;
      unsigned wv[8];
#pragma unroll
      for (int j = 0; j < 8; j++) wv[j] = ci0 + j < cin ? ws[r * n + (ci0 + j) * k + tap] : 0u;
      const wp_u32x4 hp = {(wv[0] & 0xffffu) | (wv[1] << 16), (wv[2] & 0xffffu) | (wv[3] << 16), (wv[4] & 0xffffu) | (wv[5] << 16),
                           (wv[6] & 0xffffu) | (wv[7] << 16)};
      const wp_u32x4 lp = {(wv[0] >> 16) | (wv[1] & 0xffff0000u), (wv[2] >> 16) | (wv[3] & 0xffff0000u),
                           (wv[4] >> 16) | (wv[5] & 0xffff0000u), (wv[6] >> 16) | (wv[7] & 0xffff0000u)};
      const long long fi = e.fw_off + ((long long)tap * e.fw_rows + e.fw_row0 + co) * e.fw_kp + ci0;
      *reinterpret_cast<wp_u32x4*>(whi + fi) = hp;
      *reinterpret_cast<wp_u32x4*>(wlo + fi) = lp;
      if (e.fr_mode) {  // fragment-ordered copy (hi; lo for the split-operand forward, stack2x_kernels.hip)
        const int kc = ci0 >> 4, lh = 32 * ((ci0 & 15) >> 3);
        long long fo;
        if (e.fr_mode == 6) fo = ((((long long)(co >> 5) * k + tap) * (e.fw_kp >> 4) + kc) * 64 + (co & 31) + lh) * 8;
        else if (e.fr_mode == 5) fo = ((((long long)tap * (e.fw_rows >> 5) + (co >> 5)) * (e.fw_kp >> 4) + kc) * 64 + (co & 31) + lh) * 8;
        else {
          int mt, row;
          if (e.fr_mode <= 2) { const int hc = co & 63; mt = hc >> 4; row = (hc & 15) + (co >= 64 ? 16 : 0); }
          else { mt = (co >> 5) + (e.fr_mode == 4 ? 2 : 0); row = co & 31; }
          const int tp = e.fr_mode == 1 ? tap : 0;
          fo = ((((long long)tp * 4 + mt) * 4 + kc) * 64 + row + lh) * 8;
        }
        *reinterpret_cast<wp_u32x4*>(whi + e.fr_off + fo) = hp;
        *reinterpret_cast<wp_u32x4*>(wlo + e.fr_off + fo) = lp;
      }
    }
  }
  // ---- runs of 8 output channels of (tap, input channel): data-gradient layout (hi, lo) and its fragment copy ----
  if (e.bw_off >= 0) {
    const int ext = ((e.cout + 15) & ~15) - co0;  // columns of this entry from co0 on, padded to the layout's 16
    // (the entry's LAST band also writes the zero columns up to the layout's padding: with BAND = 8 and cout % 16 in 1 .. 8 the
    // band that would own columns 8 .. 15 of the last group does not exist - it returned above because co0 >= cout)
    const int no8 = ((co0 + BAND >= e.cout) ? ext : (ext < BAND ? ext : BAND)) >> 3;
    const int per_tap = cin * no8, total = k * per_tap;
    const float i_pt = 1.f / (float)per_tap, i_no = 1.f / (float)no8;
    for (int pidx = tid; pidx < total; pidx += 256) {
      const int tap = (int)(((float)pidx + 0.5f) * i_pt), rem = pidx - tap * per_tap;
      const int ci = (int)(((float)rem + 0.5f) * i_no), o8 = rem - ci * no8;
      unsigned wv[8];
#pragma unroll
      for (int j = 0; j < 8; j++) wv[j] = o8 * 8 + j < nrow ? ws[(o8 * 8 + j) * n + ci * k + tap] : 0u;
      const wp_u32x4 hp = {(wv[0] & 0xffffu) | (wv[1] << 16), (wv[2] & 0xffffu) | (wv[3] << 16), (wv[4] & 0xffffu) | (wv[5] << 16),
                           (wv[6] & 0xffffu) | (wv[7] << 16)};
      const wp_u32x4 lp = {(wv[0] >> 16) | (wv[1] & 0xffff0000u), (wv[2] >> 16) | (wv[3] & 0xffff0000u),
                           (wv[4] >> 16) | (wv[5] & 0xffff0000u), (wv[6] >> 16) | (wv[7] & 0xffff0000u)};
      const int kcol = e.bw_col0 + co0 + o8 * 8, tf = k - 1 - tap;
      const long long bi = e.bw_off + ((long long)tf * e.bw_rows + ci) * e.bw_kp + kcol;
      *reinterpret_cast<wp_u32x4*>(whi + bi) = hp;
      *reinterpret_cast<wp_u32x4*>(wlo + bi) = lp;
      if (e.bfr_off >= 0) {  // A-fragment order (row = input channel, k = column)
        const int ln = (ci & 31) + 32 * ((kcol & 15) >> 3);
        const long long fo = e.bfr_mode == 1
            ? ((((long long)(ci >> 5) * k + tf) * (e.bw_kp >> 4) + (kcol >> 4)) * 64 + ln) * 8
            : ((((long long)tf * (e.bw_rows >> 5) + (ci >> 5)) * (e.bw_kp >> 4) + (kcol >> 4)) * 64 + ln) * 8;
        *reinterpret_cast<wp_u32x4*>(whi + e.bfr_off + fo) = hp;
      }
    }
  }
}

__global__ __launch_bounds__(256) void weight_prep_kernel(const ConvEntry* ents, const float* params, uint16_t* whi,
                                                          uint16_t* wlo, float* norms) {
  extern __shared__ unsigned wp_ws[];
  const ConvEntry e = ents[blockIdx.x];
  weight_prep_band(e, blockIdx.y, params, whi, wlo, norms, wp_ws);
}

// the same for several nets (the sub-nets of one model share an optimizer step): workgroup x belongs to the net whose
// entry range holds it
__device__ __forceinline__ int net_of_block(const NetRefs& R, int bx) {
  int r = 0;
  while (r + 1 < R.n && bx >= R.r[r + 1].first) r++;
  return r;
}
template <int BAND>
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const NetRefs R) {
  extern __shared__ unsigned wp_ws[];
  const NetRef& q = R.r[net_of_block(R, blockIdx.x)];
  const ConvEntry e = q.ents[blockIdx.x - q.first];
  if (R.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) R.bump[0] += 1.f;
  weight_prep_band<BAND>(e, blockIdx.y, q.params, q.whi, q.wlo, q.norms, wp_ws);
}
__global__ void step_bump_kernel(float* step) { step[0] += 1.f; }
int launch_step_bump(float* step, hipStream_t s) {
  hipLaunchKernelGGL(step_bump_kernel, dim3(1), dim3(1), 0, s, step);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
// nmax: largest cin * k of the entries (LDS: a band of 32 rows of that length)
static int wp_set_lds(const void* fn, size_t lds) {
  if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return CRK_ERR_HIP;
  return CRK_OK;
}
int launch_weight_prep_multi(const NetRefs& R, int total_entries, int nmax, hipStream_t s) {
  const size_t lds = (size_t)WP_BAND * nmax * 4;
  if (lds > 160 * 1024) return CRK_ERR_UNSUPPORTED;
  if (total_entries <= 16) {
    // a small net (3 - 8 convs of 64 channels): 8-row bands - twice the workgroups, half the rows each wave walks in turn
    // (the kernel is a chain of short phases, 11 us for 40 k parameters; any band that is a multiple of 8 writes the same planes)
    hipLaunchKernelGGL(weight_prep_multi_kernel<8>, dim3(total_entries, 128 / 8), dim3(256), lds / 2, s, R);
    CRK_CHECK_LAUNCH();
    return CRK_OK;
  }
  if (wp_set_lds((const void*)weight_prep_multi_kernel<WP_BAND>, lds) != CRK_OK) return CRK_ERR_HIP;
  hipLaunchKernelGGL(weight_prep_multi_kernel<WP_BAND>, dim3(total_entries, 128 / WP_BAND), dim3(256), lds, s, R);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

int launch_weight_prep(const ConvEntry* d_entries, int n_entries, int nmax, const float* params, uint16_t* wprep_hi,
                       uint16_t* wprep_lo, float* norms, hipStream_t s) {
  const size_t lds = (size_t)WP_BAND * nmax * 4;
  if (lds > 160 * 1024) return CRK_ERR_UNSUPPORTED;
  if (wp_set_lds((const void*)weight_prep_kernel, lds) != CRK_OK) return CRK_ERR_HIP;
  dim3 grid(n_entries, 128 / WP_BAND), block(256);
  hipLaunchKernelGGL(weight_prep_kernel, grid, block, lds, s, d_entries, params, wprep_hi, wprep_lo, norms);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// dW (sum of the per-group partials, fixed order) -> dg, dv, dbias accumulated into the flat gradient block.
// The kernel is a pure read of G x |params| floats (G's generator: 155 MB at the benchmark shape, just written by the
// weight-gradient kernels) and must run at memory speed.  A workgroup owns a band of WN_RB output channels of one conv:
// for a fixed (group, tap) their partial sums are ONE contiguous run of WN_RB x cin floats (layout [group][tap][cout][cin]),
// read as 16-byte pieces with 8 loads in flight per thread.  (One workgroup per output channel read 256-byte runs scattered
// over groups and taps: 2 TB/s.)  Every element is summed over the groups in ascending order - the same value as before,
// bit for bit; the dot product <dW, v> of a row is a fixed tree over (tap, cin).
#ifndef WN_RB
#define WN_RB 8
#endif
#ifndef WN_IF
#define WN_IF 16  // loads in flight per thread
#endif
// Small nets (the speaker-adversarial net: 3 convs, the classifier: 8) run bands of RB = 2 rows with IF = 32 loads in
// flight: with 8-row bands they are 24 / 64 workgroups of 3 pieces per thread and four rounds of 16 loads each - twelve
// dependent memory round trips, 16 - 19 us for 40 k / 150 k parameters; 2-row bands are one piece per thread and two rounds.
// The sum over the groups keeps its ascending order either way (same bits).
template <int RB> struct WnormShared { float dw[RB][128 * 8]; };
template <int RB, int IF>
__device__ __forceinline__ void wnorm_bwd_body(const ConvEntry& e, int band, const float* params, float* grads,
                                               const float* partials, const float* norms, WnormShared<RB>& sh) {
  const int co0 = band * RB;
  if (co0 >= e.cout) return;
  const int nrow = e.cout - co0 < RB ? e.cout - co0 : RB;
  const int G = e.pt_groups, k = e.k, cin = e.cin, cx = e.pt_cx;
  const int n = cin * k;  // <= 128 * 8
  const int tid = threadIdx.x;
  const long long gstride = (long long)e.pt_taps * e.pt_rows * e.pt_cx;
  const long long tstride = (long long)e.pt_rows * e.pt_cx;
  const float* base = partials + e.pt_off + (long long)(e.pt_row0 + co0) * cx;
  // ---- everything the finishing phase reads from memory is requested NOW, next to the partial sums: this thread's
  // pieces of its row of v and of the gradient block it accumulates into (32 threads per row), the bias partials, norm
  // and g.  (Read where they are used, each "grads[i] += f(v[i])" was a dependent load -> store -> load chain - the
  // stores may alias the next load as far as the compiler knows -: n / 32 serial memory round trips per workgroup,
  // ~20 us however little data a net has.)
  const int r = tid >> 5, l32 = tid & 31;
  const bool on = r < nrow;
  const int co = co0 + r;
  const float* __restrict__ v = params + e.off_v + (long long)co * n;
  float* __restrict__ gv = grads + e.off_v + (long long)co * n;
  constexpr int WN_PV = (128 * 8) / 32;
  float vr[WN_PV], go[WN_PV];
#pragma unroll
  for (int q = 0; q < WN_PV; q++) {
    const int i = l32 + 32 * q;
    const bool ok = on && i < n;
    vr[q] = ok ? v[i] : 0.f;
    go[q] = ok ? gv[i] : 0.f;
  }
  float sb = 0.f;
  if (on && e.off_b >= 0)
    for (int g = l32; g < e.pt_groups; g += 32) sb += partials[e.pb_off + (long long)g * e.pt_rows + e.pt_row0 + co];
  const float nrm = on ? norms[e.norm_off + co] : 1.f;
  const float gval = on ? params[e.off_g + co] : 0.f;
  const float g_old = (on && l32 == 0) ? grads[e.off_g + co] : 0.f;
  const float b_old = (on && l32 == 0 && e.off_b >= 0) ? grads[e.off_b + co] : 0.f;
  // ---- dW of the band: for every tap a run of nrow * cx floats per group ----
  const int run = nrow * cx;
  if ((cx & 3) == 0 && ((((uintptr_t)base) & 15) == 0) && ((gstride | tstride) & 3) == 0) {
    const int run4 = run >> 2;
    // (taps and pieces as one index space: a 1x1 conv has only 128 pieces per band, a loop over the taps would leave half
    // the workgroup idle and serialise k x G / 8 memory round trips per thread)
    for (int it = tid; it < k * run4; it += 256) {
      const int tap = it / run4, i4 = it - tap * run4;
      const float* src = base + tap * tstride + 4 * i4;
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
      int g = 0;
      for (; g + IF <= G; g += IF) {
        f32x4 t[IF];
#pragma unroll
        for (int u = 0; u < IF; u++) t[u] = *reinterpret_cast<const f32x4*>(src + (long long)(g + u) * gstride);
#pragma unroll
        for (int u = 0; u < IF; u++) s4 += t[u];
      }
      if (IF > 16) {
        for (; g + 16 <= G; g += 16) {
          f32x4 t[16];
#pragma unroll
          for (int u = 0; u < 16; u++) t[u] = *reinterpret_cast<const f32x4*>(src + (long long)(g + u) * gstride);
#pragma unroll
          for (int u = 0; u < 16; u++) s4 += t[u];
        }
      }
      for (; g + 4 <= G; g += 4) {
        f32x4 t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) t[u] = *reinterpret_cast<const f32x4*>(src + (long long)(g + u) * gstride);
#pragma unroll
        for (int u = 0; u < 4; u++) s4 += t[u];
      }
      for (; g < G; g++) s4 += *reinterpret_cast<const f32x4*>(src + (long long)g * gstride);
      const int r = (4 * i4) / cx, ci = 4 * i4 - r * cx;
#pragma unroll
      for (int j = 0; j < 4; j++) sh.dw[r][(ci + j) * k + tap] = s4[j] * e.pt_scale;  // position in weight_v[co] (cin, k)
    }
  } else {
    for (int tap = 0; tap < k; tap++) {
      for (int i = tid; i < run; i += 256) {
        const float* src = base + tap * tstride + i;
        float sv = 0.f;
        int g = 0;
        for (; g + 8 <= G; g += 8) {
          float t[8];
#pragma unroll
          for (int u = 0; u < 8; u++) t[u] = src[(long long)(g + u) * gstride];
#pragma unroll
          for (int u = 0; u < 8; u++) sv += t[u];
        }
        for (; g < G; g++) sv += src[(long long)g * gstride];
        const int r = i / cx, ci = i - r * cx;
        sh.dw[r][ci * k + tap] = sv * e.pt_scale;
      }
    }
  }
  __syncthreads();
  // ---- per row: dot = <dW, v>, bias sum over the groups; 32 threads per row ----
  float dot = 0.f;
#pragma unroll
  for (int q = 0; q < WN_PV; q++) {
    const int i = l32 + 32 * q;
    if (on && i < n) dot += sh.dw[r][i] * vr[q];
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { dot += __shfl_xor(dot, o, 64); sb += __shfl_xor(sb, o, 64); }
  if (on) {
    const float inv = 1.f / nrm;
    if (l32 == 0) {
      grads[e.off_g + co] = g_old + dot * inv;
      if (e.off_b >= 0) grads[e.off_b + co] = b_old + sb * e.pt_scale;
    }
    const float c1 = gval * inv, c2 = dot * inv * inv;
#pragma unroll
    for (int q = 0; q < WN_PV; q++) {
      const int i = l32 + 32 * q;
      if (i < n) gv[i] = go[q] + c1 * (sh.dw[r][i] - c2 * vr[q]);  // (LDS: not ordered against the global stores)
    }
  }
}

__global__ __launch_bounds__(256) void wnorm_bwd_kernel(const ConvEntry* ents, const float* params, float* grads,
                                                       const float* partials, const float* norms) {
  __shared__ WnormShared<WN_RB> sh;
  const ConvEntry e = ents[blockIdx.x];
  wnorm_bwd_body<WN_RB, WN_IF>(e, blockIdx.y, params, grads, partials, norms, sh);
}
template <int RB, int IF>
__global__ __launch_bounds__(256) void wnorm_bwd_multi_kernel(const NetRefs R) {
  __shared__ WnormShared<RB> sh;
  const NetRef& q = R.r[net_of_block(R, blockIdx.x)];
  const ConvEntry e = q.ents[blockIdx.x - q.first];
  wnorm_bwd_body<RB, IF>(e, blockIdx.y, q.params, q.grads, q.partials, q.norms, sh);
}
#define WN_SMALL_ENTRIES 16  // at most this many convs in the launch: the latency-bound shape (2-row bands)
int launch_wnorm_bwd_multi(const NetRefs& R, int total_entries, hipStream_t s) {
  if (total_entries <= WN_SMALL_ENTRIES) hipLaunchKernelGGL((wnorm_bwd_multi_kernel<2, 32>), dim3(total_entries, 128 / 2), dim3(256), 0, s, R);
  else hipLaunchKernelGGL((wnorm_bwd_multi_kernel<WN_RB, WN_IF>), dim3(total_entries, 128 / WN_RB), dim3(256), 0, s, R);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}


int launch_wnorm_bwd(const ConvEntry* d_entries, int n_entries, const float* params, float* grads,
                     const float* partials, const float* norms, hipStream_t s) {
  dim3 grid(n_entries, 128 / WN_RB), block(256);
  hipLaunchKernelGGL(wnorm_bwd_kernel, grid, block, 0, s, d_entries, params, grads, partials, norms);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
