// MCD evaluation with FastDTW alignment on the device (gfx950).
//
// Replaces crank/bin/evaluate_mcd.py:61-77 for a batch of utterance pairs: the third-party
// `fastdtw(cv, gt, dist=euclidean)` (radius 1) and the mel-cepstral distortion over the warping path.
// One wavefront per pair.  FastDTW is a chain of small windowed dynamic programs (coarse to fine, the
// window of a level is the widened projection of the coarser path: ~10 cells per row), so the work per
// pair is O(18 (nx + ny) D) and inherently sequential along the path; the parallelism is over pairs
// (hundreds per evaluation) and, inside a pair, over the cells of a row for the distance evaluations.
// Row costs live in LDS (no global-memory latency on the recurrence), back pointers stream to HBM.
// All arithmetic is float64 like numpy / the fastdtw package; a distance is accumulated over the
// dimensions in index order.
#include "common.h"
#include "../../include/crank_hip.h"

#pragma clang fp contract(off)

#define MCD_MAXLEV 24

struct McdArgs {
  const double* x; const long long* xoff;  // converted features, packed [sum nx][D]; pair p owns rows xoff[p]..xoff[p+1]
  const double* y; const long long* yoff;  // ground truth
  int P, D, radius, wmax;                  // wmax: LDS row capacity (>= longest y + 2)
  double* mcd; int* path_len;              // per pair
  int* path_out; long long path_stride;    // optional: (i, j) pairs of the final path, path_stride ints per pair
  unsigned char* scratch; long long scratch_stride;
  int* status;                             // per pair: 0 ok, 1 scratch / LDS capacity exceeded
};

__device__ __forceinline__ double mcd_dist(const double* a, const double* b, int D) {
  double s = 0.0;
  for (int d = 0; d < D; d++) {
    const double t = a[d] - b[d];
    s = s + t * t;
  }
  return sqrt(s);
}

__global__ __launch_bounds__(64) void mcd_dtw_kernel(const McdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* prev = reinterpret_cast<double*>(smem);  // costs of row i-1 over its window
  double* cur = prev + a.wmax;                     // row i
  double* drow = cur + a.wmax;                     // distances of row i
  const int p = blockIdx.x, lane = threadIdx.x, D = a.D, r = a.radius;
  const double INF = __builtin_inf();
  const double* x0 = a.x + a.xoff[p] * D;
  const double* y0 = a.y + a.yoff[p] * D;
  int nx[MCD_MAXLEV], ny[MCD_MAXLEV];
  nx[0] = (int)(a.xoff[p + 1] - a.xoff[p]);
  ny[0] = (int)(a.yoff[p + 1] - a.yoff[p]);
  if (nx[0] <= 0 || ny[0] <= 0 || ny[0] + 2 > a.wmax) {
    if (lane == 0) { a.status[p] = 1; a.mcd[p] = __builtin_nan(""); a.path_len[p] = 0; }
    return;
  }
  int L = 0;
  while (nx[L] >= r + 2 && ny[L] >= r + 2 && L + 1 < MCD_MAXLEV) { nx[L + 1] = nx[L] / 2; ny[L + 1] = ny[L] / 2; L++; }

  // ---- scratch of this pair ----
  unsigned char* sp = a.scratch + (long long)p * a.scratch_stride;
  const long long cells_max = 4LL * (2 * r + 1) * (2 * r + 1) * (nx[0] / 2 + ny[0] / 2 + 2) + (long long)(r + 2) * (nx[0] + ny[0]) + 64;
  double* px = reinterpret_cast<double*>(sp);                 // levels 1..L of x, concatenated (<= nx0 rows)
  double* py = px + (long long)nx[0] * D;                     // levels 1..L of y
  int* lo = reinterpret_cast<int*>(py + (long long)ny[0] * D);  // [nx0] window of the current level
  int* hi = lo + nx[0];
  int* roff = hi + nx[0];                                     // [nx0 + 1] first cell of a row
  int* pi = roff + nx[0] + 1;                                 // path of the current level, reversed: [nx0 + ny0]
  int* pj = pi + nx[0] + ny[0];
  unsigned char* back = reinterpret_cast<unsigned char*>(pj + nx[0] + ny[0]);  // [cells_max]
  const long long need = (long long)(back - sp) + cells_max;
  if (need > a.scratch_stride) {
    if (lane == 0) { a.status[p] = 1; a.mcd[p] = __builtin_nan(""); a.path_len[p] = 0; }
    return;
  }

  // ---- pyramid: level l+1 = averages of neighbouring rows of level l (an odd last row is dropped) ----
  long long xo[MCD_MAXLEV], yo[MCD_MAXLEV];  // row offsets of a level inside px / py (level 0 lives in the input)
  {
    long long ox = 0, oy = 0;
    for (int l = 1; l <= L; l++) { xo[l] = ox; yo[l] = oy; ox += nx[l]; oy += ny[l]; }
    for (int l = 1; l <= L; l++) {
      const double* sx = l == 1 ? x0 : px + xo[l - 1] * D;
      const double* sy = l == 1 ? y0 : py + yo[l - 1] * D;
      for (long long e = lane; e < (long long)nx[l] * D; e += 64) {
        const long long i = e / D, d = e - i * D;
        px[(xo[l] + i) * D + d] = (sx[(2 * i) * D + d] + sx[(2 * i + 1) * D + d]) / 2;
      }
      for (long long e = lane; e < (long long)ny[l] * D; e += 64) {
        const long long i = e / D, d = e - i * D;
        py[(yo[l] + i) * D + d] = (sy[(2 * i) * D + d] + sy[(2 * i + 1) * D + d]) / 2;
      }
      __syncthreads();
    }
  }

  // ---- base level: the full rectangle ----
  for (int i = lane; i < nx[L]; i += 64) { lo[i] = 0; hi[i] = ny[L] - 1; }
  __syncthreads();

  int plen = 0;
  for (int l = L; l >= 0; l--) {
    const double* xl = l == 0 ? x0 : px + xo[l] * D;
    const double* yl = l == 0 ? y0 : py + yo[l] * D;
    const int nxl = nx[l], nyl = ny[l];
    if (lane == 0) {
      int o = 0;
      for (int i = 0; i < nxl; i++) { roff[i] = o; o += hi[i] - lo[i] + 1; }
      roff[nxl] = o;
    }
    __syncthreads();
    if ((long long)roff[nxl] > cells_max) {
      if (lane == 0) { a.status[p] = 1; a.mcd[p] = __builtin_nan(""); a.path_len[p] = 0; }
      return;
    }
    // ---- windowed DTW, row by row: distances of the row by all lanes, the recurrence by lane 0 ----
    int plo = 0, phi = -1;  // window of the previous row (row -1: D[0][0] = 0 handled below)
    for (int i = 0; i < nxl; i++) {
      const int l0 = lo[i], h0 = hi[i], w = h0 - l0 + 1;
      const double* xr = xl + (long long)i * D;
      for (int c = lane; c < w; c += 64) drow[c] = mcd_dist(xr, yl + (long long)(l0 + c) * D, D);
      __syncthreads();
      if (lane == 0) {
        unsigned char* brow = back + roff[i];
        for (int c = 0; c < w; c++) {
          const int j = l0 + c;
          const double dt = drow[c];
          double up, diag;
          if (i == 0) {
            up = INF;                      // D[0][J] = inf for J >= 1
            diag = j == 0 ? 0.0 : INF;     // D[0][0] = 0
          } else {
            up = (j >= plo && j <= phi) ? prev[j - plo] : INF;
            diag = (j - 1 >= plo && j - 1 <= phi) ? prev[j - 1 - plo] : INF;
          }
          const double left = c > 0 ? cur[c - 1] : INF;
          // the package takes min() over (up + dt, left + dt, diag + dt) in this order: first minimum wins
          double best = up + dt;
          unsigned char bp = 0;
          const double cl = left + dt, cd = diag + dt;
          if (cl < best) { best = cl; bp = 1; }
          if (cd < best) { best = cd; bp = 2; }
          cur[c] = best;
          brow[c] = bp;
        }
      }
      __syncthreads();
      double* t = prev; prev = cur; cur = t;
      plo = l0; phi = h0;
    }
    // ---- backtrack (lane 0): path of this level, stored end to start ----
    if (lane == 0) {
      int i = nxl - 1, j = nyl - 1, n = 0;
      const int cap = nx[0] + ny[0];
      while (i >= 0 && j >= 0 && n < cap) {
        pi[n] = i; pj[n] = j; n++;
        const unsigned char bp = (j >= lo[i] && j <= hi[i]) ? back[roff[i] + j - lo[i]] : 2;
        if (bp == 0) i--;
        else if (bp == 1) j--;
        else { i--; j--; }
      }
      plen = n;
      if (l > 0) {
        // ---- window of level l-1: per coarse row the path's j range, widened by the radius in both
        // directions, doubled; rows scanned like the package's __expand_window (start_j carried over) ----
        const int nxf = nx[l - 1], nyf = ny[l - 1];
        // jmin / jmax per coarse row into hi/lo scratch of the NEXT level is not possible in place (the fine
        // level has more rows than the coarse one): use `roff` (free now) for jmin and the tail of pi/pj
        // is still needed, so jmax goes to the upper half of roff's companion: reuse prev/cur in LDS (ints)
        int* jmin = reinterpret_cast<int*>(prev);
        int* jmax = reinterpret_cast<int*>(cur);
        for (int ci = 0; ci < nxl; ci++) { jmin[ci] = 0x7fffffff; jmax[ci] = -1; }
        for (int k = 0; k < n; k++) {
          const int ci = pi[k], cj = pj[k];
          if (cj < jmin[ci]) jmin[ci] = cj;
          if (cj > jmax[ci]) jmax[ci] = cj;
        }
        int start_j = 0;
        for (int fi = 0; fi < nxf; fi++) {
          const int ci = fi >> 1;
          int e0 = 0x7fffffff, e1 = -1;
          for (int aa = -r; aa <= r; aa++) {
            const int cr = ci + aa;
            if (cr >= 0 && cr < nxl && jmax[cr] >= 0) {
              if (jmin[cr] - r < e0) e0 = jmin[cr] - r;
              if (jmax[cr] + r > e1) e1 = jmax[cr] + r;
            }
          }
          int f0 = 2 * e0, f1 = 2 * e1 + 1;
          if (f0 < start_j) f0 = start_j;
          if (f1 > nyf - 1) f1 = nyf - 1;
          if (e1 < 0 || f0 > f1) { f0 = start_j < nyf ? start_j : nyf - 1; f1 = f0; }  // (cannot happen for radius >= 1)
          lo[fi] = f0; hi[fi] = f1;
          start_j = f0;
        }
      }
    }
    __syncthreads();
    plen = __builtin_amdgcn_readfirstlane(plen);
  }

  // ---- MCD over the final path: 10 / ln 10 * sqrt(2 * sum_d (cv - gt)^2), averaged ----
  double s = 0.0;
  for (int k = lane; k < plen; k += 64) {
    const double* u = x0 + (long long)pi[k] * D;
    const double* v = y0 + (long long)pj[k] * D;
    double ss = 0.0;
    for (int d = 0; d < D; d++) {
      const double t = u[d] - v[d];
      ss = ss + t * t;
    }
    s += 10.0 / 2.30258509299404568402 * sqrt(2.0 * ss);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    a.mcd[p] = plen > 0 ? s / (double)plen : __builtin_nan("");
    a.path_len[p] = plen;
    a.status[p] = 0;
  }
  if (a.path_out) {
    int* po = a.path_out + (long long)p * a.path_stride;
    for (int k = lane; k < plen && 2LL * k + 1 < a.path_stride; k += 64) {
      po[2 * k] = pi[plen - 1 - k];
      po[2 * k + 1] = pj[plen - 1 - k];
    }
  }
}

static long long mcd_scratch_stride(int max_nx, int max_ny, int D, int radius) {
  const long long cells = 4LL * (2 * radius + 1) * (2 * radius + 1) * (max_nx / 2 + max_ny / 2 + 2) +
                          (long long)(radius + 2) * (max_nx + max_ny) + 64;
  long long b = 8LL * D * ((long long)max_nx + max_ny);           // pyramids
  b += 4LL * (3LL * max_nx + 1 + 2LL * (max_nx + max_ny));        // lo, hi, roff, path
  b += cells;                                                     // back pointers
  return (b + 255) & ~255LL;
}

extern "C" long long crk_mcd_scratch_bytes(int P, int max_nx, int max_ny, int D, int radius) {
  if (P <= 0 || max_nx <= 0 || max_ny <= 0 || D <= 0 || radius < 1) return -1;
  return mcd_scratch_stride(max_nx, max_ny, D, radius) * P;
}

extern "C" int crk_mcd_fastdtw(const double* cv, const long long* cv_off, const double* gt, const long long* gt_off, int P,
                               int D, int radius, int max_nx, int max_ny, double* mcd, int* path_len, int* path_out,
                               long long path_stride, void* scratch, int* status, void* stream) {
  if (!cv || !cv_off || !gt || !gt_off || !mcd || !path_len || !scratch || !status || P <= 0 || D <= 0 || radius < 1)
    return CRK_ERR_ARG;
  const int wmax = max_ny + 2 > max_nx + 2 ? max_ny + 2 : max_nx + 2;  // row costs; also the int scratch of the window expansion
  const size_t lds = (size_t)3 * wmax * sizeof(double);
  if (lds > 150 * 1024) return CRK_ERR_UNSUPPORTED;  // sequences longer than ~6000 voiced frames
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)mcd_dtw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return CRK_ERR_HIP;
    attr_set = true;
  }
  McdArgs a;
  a.x = cv; a.xoff = cv_off; a.y = gt; a.yoff = gt_off; a.P = P; a.D = D; a.radius = radius; a.wmax = wmax;
  a.mcd = mcd; a.path_len = path_len; a.path_out = path_out; a.path_stride = path_stride;
  a.scratch = (unsigned char*)scratch; a.scratch_stride = mcd_scratch_stride(max_nx, max_ny, D, radius);
  a.status = status;
  hipLaunchKernelGGL(mcd_dtw_kernel, dim3(P), dim3(64), lds, (hipStream_t)stream, a);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}
