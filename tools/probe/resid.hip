// Residency probe: how many 256-thread workgroups of a given LDS / VGPR footprint does a CU of the MI355X hold at once?
// Each block records its XCC id, CU id (HW_ID) and start / end times (s_memrealtime, 100 MHz); the host counts the
// maximum number of blocks whose [start, end] intervals overlap on the same (xcc, se, cu).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <map>
struct Rec { unsigned long long t0, t1; unsigned hwid, xcc; };
template <int NV>
__global__ __launch_bounds__(256, 2) void spin(Rec* out, int iters, float* sink) {
  extern __shared__ unsigned char smem[];
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = threadIdx.x * 0.001f + i;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = v[i] * 1.0001f + v[(i + 1) % NV];
    if (it == 1) smem[threadIdx.x] = (unsigned char)it;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; i++) s += v[i];
  sink[blockIdx.x * 256 + threadIdx.x] = s + smem[threadIdx.x];
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = {t0, t1, hwid, xcc};
  }
}
template <int NV>
void run(int lds, int grid) {
  Rec* d; float* sink;
  hipMalloc(&d, sizeof(Rec) * grid); hipMalloc(&sink, sizeof(float) * grid * 256);
  hipFuncSetAttribute((const void*)spin<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int nb = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin<NV>, 256, lds);
  hipLaunchKernelGGL(spin<NV>, dim3(grid), dim3(256), lds, 0, d, 3000, sink);
  hipDeviceSynchronize();
  std::vector<Rec> h(grid); hipMemcpy(h.data(), d, sizeof(Rec) * grid, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
  for (auto& r : h) {
    unsigned cu = (r.hwid >> 8) & 0xf, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 0x7;
    unsigned long long key = ((unsigned long long)(r.xcc & 0xf) << 16) | (se << 8) | (sh << 4) | cu;
    ev[key].push_back({r.t0, +1}); ev[key].push_back({r.t1, -1});
  }
  int mx = 0; unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& kv : ev) { std::sort(kv.second.begin(), kv.second.end()); int c = 0; for (auto& e : kv.second) { c += e.second; mx = std::max(mx, c); } }
  for (auto& r : h) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
  printf("NV %3d lds %6d B grid %4d: API says %d blocks/CU; observed max co-resident per CU %d over %zu distinct CUs; wall %.1f us\n", NV, lds, grid, nb,
         mx, ev.size(), (tmax - tmin) / 100.0);
  hipFree(d); hipFree(sink);
}
int main() {
  for (int lds : {1024, 40 * 1024, 50304, 68736, 80 * 1024}) {
    run<16>(lds, 512); run<100>(lds, 512); run<200>(lds, 512);
  }
  return 0;
}
