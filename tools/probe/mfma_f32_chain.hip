// Issue rate of v_mfma_f32_32x32x2_f32 (the VQ search's instruction): a dependent accumulate chain into one accumulator vs
// two / four independent accumulators, one wave per SIMD (256 threads per CU) and two (512), cycles per MFMA of a wave.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f32_chain.hip -o /tmp/mfma_f32_chain && /tmp/mfma_f32_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int VALU, int NT>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[4];
  for (int a = 0; a < 4; a++) for (int i = 0; i < 16; i++) acc[a][i] = 0.f;
  float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f, v = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < 32; m++) {
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m % NACC], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < VALU; u++) v = __builtin_fmaf(v, 1.0001f, x);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = v;
  for (int a = 0; a < 4; a++) for (int i = 0; i < 16; i++) s += acc[a][i];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, int VALU, int NT = 256>
void run(const char* what) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * NT * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 200;
  hipLaunchKernelGGL((k<NACC, VALU, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<NACC, VALU, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; i++) m += h[i];
  printf("%-52s %7.1f cycles per mfma\n", what, m / 256 / (iters * 32.0));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 0>("32x32x2 f32, dependent chain");
  run<2, 0>("32x32x2 f32, 2 accumulators alternating");
  run<4, 0>("32x32x2 f32, 4 accumulators");
  run<1, 3>("dependent chain + 3 dependent v_fma per mfma");
  run<2, 3>("2 accumulators + 3 v_fma per mfma");
  // two waves per SIMD (512 threads per CU): cycles per mfma of ONE wave - 128 if the pipe is simply shared
  run<1, 0, 512>("2 waves/SIMD: dependent chain, no VALU");
  run<1, 3, 512>("2 waves/SIMD: dependent chain + 3 v_fma per mfma");
  run<1, 8, 512>("2 waves/SIMD: dependent chain + 8 v_fma per mfma");
  run<1, 8, 256>("1 wave/SIMD:  dependent chain + 8 v_fma per mfma");
  return 0;
}
