// Rate probe: v_mfma_f32_32x32x2_f32 chains per wave, 1 wave per SIMD, 4 waves per workgroup, 250 workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC, int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  float x = threadIdx.x * 0.001f, y = threadIdx.x * 0.002f + 1.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 32; j++) {
#pragma unroll
      for (int a = 0; a < NACC; a++) {
        if (MODE == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
        else {
          f32x4 t = {acc[a][0], acc[a][1], acc[a][2], acc[a][3]};
          t = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, t, 0, 0, 0);
          acc[a][0] = t[0]; acc[a][1] = t[1]; acc[a][2] = t[2]; acc[a][3] = t[3];
        }
      }
    }
    x += 1e-9f;
  }
  float s = 0.f;
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int MODE>
void run(const char* name, double flop_per_mfma) {
  float* d; hipMalloc(&d, 250 * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 16 / NACC;  // 512 MFMAs per wave
  probe<NACC, MODE><<<250, 256>>>(d, iters); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 20; i++) probe<NACC, MODE><<<250, 256>>>(d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / 20, n = 250.0 * 4 * 512;
  printf("%s: %.1f us per launch, %.1f ns per MFMA per wave, %.1f TFLOP/s\n", name, us, us * 1e3 / 512, n * flop_per_mfma / (us * 1e-6) / 1e12);
}
int main() {
  run<1, 0>("32x32x2 f32, 1 accumulator chain", 4096);
  run<2, 0>("32x32x2 f32, 2 accumulators", 4096);
  run<4, 0>("32x32x2 f32, 4 accumulators", 4096);
  run<1, 1>("16x16x4 f32, 1 accumulator chain", 2048);
  run<4, 1>("16x16x4 f32, 4 accumulators", 2048);
  return 0;
}
