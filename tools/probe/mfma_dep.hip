// Dependent-MFMA probe (gfx950): what an instruction BETWEEN two v_mfma_f32_32x32x16_bf16 on the SAME accumulator costs, next
// to the same fillers between MFMAs on alternating accumulators - the difference between stack2_fwd_kernel's per-tile chains
// (one accumulator, a fragment read and ~8 gate instructions in every gap) and its tap-major phase (three accumulators).
// One workgroup per CU, 1 or 2 waves per SIMD; cycles per MFMA group from clock64(), mean over workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_dep.hip -o tools/probe/mfma_dep && tools/probe/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// NACC accumulators used round robin, NV fillers (KIND 0 v_add_f32, 1 ds_read_b128, 2 v_exp_f32) behind every MFMA
template <int NACC, int NV, int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void probe(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  const int lane = threadIdx.x & 63;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = 1.0f + 0.001f * (lane + i);
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  u32x4 fa = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  bf16x8 A = __builtin_bit_cast(bf16x8, fa), B = A;
  unsigned char* lp = lds + (threadIdx.x & 255) * 16;
  *reinterpret_cast<u32x4*>(lp) = fa;
  __syncthreads();
  u32x4 dsv[4] = {fa, fa, fa, fa};
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      acc[g % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[g % NACC], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; k++) {
        const int r = (g * NV + k) & 7;
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 3) & 7]));
        if (KIND == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(dsv[k & 3]) : "v"((unsigned)(size_t)lp));
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
      }
    }
    if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += v[i];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[a][r];
  s += __builtin_bit_cast(float, dsv[0][0] + dsv[1][1] + dsv[2][2] + dsv[3][3]);
  sink[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int NV, int KIND, int WAVES>
void run() {
  static unsigned long long* d = nullptr; static float* sink = nullptr;
  if (!d) { hipMalloc(&d, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4); }
  const int iters = 64;
  for (int rep = 0; rep < 2; rep++) { probe<NACC, NV, KIND, WAVES><<<256, 64 * WAVES>>>(d, sink, iters); hipDeviceSynchronize(); }
  unsigned long long h[256 * 16];
  hipMemcpy(h, d, 256 * WAVES * 8, hipMemcpyDeviceToHost);
  double av = 0;
  for (int i = 0; i < 256 * WAVES; i++) av += h[i];
  av /= 256 * WAVES;
  const char* kn[] = {"v_add_f32", "ds_read_b128", "v_exp_f32"};
  printf("%d accumulator(s), %d x %-12s per MFMA, %d wave(s)/SIMD: %7.1f cycles per MFMA (wave's own clock)\n", NACC, NV, kn[KIND], WAVES / 4, av / (iters * 16.0));
}
#define ROW(NACC, KIND) run<NACC, 0, KIND, 4>(); run<NACC, 1, KIND, 4>(); run<NACC, 2, KIND, 4>(); run<NACC, 4, KIND, 4>(); run<NACC, 8, KIND, 4>(); \
                        run<NACC, 0, KIND, 8>(); run<NACC, 1, KIND, 8>(); run<NACC, 2, KIND, 8>(); run<NACC, 4, KIND, 8>(); run<NACC, 8, KIND, 8>();
int main() {
  ROW(1, 0) ROW(2, 0) ROW(3, 0)
  ROW(1, 1) ROW(3, 1)
  run<1, 4, 2, 4>(); run<3, 4, 2, 4>(); run<1, 4, 2, 8>(); run<3, 4, 2, 8>();
  return 0;
}
