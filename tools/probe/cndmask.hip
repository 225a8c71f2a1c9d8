// Issue cost of v_cndmask_b32 (mask in VCC / in an SGPR pair) against the instructions that can replace it in the fused stack
// kernels' masking (v_and_b32 with a lane mask, v_mul_f32 by 0 / 1), alone and behind a v_mfma_f32_32x32x16_bf16.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/cndmask tools/probe/cndmask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
enum { K_ADD = 0, K_CND_VCC, K_CND_SGPR, K_AND, K_MUL };

template <int KIND, int NV, int NM, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void probe(unsigned long long* out, float* sink, int iters, unsigned long long m) {
  const int lane = threadIdx.x & 63;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = 1.0f + 0.001f * (lane + i);
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  u32x4 fa = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  bf16x8 A = __builtin_bit_cast(bf16x8, fa), B = A;
  const unsigned long long sm = __builtin_amdgcn_readfirstlane((unsigned)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32);
  const unsigned lm = (m >> lane) & 1 ? 0xffffffffu : 0u;
  const float fm = (m >> lane) & 1 ? 1.f : 0.f;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 8; g++) {
#pragma unroll
      for (int mm = 0; mm < NM; mm++) acc[(g * NM + mm) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[(g * NM + mm) & 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; k++) {
        const int r = (g * NV + k) & 15;
        if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 5) & 15]));
        if (KIND == K_CND_VCC) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[r]) : "v"(v[(r + 5) & 15]));
        if (KIND == K_CND_SGPR) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(v[r]) : "s"(sm));
        if (KIND == K_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[r]) : "v"(lm));
        if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[r]) : "v"(fm));
      }
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[a][r];
  sink[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NV, int NM, int WAVES>
void run(const char* name) {
  static unsigned long long* d = nullptr; static float* sink = nullptr;
  if (!d) { hipMalloc(&d, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4); }
  const int iters = 64;
  for (int rep = 0; rep < 2; rep++) { probe<KIND, NV, NM, WAVES><<<256, 64 * WAVES>>>(d, sink, iters, 0xfffffffffffffff0ull); hipDeviceSynchronize(); }
  unsigned long long h[256 * 16];
  hipMemcpy(h, d, 256 * WAVES * 8, hipMemcpyDeviceToHost);
  double mx = 0, mn = 1e30, av = 0;
  for (int i = 0; i < 256 * WAVES; i++) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; av += h[i]; }
  av /= 256 * WAVES;
  const double groups = iters * 8.0;
  printf("%-36s waves/CU %2d: %7.1f cycles per group (min %.1f max %.1f)  [group = %d MFMA + %d x op]\n", name, WAVES, av / groups, mn / groups, mx / groups, NM, NV);
}
#define ALL(K, name) run<K, 8, 0, 4>(name " alone x8"); run<K, 8, 0, 8>(name " alone x8"); run<K, 8, 1, 4>("1 mfma + 8 " name); run<K, 8, 1, 8>("1 mfma + 8 " name); run<K, 16, 1, 8>("1 mfma + 16 " name);
int main() {
  ALL(K_ADD, "v_add_f32")
  ALL(K_CND_VCC, "v_cndmask vcc")
  ALL(K_CND_SGPR, "v_cndmask_e64 sgpr")
  ALL(K_AND, "v_and_b32")
  ALL(K_MUL, "v_mul_f32")
  return 0;
}
