// Issue-rate probe (gfx950): cycles per instruction of the VALU / transcendental / conversion / lane-swap / LDS instructions the
// fused stack kernels are made of, alone and interleaved with v_mfma_f32_32x32x16_bf16, with one and with two waves per SIMD.
// One workgroup per CU; s_memtime-free: cycles from clock64() (shader clock) around an unrolled body, minimum over workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/issue.hip -o tools/probe/issue && tools/probe/issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

enum { K_ADD = 0, K_EXP, K_RCP, K_CVT, K_SWAP, K_PKFMA, K_FMA, K_MFMA, K_MFMA_DEP, K_DSR, K_DSW, K_CNDMASK, K_NKINDS };

// NV instructions of kind KIND per MFMA (NM MFMAs per group; NM = 0: VALU only), REP groups per iteration
template <int KIND, int NV, int NM, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void probe(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
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
  unsigned char* lp = lds + (threadIdx.x & 255) * 16 + (threadIdx.x >> 8) * 8192;
  *reinterpret_cast<u32x4*>(lp) = fa;
  __syncthreads();
  u32x4 dsv[4] = {fa, fa, fa, fa};
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 8; g++) {
#pragma unroll
      for (int m = 0; m < NM; m++) {
        if (KIND == K_MFMA_DEP) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[0], 0, 0, 0);
        else acc[(g * NM + m) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[(g * NM + m) & 3], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < NV; k++) {
        const int r = (g * NV + k) & 15;
        if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 5) & 15]));
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(v[(r + 5) & 15]));
        if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
        if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[r]));
        if (KIND == K_CVT) { unsigned o; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(v[r]), "v"(v[(r + 1) & 15])); v[(r + 2) & 15] = __builtin_bit_cast(float, o); }
        if (KIND == K_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[r]), "+v"(v[(r + 8) & 15]));
        if (KIND == K_PKFMA) {
          f32x2 p = {v[(2 * r) & 15], v[(2 * r + 1) & 15]};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
          v[(2 * r) & 15] = p[0]; v[(2 * r + 1) & 15] = p[1];
        }
        if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[r]) : "v"(v[(r + 5) & 15]));
        if (KIND == K_DSR) asm volatile("ds_read_b128 %0, %1" : "=v"(dsv[k & 3]) : "v"((unsigned)(size_t)lp));
        if (KIND == K_DSW) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)lp), "v"(dsv[k & 3]));
      }
    }
    if (KIND == K_DSR || KIND == K_DSW) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  if (KIND == K_DSR || KIND == K_DSW) asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[a][r];
  s += __builtin_bit_cast(float, dsv[0][0] + dsv[1][1] + dsv[2][2] + dsv[3][3]);
  sink[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NV, int NM, int WAVES>
void run(const char* name) {
  static unsigned long long* d = nullptr; static float* sink = nullptr;
  if (!d) { hipMalloc(&d, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4); }
  const int iters = 64;
  probe<KIND, NV, NM, WAVES><<<256, 64 * WAVES>>>(d, sink, iters);
  hipDeviceSynchronize();
  probe<KIND, NV, NM, WAVES><<<256, 64 * WAVES>>>(d, sink, iters);
  hipDeviceSynchronize();
  unsigned long long h[256 * 16];
  hipMemcpy(h, d, 256 * WAVES * 8, hipMemcpyDeviceToHost);
  double mx = 0, mn = 1e30, av = 0;
  for (int i = 0; i < 256 * WAVES; i++) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; av += h[i]; }
  av /= 256 * WAVES;
  const double groups = iters * 8.0;
  printf("%-44s waves/CU %2d: %8.1f cycles per group (min %.1f max %.1f)  [group = %d MFMA + %d x op]\n", name, WAVES, av / groups, mn / groups, mx / groups, NM, NV);
}

#define ALONE(K, name) run<K, 8, 0, 4>(name " alone x8"); run<K, 8, 0, 8>(name " alone x8"); run<K, 8, 0, 16>(name " alone x8");
int main() {
  ALONE(K_ADD, "v_add_f32")
  ALONE(K_FMA, "v_fma_f32")
  ALONE(K_PKFMA, "v_pk_fma_f32")
  ALONE(K_EXP, "v_exp_f32")
  ALONE(K_RCP, "v_rcp_f32")
  ALONE(K_CVT, "v_cvt_pk_bf16_f32")
  ALONE(K_SWAP, "v_permlane32_swap")
  ALONE(K_CNDMASK, "v_cndmask_b32")
  ALONE(K_DSR, "ds_read_b128")
  ALONE(K_DSW, "ds_write_b128")
  run<K_MFMA, 0, 1, 4>("mfma 32x32x16 bf16, 4 accumulators");
  run<K_MFMA, 0, 1, 8>("mfma 32x32x16 bf16, 4 accumulators");
  run<K_MFMA_DEP, 0, 1, 4>("mfma 32x32x16 bf16, dependent chain");
  run<K_MFMA_DEP, 0, 1, 8>("mfma 32x32x16 bf16, dependent chain");
  // one MFMA + N VALU per group
  run<K_ADD, 4, 1, 4>("1 mfma + 4 v_add"); run<K_ADD, 8, 1, 4>("1 mfma + 8 v_add"); run<K_ADD, 12, 1, 4>("1 mfma + 12 v_add"); run<K_ADD, 16, 1, 4>("1 mfma + 16 v_add");
  run<K_ADD, 4, 1, 8>("1 mfma + 4 v_add"); run<K_ADD, 8, 1, 8>("1 mfma + 8 v_add"); run<K_ADD, 12, 1, 8>("1 mfma + 12 v_add"); run<K_ADD, 16, 1, 8>("1 mfma + 16 v_add");
  run<K_EXP, 2, 1, 4>("1 mfma + 2 v_exp"); run<K_EXP, 4, 1, 4>("1 mfma + 4 v_exp"); run<K_EXP, 8, 1, 4>("1 mfma + 8 v_exp");
  run<K_EXP, 2, 1, 8>("1 mfma + 2 v_exp"); run<K_EXP, 4, 1, 8>("1 mfma + 4 v_exp"); run<K_EXP, 8, 1, 8>("1 mfma + 8 v_exp");
  run<K_PKFMA, 8, 1, 4>("1 mfma + 8 v_pk_fma"); run<K_PKFMA, 8, 1, 8>("1 mfma + 8 v_pk_fma");
  run<K_CVT, 8, 1, 8>("1 mfma + 8 cvt_pk"); run<K_SWAP, 8, 1, 8>("1 mfma + 8 permlane32_swap");
  run<K_DSR, 1, 1, 8>("1 mfma + 1 ds_read_b128"); run<K_DSR, 2, 1, 8>("1 mfma + 2 ds_read_b128"); run<K_DSR, 4, 1, 8>("1 mfma + 4 ds_read_b128");
  run<K_DSW, 1, 1, 8>("1 mfma + 1 ds_write_b128"); run<K_DSW, 2, 1, 8>("1 mfma + 2 ds_write_b128");
  run<K_MFMA_DEP, 0, 1, 8>("dependent mfma chain, 2 waves/SIMD"); 
  return 0;
}
