// Empirical semantics of ds_read_b64_tr_b16 on gfx950 (run on the GPU box):
// LDS holds a row-major [rows][RS/2] bf16 matrix whose element value = its index.
// Lane l (group i=l&15) supplies address of row (i>>2), 4-element column chunk (i&3).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(const int* lane_byte_off, int* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) v4s*)((__attribute__((address_space(3))) char*)lds + lane_byte_off[threadIdx.x]));
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = (int)(uint16_t)r[j];
}
int main() {
  const int RS = 144;  // row stride in bytes (72 elements)
  int h_off[64], h_out[256], *d_off, *d_out;
  for (int l = 0; l < 64; l++) {
    int i = l & 15, g = l >> 4;
    int row = (g >> 1) * 8 + (i >> 2);          // k-group (lanes 32-63) starts 8 rows later
    int col = (g & 1) * 16 + (i & 3) * 4;       // second 16-lane group: channels 16..31
    h_off[l] = row * RS + col * 2;
  }
  hipMalloc(&d_off, sizeof(h_off)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_off, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; l++) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; j++) {
      int e = h_out[l * 4 + j], row = e / (RS / 2), col = e % (RS / 2);
      printf(" (r%d,c%d)", row, col);
      int g = l >> 4, exp_row = (g >> 1) * 8 + j, exp_col = (g & 1) * 16 + (l & 15);
      if (row != exp_row || col != exp_col) ok = 0;
    }
    printf("\n");
  }
  printf("HYPOTHESIS %s: lane gets rows j=0..3 of column (l&15) of its 16-lane group's 4x16 tile\n", ok ? "CONFIRMED" : "REFUTED");
  return 0;
}
