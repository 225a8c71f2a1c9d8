// HBM write / read efficiency of the plane access patterns of the fused stack kernels (gfx950).
//   pattern 0: per wave instruction 32 rows x 32 B (two lanes per row, rows 128 B apart)   - [N,64] bf16 planes, 16-byte pieces
//   pattern 1: per wave instruction 1 KB contiguous                                           - blocked "lane-record" planes
// Every workgroup (512 threads) owns 192 rows per "block" of 8, writes 4 planes (128 B per row) per block like the forward,
// or reads 2 + writes 3 like the data-gradient chain.  hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_pattern.hip -o tools/probe/hbm_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int NREAD, int NWRITE>
__global__ __launch_bounds__(512) void k(unsigned char* base, long plane_bytes, int rows_per_wg, int nblocks) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * rows_per_wg;
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  u32x4 acc = {0, 0, 0, 0};
  for (int blk = 0; blk < nblocks; blk++) {
    for (int pl = 0; pl < NREAD + NWRITE; pl++) {
      unsigned char* plane = base + ((long)blk * (NREAD + NWRITE) + pl) * plane_bytes;
      // a wave covers 32 rows x 128 B = 4 KB of a plane with 4 instructions; 8 waves x ... rows_per_wg rows
      for (int r = wave * 32; r < rows_per_wg; r += 8 * 32) {
        for (int i = 0; i < 4; i++) {
          long off;
          if (PATTERN == 0) off = (row0 + r + (lane & 31)) * 128 + i * 32 + (lane >> 5) * 16;
          else off = (row0 + r) * 128 + i * 1024 + lane * 16;
          if (pl < NREAD) { u32x4 t = *reinterpret_cast<const u32x4*>(plane + off); acc += t; }
          else *reinterpret_cast<u32x4*>(plane + off) = v;
        }
      }
    }
  }
  if (acc[0] == 0x12345678u) base[0] = 1;
}

template <int PATTERN, int NREAD, int NWRITE>
void run(const char* name, unsigned char* d, long plane_bytes, int rows_per_wg, int nblocks, int nwg) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<PATTERN, NREAD, NWRITE><<<nwg, 512>>>(d, plane_bytes, rows_per_wg, nblocks);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; i++) k<PATTERN, NREAD, NWRITE><<<nwg, 512>>>(d, plane_bytes, rows_per_wg, nblocks);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)nwg * rows_per_wg * 128.0 * (NREAD + NWRITE) * nblocks;
  printf("%-60s %7.1f us  %6.2f TB/s (%.0f MB)\n", name, ms * 1e3 / 5, bytes / (ms * 1e-3 / 5) / 1e12, bytes / 1e6);
}
int main() {
  const int nwg = 256, rows = 128, nblocks = 8;   // 32768 rows: the benchmark's 32 000 frames
  const long plane_bytes = (long)nwg * rows * 128;
  unsigned char* d; hipMalloc(&d, plane_bytes * 8 * 5 + 4096); hipMemset(d, 0, plane_bytes * 8 * 5);
  run<0, 0, 4>("forward-like: 4 planes written, 16-byte pieces (32 rows x 32 B)", d, plane_bytes, rows, nblocks, nwg);
  run<1, 0, 4>("forward-like: 4 planes written, 1 KB contiguous", d, plane_bytes, rows, nblocks, nwg);
  run<0, 2, 3>("chain-like: 2 read + 3 written, 16-byte pieces", d, plane_bytes, rows, nblocks, nwg);
  run<1, 2, 3>("chain-like: 2 read + 3 written, 1 KB contiguous", d, plane_bytes, rows, nblocks, nwg);
  run<0, 5, 0>("5 planes read, 16-byte pieces", d, plane_bytes, rows, nblocks, nwg);
  run<1, 5, 0>("5 planes read, 1 KB contiguous", d, plane_bytes, rows, nblocks, nwg);
  return 0;
}
