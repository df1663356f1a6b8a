// Lab: per-CU streaming-load throughput as a function of bytes in flight.  One 512-thread workgroup per CU reads its own
// slice of a big buffer with global_load_dwordx4, U loads in flight per thread, rows x 128-byte pattern like the GEMM tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int U, int THREADS>
__global__ void __launch_bounds__(THREADS) stream(const float4* __restrict__ src, size_t per_block_f4, float* __restrict__ out,
                                                  int pattern, size_t total_f4) {
  const float4* p = src + (pattern == 2 ? 0 : (size_t)blockIdx.x * per_block_f4);
  float4 acc = make_float4(0, 0, 0, 0);
  const size_t iters = per_block_f4 / (THREADS * U);
  for (size_t it = 0; it < iters; ++it) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t idx;
      if (pattern == 0) idx = (it * U + u) * THREADS + threadIdx.x;                       // 1 KB contiguous per wave-instr
      else idx = (((it * U + u) * (THREADS / 8) + (threadIdx.x >> 3)) * 128 + (threadIdx.x & 7)) % per_block_f4;  // 8 rows x 128 B, row stride 2 KB
      v[u] = p[idx];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int U, int THREADS>
static void run(const float4* src, size_t total_f4, float* out, int blocks, int pattern, const char* name) {
  const size_t per_block = total_f4 / blocks / (THREADS * U) * (THREADS * U);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((stream<U, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, src, per_block, out, pattern, total_f4);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream<U, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, src, per_block, out, pattern, total_f4);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)per_block * 16 * blocks * 5;
  printf("%-46s U=%2d threads=%4d blocks=%4d: %7.2f TB/s  (%5.1f GB/s per block, %5.1f KB in flight per block)\n", name, U, THREADS, blocks,
         bytes / ms / 1e9, bytes / ms / 1e6 / blocks, U * THREADS * 16 / 1024.0);
}

int main() {
  const size_t total = (size_t)1 << 30;      // 1 GiB
  float4* src;
  float* out;
  hipMalloc(&src, total);
  hipMalloc(&out, 64);
  hipMemset(src, 0, total);
  const size_t f4 = total / 16;
  run<8, 512>(src, f4, out, 256, 0, "HBM stream, contiguous");
  run<8, 512>(src, f4, out, 256, 1, "HBM stream, 8 rows x 128 B");
  run<16, 512>(src, f4, out, 256, 0, "HBM stream, contiguous");
  run<8, 1024>(src, f4, out, 256, 0, "HBM stream, contiguous");
  run<8, 512>(src, f4, out, 512, 0, "HBM stream, 2 blocks/CU");
  run<8, 512>(src, f4, out, 1024, 0, "HBM stream, 4 blocks/CU");
  // L2/MALL-resident: every block re-reads the same 2 MB (pattern 2 ignores blockIdx)
  const size_t small = (size_t)(2 << 20) / 16 * 256;   // per_block = 2 MB
  run<8, 512>(src, small, out, 256, 2, "all blocks read the same 2 MB (L2 hits)");
  run<16, 512>(src, small, out, 256, 2, "all blocks read the same 2 MB (L2 hits)");
  run<8, 1024>(src, small, out, 256, 2, "all blocks read the same 2 MB (L2 hits)");
  run<8, 512>(src, small * 2, out, 512, 2, "same 2 MB, 2 blocks/CU");
  run<4, 512>(src, small, out, 256, 2, "all blocks read the same 2 MB (L2 hits)");
  run<2, 512>(src, small, out, 256, 2, "all blocks read the same 2 MB (L2 hits)");
  return 0;
}
