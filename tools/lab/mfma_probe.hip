// round 6 lab probe: issue / dependency behaviour of v_mfma_f32_32x32x16_bf16 on gfx950 inside ONE wave (and with two waves per SIMD):
// cycles per MFMA for 1 / 2 / 4 / 8 independent accumulator chains, alone and with k independent VALU instructions after every MFMA,
// and the VALU-only loop.  build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/lab/mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int NCH, int NVALU, bool DO_MFMA>
__global__ void __launch_bounds__(1024) probe(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x16 acc[NCH];
  for (int c = 0; c < NCH; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = seed * (c + r);
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(seed + e + threadIdx.x);
    b[e] = (__bf16)(seed - e);
  }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8 / (NCH > 8 ? 8 : NCH) * (NCH > 8 ? 1 : 1); ++rep) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (DO_MFMA) acc[c] = MFMA(a, b, acc[c]);
#pragma unroll
        for (int k = 0; k < NVALU; ++k) v[k % 8] = v[k % 8] * 1.0001f + 0.5f;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < NCH; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
    cyc[2 * (threadIdx.x >> 6)] = t0;
    cyc[2 * (threadIdx.x >> 6) + 1] = t1;
  }
}

template <int NCH, int NVALU, bool DO_MFMA>
static void run(const char* name, int threads, int blocks, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((probe<NCH, NVALU, DO_MFMA>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((probe<NCH, NVALU, DO_MFMA>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  unsigned long long c[16];
  hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  const int per_it = (8 / NCH) * NCH, nw = threads / 64;      // MFMA slots per iteration
  unsigned long long lo = c[0], hi = c[1];
  for (int w = 0; w < nw; ++w) {
    if (c[2 * w] < lo) lo = c[2 * w];
    if (c[2 * w + 1] > hi) hi = c[2 * w + 1];
  }
  // wave 0 alone (the oldest wave wins the issue arbitration) and the whole workgroup (first start to last end)
  printf("%-30s threads %3d blocks %4d: wave 0 %6.1f, workgroup %6.1f cycles per slot and wave (MFMA%s + %d VALU); per SIMD: %6.1f per slot\n", name, threads, blocks,
         (double)(c[1] - c[0]) / iters / per_it, (double)(hi - lo) / iters / per_it, DO_MFMA ? "" : " off", NVALU, (double)(hi - lo) / iters / per_it / (nw / 4));
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 4096 * 512 * 4);
  hipMalloc(&cyc, 16 * 8);
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int threads = cfg == 1 ? 512 : (cfg == 2 ? 1024 : 256), blocks = 1;
    printf("--- %d wave(s) per SIMD\n", threads / 256);
    run<1, 0, true>("1 chain", threads, blocks, out, cyc);
    run<2, 0, true>("2 chains", threads, blocks, out, cyc);
    run<4, 0, true>("4 chains", threads, blocks, out, cyc);
    run<8, 0, true>("8 chains", threads, blocks, out, cyc);
    run<4, 4, true>("4 chains + 4 VALU", threads, blocks, out, cyc);
    run<4, 7, true>("4 chains + 7 VALU", threads, blocks, out, cyc);
    run<4, 10, true>("4 chains + 10 VALU", threads, blocks, out, cyc);
    run<2, 7, true>("2 chains + 7 VALU", threads, blocks, out, cyc);
    run<4, 7, false>("VALU only, 7 per slot", threads, blocks, out, cyc);
    run<4, 10, false>("VALU only, 10 per slot", threads, blocks, out, cyc);
  }
  return 0;
}
