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
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cyc, int iters, float seed) {
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
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NCH, int NVALU, bool DO_MFMA>
static void run(const char* name, int threads, int blocks, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((probe<NCH, NVALU, DO_MFMA>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((probe<NCH, NVALU, DO_MFMA>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int per_it = (8 / NCH) * NCH;      // MFMA slots per iteration
  printf("%-44s threads %3d blocks %4d: %7.1f cycles per slot (MFMA%s + %d VALU)\n", name, threads, blocks, (double)c / iters / per_it, DO_MFMA ? "" : " off", NVALU);
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 4096 * 512 * 4);
  hipMalloc(&cyc, 8);
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int threads = cfg == 1 ? 512 : 256, blocks = cfg == 2 ? 256 : 1;
    printf("--- %d waves per SIMD, %d workgroup(s)\n", threads / 256, blocks);
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
