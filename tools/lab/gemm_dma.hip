// Lab: 256 x 256 x 32 split-bf16 NT GEMM with LDS-DMA staging: raw fp32 tiles go global -> LDS directly
// (global_load_lds_dwordx4, no VGPR round trip, no ds_write), each wave splits its fragments into hi/lo bf16 in registers
// right before the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BK = 32;

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
// raw fp32 tile [256 rows][32 floats]; the 16-byte chunk c of row r lives at chunk position c ^ ((r >> 1) & 7)
struct Stage {
  float a[256 * BK];
  float b[256 * BK];
};
__device__ __forceinline__ void split8(const f32x4 lo4, const f32x4 hi4, bf16x8& h, bf16x8& l) {
  const float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)x[e];
    h[e] = hh;
    l[e] = (__bf16)(x[e] - (float)hh);
  }
}

template <int VAR>
__global__ void __launch_bounds__(512) gemm_dma(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                float* __restrict__ C, int ldc, int M, int N, int K,
                                                const float* __restrict__ bias, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Stage* s = reinterpret_cast<Stage*>(smem_raw);
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (id / tiles_n) * 256, n0 = (id % tiles_n) * 256;
  const int wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  // DMA map: instruction p of wave w covers tile rows 32 w + 8 p .. + 7; lane -> (row = lane >> 3, position = lane & 7)
  const int drow = 32 * wave + (lane >> 3);
  const int dpos = lane & 7;
  const float* Ag[4];
  const float* Bg[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = drow + 8 * p;
    const int chunk = dpos ^ ((row >> 1) & 7);
    Ag[p] = A + (size_t)(m0 + row) * lda + 4 * chunk;
    Bg[p] = B + (size_t)(n0 + row) * ldb + 4 * chunk;
  }
  auto dma = [&](Stage& d, int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(Ag[p] + k0),
                                       (void __attribute__((address_space(3)))*)(d.a + (32 * wave + 8 * p) * BK), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(Bg[p] + k0),
                                       (void __attribute__((address_space(3)))*)(d.b + (32 * wave + 8 * p) * BK), 16, 0, 0);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fragment (row, ks): floats [ks*16 + 8*half, +8) = chunks 4 ks + 2 half and + 1
  auto frag = [&](const float* img, int row, int ks, bf16x8& h, bf16x8& l) {
    const int sw = (row >> 1) & 7;
    const int c0 = 4 * ks + 2 * half;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(img + row * BK + ((c0 ^ sw) << 2));
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(img + row * BK + (((c0 + 1) ^ sw) << 2));
    split8(v0, v1, h, l);
  };
  auto mma = [&](const Stage& t) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) frag(t.b, wc * 64 + j * 32 + l31, ks, bh[j], bl[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) frag(t.a, wr * 128 + i * 32 + l31, ks, ah[i], al[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };
  const int nk = K / BK;
  dma(s[0], 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) dma(s[(kt + 1) & 1], (kt + 1) * BK);
    mma(s[kt & 1]);
    __syncthreads();                   // vmcnt(0) + barrier: tile kt+1 has landed, tile kt is consumed
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + l31;
    const float bv = bias[col];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 128 + i * 32 + rowmap(r, half);
        if (VAR & 1) __builtin_nontemporal_store(fmaxf(acc[i][j][r] + bv, 0.f), &C[(size_t)row * ldc + col]);
        else C[(size_t)row * ldc + col] = fmaxf(acc[i][j][r] + bv, 0.f);
      }
  }
}

template <int VAR>
static float run(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int iters) {
  const int tiles_n = N / 256, tiles = (M / 256) * tiles_n;
  hipFuncSetAttribute((const void*)gemm_dma<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sizeof(Stage));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i)
    hipLaunchKernelGGL((gemm_dma<VAR>), dim3(tiles), dim3(512), 2 * sizeof(Stage), 0, A, K, B, K, C, N, M, N, K, bias, tiles_n);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((gemm_dma<VAR>), dim3(tiles), dim3(512), 2 * sizeof(Stage), 0, A, K, B, K, C, N, M, N, K, bias, tiles_n);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("HIP error %s\n", hipGetErrorString(e));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 61440, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
  float *A, *B, *C, *bias;
  hipMalloc(&A, (size_t)M * K * 4);
  hipMalloc(&B, (size_t)N * K * 4);
  hipMalloc(&C, (size_t)M * N * 4);
  hipMalloc(&bias, (size_t)N * 4);
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K), hbias(N);
  for (size_t i = 0; i < ha.size(); ++i) ha[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  for (size_t i = 0; i < hb.size(); ++i) hb[i] = (float)((i * 40503u + 7) % 1999) / 1000.f - 1.f;
  for (int i = 0; i < N; ++i) hbias[i] = 0.01f * (i % 13);
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, hbias.data(), (size_t)N * 4, hipMemcpyHostToDevice);
  const double fl = 2.0 * M * N * K;
  printf("M=%d N=%d K=%d\n", M, N, K);
  for (int var = 0; var < 2; ++var) {
    hipMemset(C, 0, (size_t)M * N * 4);
    float us = var == 0 ? run<0>(A, B, C, bias, M, N, K, 10) : run<1>(A, B, C, bias, M, N, K, 10);
    std::vector<float> hc((size_t)256 * N);
    hipMemcpy(hc.data(), C + (size_t)(M - 256) * N, hc.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 256; ++t) {
      const int r = (t * 37) % 256, c = (t * 101) % N;
      double ref = hbias[c];
      for (int k = 0; k < K; ++k) ref += (double)ha[(size_t)(M - 256 + r) * K + k] * hb[(size_t)c * K + k];
      ref = ref > 0 ? ref : 0;
      maxerr = fmax(maxerr, fabs(ref - hc[(size_t)r * N + c]) / (1 + fabs(ref)));
    }
    printf("%s  %9.1f us  %7.1f TF(alg)  %7.1f TF(exec x3)  maxrelerr %.2e\n", var ? "LDS-DMA, nt stores " : "LDS-DMA            ", us,
           fl / us / 1e6, 3 * fl / us / 1e6, maxerr);
  }
  return 0;
}
