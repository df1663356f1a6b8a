// Lab: where does the time of the split-bf16 NT GEMM go?  Same kernel as allrank_amd/csrc/ltrx_gemm.hip (128x128x32 tile,
// 4 waves, 2 products hi/lo, 3 MFMAs) with switches that remove one resource at a time.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/gemm_ablate tools/lab/gemm_ablate.hip && tools/lab/gemm_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
enum { A_NONE = 0, A_NO_GLOAD = 1, A_NO_LDSWRITE = 2, A_NO_MFMA = 4, A_NO_EPILOGUE = 8, A_NO_DSREAD = 16, A_NO_SPLIT = 32 };

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ int swz_off(int row, int k) {
  const int c = (k >> 3) ^ ((row >> 2) & 3);
  return row * BK + c * 8 + (k & 7);
}
__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}
struct Smem {
  __bf16 a[2][BM * BK];
  __bf16 b[2][BN * BK];
};

template <int ABL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3)))
gemm_nt(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc, int M, int N,
        int K, const float* __restrict__ bias, int tiles_n) {
  __shared__ __attribute__((aligned(16))) Smem s;
  const int id = blockIdx.x;
  const int m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * BN;
  const int wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int srow = threadIdx.x >> 3, sc4 = (threadIdx.x & 7) * 4;
  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
    const int k = k0 + sc4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = srow + 32 * p;
      if (ABL & A_NO_GLOAD) {
        ra[p] = make_float4(1.f + k, 2.f, 3.f, 4.f + r);
        rb[p] = make_float4(1.f, 2.f + r, 3.f + k, 4.f);
      } else {
        ra[p] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + r) * lda + k);
        rb[p] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + r) * ldb + k);
      }
    }
  };
  auto sstore = [&]() {
    if (ABL & A_NO_LDSWRITE) return;
    bf16x4 h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int o = swz_off(srow + 32 * p, sc4);
      if (ABL & A_NO_SPLIT) {
        *reinterpret_cast<float2*>(&s.a[0][o]) = make_float2(ra[p].x, ra[p].y);
        *reinterpret_cast<float2*>(&s.a[1][o]) = make_float2(ra[p].z, ra[p].w);
        *reinterpret_cast<float2*>(&s.b[0][o]) = make_float2(rb[p].x, rb[p].y);
        *reinterpret_cast<float2*>(&s.b[1][o]) = make_float2(rb[p].z, rb[p].w);
      } else {
        split4(ra[p], h, l);
        *reinterpret_cast<bf16x4*>(&s.a[0][o]) = h;
        *reinterpret_cast<bf16x4*>(&s.a[1][o]) = l;
        split4(rb[p], h, l);
        *reinterpret_cast<bf16x4*>(&s.b[0][o]) = h;
        *reinterpret_cast<bf16x4*>(&s.b[1][o]) = l;
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int nk = K / BK;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[2][2], bfr[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (ABL & A_NO_DSREAD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              af[t][i][e] = (__bf16)(float)(lane + e + kt);
              bfr[t][i][e] = (__bf16)(float)(lane - e + ks);
            }
          } else {
            af[t][i] = *reinterpret_cast<const bf16x8*>(&s.a[t][swz_off(wr * 64 + i * 32 + l31, ks * 16 + 8 * half)]);
            bfr[t][i] = *reinterpret_cast<const bf16x8*>(&s.b[t][swz_off(wc * 64 + i * 32 + l31, ks * 16 + 8 * half)]);
          }
        }
      if (ABL & A_NO_MFMA) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j][0] += (float)af[0][i][0] * (float)bfr[1][j][1] + (float)af[1][i][2] * (float)bfr[0][j][3];
      } else {
#define MMA(TA, TB)                                                                                                \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =         \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA][i], bfr[TB][j], acc[i][j], 0, 0, 0);
        MMA(0, 1) MMA(1, 0) MMA(0, 0)
#undef MMA
      }
    }
  }
  if (ABL & A_NO_EPILOGUE) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 12345.678f) C[0] = t;
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + l31;
    const float bv = bias[col];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + rowmap(r, half);
        C[(size_t)row * ldc + col] = fmaxf(acc[i][j][r] + bv, 0.f);
      }
  }
}

template <int ABL>
static float run(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int iters) {
  const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_nt<ABL>), dim3(tiles), dim3(256), 0, 0, A, K, B, K, C, N, M, N, K, bias, tiles_n);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_nt<ABL>), dim3(tiles), dim3(256), 0, 0, A, K, B, K, C, N, M, N, K, bias, tiles_n);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 61440, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
  float *A, *B, *C, *bias;
  hipMalloc(&A, (size_t)M * K * 4);
  hipMalloc(&B, (size_t)N * K * 4);
  hipMalloc(&C, (size_t)M * N * 4);
  hipMalloc(&bias, (size_t)N * 4);
  std::vector<float> h((size_t)M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, (size_t)N * 4);
  const double fl = 2.0 * M * N * K;
#define RUN(name, abl)                                                                              \
  {                                                                                                 \
    float us = run<abl>(A, B, C, bias, M, N, K, 10);                                                \
    printf("%-34s %9.1f us  %7.1f TF(alg)  %7.1f TF(exec x3)\n", name, us, fl / us / 1e6, 3 * fl / us / 1e6); \
  }
  printf("M=%d N=%d K=%d\n", M, N, K);
  RUN("full", A_NONE)
  RUN("no epilogue stores", A_NO_EPILOGUE)
  RUN("no global loads", A_NO_GLOAD)
  RUN("no split (raw store)", A_NO_SPLIT)
  RUN("no LDS writes", A_NO_LDSWRITE)
  RUN("no ds_reads", A_NO_DSREAD)
  RUN("no MFMA", A_NO_MFMA)
  RUN("no gload+no epilogue", A_NO_GLOAD | A_NO_EPILOGUE)
  RUN("no gload/ldswrite/epilogue", A_NO_GLOAD | A_NO_LDSWRITE | A_NO_EPILOGUE)
  RUN("MFMA only", A_NO_GLOAD | A_NO_LDSWRITE | A_NO_EPILOGUE | A_NO_DSREAD)
  RUN("all but MFMA", A_NO_MFMA)
  return 0;
}
