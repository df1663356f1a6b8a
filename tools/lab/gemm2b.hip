// Lab: 256 x 128 x 32 split-bf16 NT GEMM, 4 waves (2 x 2), wave tile 128 x 64, two workgroups per CU (they drift out of
// phase, so one's MFMA phase covers the other's load/split/store phase and epilogue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = 32;

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
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int VAR>
__global__ void __launch_bounds__(256, 2) gemm2b(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                 float* __restrict__ C, int ldc, int M, int N, int K,
                                                 const float* __restrict__ bias, int tiles_n) {
  constexpr int NBUF = (VAR & 1) ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem* s = reinterpret_cast<Smem*>(smem_raw);
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * BN;
  const int wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int srow = threadIdx.x >> 3, sc4 = (threadIdx.x & 7) * 4;      // 32 rows per pass
  float4 ra[8], rb[4];
  const float* Ap = A + (size_t)(m0 + srow) * lda + sc4;
  const float* Bp = B + (size_t)(n0 + srow) * ldb + sc4;
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) ra[p] = *reinterpret_cast<const float4*>(Ap + (size_t)(32 * p) * lda + k0);
#pragma unroll
    for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const float4*>(Bp + (size_t)(32 * p) * ldb + k0);
  };
  auto sstore = [&](Smem& d) {
    bf16x4 h, l;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int o = swz_off(srow + 32 * p, sc4);
      split4(ra[p], h, l);
      *reinterpret_cast<bf16x4*>(&d.a[0][o]) = h;
      *reinterpret_cast<bf16x4*>(&d.a[1][o]) = l;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int o = swz_off(srow + 32 * p, sc4);
      split4(rb[p], h, l);
      *reinterpret_cast<bf16x4*>(&d.b[0][o]) = h;
      *reinterpret_cast<bf16x4*>(&d.b[1][o]) = l;
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mma = [&](const Smem& t) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2][4], bfr[2][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bfr[tt][j] = *reinterpret_cast<const bf16x8*>(&t.b[tt][swz_off(wc * 64 + j * 32 + l31, ks * 16 + 8 * half)]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          af[tt][i] = *reinterpret_cast<const bf16x8*>(&t.a[tt][swz_off(wr * 128 + i * 32 + l31, ks * 16 + 8 * half)]);
      }
#define MMA(TA, TB)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =       \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA][i], bfr[TB][j], acc[i][j], 0, 0, 0);
      MMA(0, 1) MMA(1, 0) MMA(0, 0)
#undef MMA
    }
  };
  const int nk = K / BK;
  gload(0);
  if (NBUF == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      if (kt) lds_barrier();               // the previous tile's fragments are consumed
      sstore(s[0]);
      if (kt + 1 < nk) gload((kt + 1) * BK);
      lds_barrier();
      mma(s[0]);
    }
  } else {
    sstore(s[0]);
    if (nk > 1) gload(BK);
    lds_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      mma(s[kt & 1]);
      if (kt + 1 < nk) sstore(s[(kt + 1) & 1]);
      if (kt + 2 < nk) gload((kt + 2) * BK);
      lds_barrier();
    }
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
        C[(size_t)row * ldc + col] = fmaxf(acc[i][j][r] + bv, 0.f);
      }
  }
}

template <int VAR>
static float run(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int iters) {
  const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
  const size_t lds = ((VAR & 1) ? 2 : 1) * sizeof(Smem);
  hipFuncSetAttribute((const void*)gemm2b<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i)
    hipLaunchKernelGGL((gemm2b<VAR>), dim3(tiles), dim3(256), lds, 0, A, K, B, K, C, N, M, N, K, bias, tiles_n);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((gemm2b<VAR>), dim3(tiles), dim3(256), lds, 0, A, K, B, K, C, N, M, N, K, bias, tiles_n);
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
    printf("%s  %9.1f us  %7.1f TF(alg)  %7.1f TF(exec x3)  maxrelerr %.2e\n", var ? "2 LDS buffers, 1 barrier " : "1 LDS buffer, 2 barriers ", us,
           fl / us / 1e6, 3 * fl / us / 1e6, maxerr);
  }
  return 0;
}
