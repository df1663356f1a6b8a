// Lab: weight-gradient GEMM  C[NP,KP] = A[M,NP]^T B[M,KP]  (split-bf16), 256 x 256 output tile, 8 waves, split-K over M.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32;

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
#ifdef MAP2
__device__ __forceinline__ int swz_off(int row, int k) {      // + row-pair flip by bit 4: rows r and r+16 swap bank halves
  const int c = (k >> 3) ^ ((row >> 2) & 3);
  return (row ^ ((row >> 4) & 1)) * BK + c * 8 + (k & 7);
}
#else
__device__ __forceinline__ int swz_off(int row, int k) {
  const int c = (k >> 3) ^ ((row >> 2) & 3);
  return row * BK + c * 8 + (k & 7);
}
#endif
__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}
// operand images [term][row = output index][k = contraction] bf16; the B image starts 64 B later so that an A-writer and a
// B-writer of the same 16-lane store group land in different bank halves
struct SmemT {
  __bf16 a[2][256 * BK];
  __bf16 pad[32];
  __bf16 b[2][256 * BK];
  __bf16 pad2[32];
};
__device__ __forceinline__ void lds_only_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__global__ void __launch_bounds__(512) gemm_tn256(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                  float* __restrict__ slabs, float* __restrict__ bias_slabs, int M, int NP, int KP,
                                                  int tiles_k, int m_per_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SmemT* s = reinterpret_cast<SmemT*>(smem_raw);
#ifdef XCD
  const int ntiles = (NP / 256) * tiles_k;
  int idr;
  {
    const int n = gridDim.x, id = blockIdx.x;
    const int q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
    idr = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tile = idr % ntiles, split = idr / ntiles;
#else
  const int tile = blockIdx.x, split = blockIdx.y;
#endif
  const int n0 = (tile / tiles_k) * 256, k0 = (tile % tiles_k) * 256;
  const int wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int mbeg = split * m_per_split, mend = min(M, mbeg + m_per_split);
  // staging: thread -> (mg = 4 contraction rows, op = which operand, cg = two groups of 4 output columns)
#ifdef MAP2
  // staging: 8 consecutive lanes cover 8 column groups = 128 contiguous bytes of one contraction row; a thread owns 4 rows x
  // 4 columns of A and of B (g = operand)
#ifdef MAP3
  const int cg = threadIdx.x & 63, mg = threadIdx.x >> 6;           // a wave-instruction = 1 KB of ONE contraction row
#else
  const int cg = (threadIdx.x & 7) + 8 * (threadIdx.x >> 6), mg = (threadIdx.x >> 3) & 7;
#endif
  const float* PA = A + n0 + (size_t)(4 * mg) * lda + 4 * cg;
  const float* PB = B + k0 + (size_t)(4 * mg) * ldb + 4 * cg;
  float4 r[2][4];                    // [operand][m row]
  auto gload = [&](int mt) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#ifdef NOGLOAD
      r[0][e] = make_float4(1.f + mt, 2.f + e, 3.f, 4.f);
      r[1][e] = make_float4(1.f, 2.f + mt, 3.f + e, 4.f);
#else
      r[0][e] = *reinterpret_cast<const float4*>(PA + (size_t)(mt + e) * lda);
      r[1][e] = *reinterpret_cast<const float4*>(PB + (size_t)(mt + e) * ldb);
#endif
    }
  };
  float bsum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const bool want_bias = bias_slabs != nullptr && (tile % tiles_k) == 0;
  auto sstore = [&](SmemT& d) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      __bf16* img0 = g ? d.b[0] : d.a[0];
      __bf16* img1 = g ? d.b[1] : d.a[1];
      const float cx[4][4] = {{r[g][0].x, r[g][1].x, r[g][2].x, r[g][3].x}, {r[g][0].y, r[g][1].y, r[g][2].y, r[g][3].y},
                              {r[g][0].z, r[g][1].z, r[g][2].z, r[g][3].z}, {r[g][0].w, r[g][1].w, r[g][2].w, r[g][3].w}};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x4 h, l;
        split4(make_float4(cx[c][0], cx[c][1], cx[c][2], cx[c][3]), h, l);
        const int o = swz_off(4 * cg + c, 4 * mg);
        *reinterpret_cast<bf16x4*>(&img0[o]) = h;
        *reinterpret_cast<bf16x4*>(&img1[o]) = l;
        if (want_bias && g == 0) bsum[0][c] += (cx[c][0] + cx[c][1]) + (cx[c][2] + cx[c][3]);
      }
    }
  };
#else
  const int mg = threadIdx.x & 7, op = (threadIdx.x >> 3) & 1, cg = threadIdx.x >> 4;      // cg 0..31 (and cg + 32)
  const float* P = (op ? B + k0 : A + n0) + (size_t)(4 * mg) * (op ? ldb : lda) + 4 * cg;
  const int ld = op ? ldb : lda;
  float4 r[2][4];                    // [column group][m row]
  auto gload = [&](int mt) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) r[g][e] = *reinterpret_cast<const float4*>(P + (size_t)(mt + e) * ld + 128 * g);
  };
  float bsum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const bool want_bias = bias_slabs != nullptr && (tile % tiles_k) == 0 && op == 0;
  auto sstore = [&](SmemT& d) {
    __bf16* img0 = op ? d.b[0] : d.a[0];
    __bf16* img1 = op ? d.b[1] : d.a[1];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float cx[4][4] = {{r[g][0].x, r[g][1].x, r[g][2].x, r[g][3].x}, {r[g][0].y, r[g][1].y, r[g][2].y, r[g][3].y},
                              {r[g][0].z, r[g][1].z, r[g][2].z, r[g][3].z}, {r[g][0].w, r[g][1].w, r[g][2].w, r[g][3].w}};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x4 h, l;
        split4(make_float4(cx[c][0], cx[c][1], cx[c][2], cx[c][3]), h, l);
        const int o = swz_off(128 * g + 4 * cg + c, 4 * mg);
        *reinterpret_cast<bf16x4*>(&img0[o]) = h;
        *reinterpret_cast<bf16x4*>(&img1[o]) = l;
        if (want_bias) bsum[g][c] += (cx[c][0] + cx[c][1]) + (cx[c][2] + cx[c][3]);
      }
    }
  };
#endif
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  auto mma = [&](const SmemT& t) {
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
#ifdef NOMMA
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j][0] += (float)af[0][i][0] * (float)bfr[1][j][1] + (float)af[1][i][2] * (float)bfr[0][j][3];
#else
#define MMA(TA, TB)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =       \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA][i], bfr[TB][j], acc[i][j], 0, 0, 0);
      MMA(0, 1) MMA(1, 0) MMA(0, 0)
#undef MMA
#endif
    }
  };
  const int nk = (mend - mbeg) / BK;             // m_per_split and M are multiples of 32
  if (nk > 0) {
    gload(mbeg);
    sstore(s[0]);
    if (nk > 1) gload(mbeg + BK);
    __syncthreads();
    int kt = 0;
#ifdef DEEP
    {
      float4 r2[2][4];
      auto gload_b = [&](int mt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          r2[0][e] = *reinterpret_cast<const float4*>(PA + (size_t)(mt + e) * lda);
          r2[1][e] = *reinterpret_cast<const float4*>(PB + (size_t)(mt + e) * ldb);
        }
      };
      auto swap_in = [&]() {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) r[g][e] = r2[g][e];
      };
      // r holds tile kt+1; r2 <- tile kt+2; each step: store r, r <- r2, r2 <- tile kt+3
      if (nk > 2) gload_b(mbeg + 2 * BK);
      for (; kt + 3 < nk; ++kt) {
        sstore(s[(kt + 1) & 1]);
        swap_in();
        gload_b(mbeg + (kt + 3) * BK);
        mma(s[kt & 1]);
        lds_only_barrier();
      }
      if (kt + 2 < nk) {
        sstore(s[(kt + 1) & 1]);
        swap_in();
        mma(s[kt & 1]);
        lds_only_barrier();
        ++kt;
      }
    }
#endif
    for (; kt + 2 < nk; ++kt) {
#ifdef ORDER2
      sstore(s[(kt + 1) & 1]);
      gload(mbeg + (kt + 2) * BK);
      mma(s[kt & 1]);
#else
      mma(s[kt & 1]);
      sstore(s[(kt + 1) & 1]);
      gload(mbeg + (kt + 2) * BK);
#endif
      lds_only_barrier();
    }
    for (; kt < nk; ++kt) {
      mma(s[kt & 1]);
      if (kt + 1 < nk) sstore(s[(kt + 1) & 1]);
      __syncthreads();
    }
  }
#ifdef MAP2
  if (want_bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = bsum[0][c];
#ifdef MAP3
      __shared__ float bred[8][256];
      bred[mg][4 * cg + c] = v;        // (lab only: not synchronised with the K loop's last barrier -> needs one)
      __syncthreads();
      if (mg == 0) {
        v = 0.f;
        for (int q = 0; q < 8; ++q) v += bred[q][4 * cg + c];
        bias_slabs[(size_t)split * NP + n0 + 4 * cg + c] = v;
      }
      __syncthreads();
#else
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (mg == 0) bias_slabs[(size_t)split * NP + n0 + 4 * cg + c] = v;
#endif
    }
  }
#else
  if (want_bias) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = bsum[g][c];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        if (mg == 0) bias_slabs[(size_t)split * NP + n0 + 128 * g + 4 * cg + c] = v;
      }
  }
#endif
  float* slab = slabs + (size_t)split * NP * KP;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = k0 + wc * 64 + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = n0 + wr * 128 + i * 32 + rowmap(q, half);
        slab[(size_t)row * KP + col] = acc[i][j][q];
      }
  }
}

__global__ void slab_reduce(const float* __restrict__ slabs, int splits, size_t n, float* __restrict__ C) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = 0.f;
    for (int sidx = 0; sidx < splits; ++sidx) a += slabs[(size_t)sidx * n + i];
    C[i] = a;
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 61440, NP = argc > 2 ? atoi(argv[2]) : 2048, KP = argc > 3 ? atoi(argv[3]) : 512;
  const int target = argc > 4 ? atoi(argv[4]) : 512;
  float *A, *B, *C, *ws, *gb;
  hipMalloc(&A, (size_t)M * NP * 4);
  hipMalloc(&B, (size_t)M * KP * 4);
  hipMalloc(&C, (size_t)NP * KP * 4);
  hipMalloc(&gb, (size_t)NP * 4);
  std::vector<float> ha((size_t)M * NP), hb((size_t)M * KP);
  for (size_t i = 0; i < ha.size(); ++i) ha[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  for (size_t i = 0; i < hb.size(); ++i) hb[i] = (float)((i * 40503u + 7) % 1999) / 1000.f - 1.f;
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  const int tiles_k = KP / 256, tiles = (NP / 256) * tiles_k;
  int splits = (target + tiles - 1) / tiles;
  int mps = ((M + splits - 1) / splits + 31) / 32 * 32;
  splits = (M + mps - 1) / mps;
  hipMalloc(&ws, ((size_t)splits * NP * KP + (size_t)splits * NP) * 4);
  float* bsl = ws + (size_t)splits * NP * KP;
  hipFuncSetAttribute((const void*)gemm_tn256, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sizeof(SmemT));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
#ifdef XCD
  const dim3 GRID(tiles * splits);
#else
  const dim3 GRID(tiles, splits);
#endif
  auto launch = [&]() {
    hipLaunchKernelGGL(gemm_tn256, GRID, dim3(512), 2 * sizeof(SmemT), 0, A, NP, B, KP, ws, bsl, M, NP, KP, tiles_k, mps);
    hipLaunchKernelGGL(slab_reduce, dim3((NP + 255) / 256), dim3(256), 0, 0, bsl, splits, (size_t)NP, gb);
    hipLaunchKernelGGL(slab_reduce, dim3(2048), dim3(256), 0, 0, ws, splits, (size_t)NP * KP, C);
  };
  launch();
  launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("HIP error %s\n", hipGetErrorString(e));
  const float us = ms * 100.f;
  std::vector<float> hc((size_t)NP * KP), hg(NP);
  hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hg.data(), gb, hg.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxb = 0;
  for (int t = 0; t < 48; ++t) {
    const int n = (t * 137 + 5) % NP, k = (t * 211 + 3) % KP;
    double ref = 0, sc = 0;
    for (int m = 0; m < M; ++m) {
      ref += (double)ha[(size_t)m * NP + n] * hb[(size_t)m * KP + k];
      sc += fabs((double)ha[(size_t)m * NP + n] * hb[(size_t)m * KP + k]);
    }
    maxerr = fmax(maxerr, fabs(ref - hc[(size_t)n * KP + k]) / sc);
    double br = 0, bs = 0;
    for (int m = 0; m < M; ++m) { br += ha[(size_t)m * NP + n]; bs += fabs(ha[(size_t)m * NP + n]); }
    maxb = fmax(maxb, fabs(br - hg[n]) / bs);
  }
  const double fl = 2.0 * M * NP * KP;
  printf("M=%d NP=%d KP=%d tiles=%d splits=%d: %9.1f us  %7.1f TF(alg)  rel err %.2e  bias rel err %.2e\n", M, NP, KP, tiles, splits, us,
         fl / us / 1e6, maxerr, maxb);
  return 0;
}
