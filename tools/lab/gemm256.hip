// Lab: 256 x 256 x 32 split-bf16 NT GEMM, 8 waves (2 x 4), wave tile 128 x 64, LDS double buffer, one barrier per K-step.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/lab/gemm256 tools/lab/gemm256.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 32;

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

__device__ unsigned long long g_stamps[8][64][5];
__device__ __forceinline__ unsigned long long stamp() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
template <int VAR>
__global__ void __launch_bounds__(512) gemm256(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                               float* __restrict__ C, int ldc, int M, int N, int K,
                                               const float* __restrict__ bias, int tiles_n,
                                               const __bf16* __restrict__ Bh = nullptr, const __bf16* __restrict__ Bl = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem* s = reinterpret_cast<Smem*>(smem_raw);
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * BN;
  const int wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int srow = threadIdx.x >> 3, sc4 = (threadIdx.x & 7) * 4;
  float4 ra[4], rb[4], ra2[4], rb2[4];
  const float* Ap = A + (size_t)(((VAR & 256) ? 0 : m0) + srow) * lda + sc4;
  const float* Bp = B + (size_t)(((VAR & 256) ? 0 : n0) + srow) * ldb + sc4;
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (VAR & 2) {
        ra[p] = make_float4(1.f + k0, 2.f, 3.f, 4.f + p);
        rb[p] = make_float4(1.f, 2.f + p, 3.f + k0, 4.f);
      } else {
        ra[p] = *reinterpret_cast<const float4*>(Ap + (size_t)(64 * p) * lda + k0);
        rb[p] = *reinterpret_cast<const float4*>(Bp + (size_t)(64 * p) * ldb + k0);
      }
    }
  };
  auto sstore = [&](Smem& d) {
    if (VAR & 4) return;
    bf16x4 h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int o = swz_off(srow + 64 * p, sc4);
      split4(ra[p], h, l);
      *reinterpret_cast<bf16x4*>(&d.a[0][o]) = h;
      *reinterpret_cast<bf16x4*>(&d.a[1][o]) = l;
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
      if (VAR & 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j][0] += (float)af[0][i][0] * (float)bfr[1][j][1] + (float)af[1][i][2] * (float)bfr[0][j][3];
      } else {
#define MMA(TA, TB)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =       \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA][i], bfr[TB][j], acc[i][j], 0, 0, 0);
        MMA(0, 1) MMA(1, 0) MMA(0, 0)
#undef MMA
      }
    }
  };

  const int nk = K / BK;
  if (VAR & 16384) {                       // static MFMA priority for the first wave of every SIMD
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) == 0) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
  }
  gload(0);
  sstore(s[0]);
  if (nk > 1) gload(BK);
  __syncthreads();
  if (VAR & 1024) {
    auto lds_barrier = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    auto gload2 = [&](float4 (&xa)[4], float4 (&xb)[4], int k0) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        xa[p] = *reinterpret_cast<const float4*>(Ap + (size_t)(64 * p) * lda + k0);
        xb[p] = *reinterpret_cast<const float4*>(Bp + (size_t)(64 * p) * ldb + k0);
      }
    };
    auto sstore2 = [&](Smem& d, const float4 (&xa)[4], const float4 (&xb)[4]) {
      bf16x4 h, l;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int o = swz_off(srow + 64 * p, sc4);
        split4(xa[p], h, l);
        *reinterpret_cast<bf16x4*>(&d.a[0][o]) = h;
        *reinterpret_cast<bf16x4*>(&d.a[1][o]) = l;
        split4(xb[p], h, l);
        *reinterpret_cast<bf16x4*>(&d.b[0][o]) = h;
        *reinterpret_cast<bf16x4*>(&d.b[1][o]) = l;
      }
    };
    // on entry: LDS s[0] = tile 0, set (ra, rb) = tile 1 in flight.  Invariant at the top of step kt: set[kt&1] is free,
    // set[(kt+1)&1] holds tile kt+1.
    int kt = 0;
    for (; kt + 3 < nk; kt += 2) {
      gload2(ra2, rb2, (kt + 2) * BK);               // free set <- tile kt+2 (needed two phases from now)
      mma(s[0]);
      sstore2(s[1], ra, rb);                         // tile kt+1
      lds_barrier();
      gload2(ra, rb, (kt + 3) * BK);
      mma(s[1]);
      sstore2(s[0], ra2, rb2);                       // tile kt+2
      lds_barrier();
    }
    // tail: nk - kt in {1, 2, 3} tiles left; (ra, rb) holds tile kt+1 if it exists
    for (; kt < nk; ++kt) {
      mma(s[kt & 1]);
      if (kt + 1 < nk) {
        sstore(s[(kt + 1) & 1]);                     // uses (ra, rb)
        if (kt + 2 < nk) gload((kt + 2) * BK);
      }
      __syncthreads();
    }
  } else if (VAR & 32) {
    auto lds_barrier = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    const bool late = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;     // waves 4..7: store first
    const int stag_pos = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // 0..7
    int kt = 0;
    if (VAR & 65536) {
      // B pre-split: per K-step a thread fetches 2 x 16 B of the hi image and 2 x 16 B of the lo image (row = t >> 2 (+128),
      // chunk = t & 3) and stores them with ds_write_b128; A is staged as before (4 float4 -> split -> 8 ds_write_b64)
      const int brow = threadIdx.x >> 2, bch = threadIdx.x & 3;
      typedef __bf16 bx8 __attribute__((ext_vector_type(8)));
      bx8 bh[2], bl[2];
      auto gloadB = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          bh[p] = *reinterpret_cast<const bx8*>(Bh + (size_t)(n0 + brow + 128 * p) * ldb + k0 + 8 * bch);
          bl[p] = *reinterpret_cast<const bx8*>(Bl + (size_t)(n0 + brow + 128 * p) * ldb + k0 + 8 * bch);
        }
      };
      auto gloadA = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const float4*>(Ap + (size_t)(64 * p) * lda + k0);
      };
      auto sstoreAB = [&](Smem& d) {
        bf16x4 h, l;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int o = swz_off(srow + 64 * p, sc4);
          split4(ra[p], h, l);
          *reinterpret_cast<bf16x4*>(&d.a[0][o]) = h;
          *reinterpret_cast<bf16x4*>(&d.a[1][o]) = l;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int o = swz_off(brow + 128 * p, 8 * bch);
          *reinterpret_cast<bx8*>(&d.b[0][o]) = bh[p];
          *reinterpret_cast<bx8*>(&d.b[1][o]) = bl[p];
        }
      };
      // (the generic prologue above staged tile 0 from the fp32 B; redo it from the pre-split images for a clean measurement)
      gloadA(0); gloadB(0);
      __syncthreads();
      sstoreAB(s[0]);
      if (nk > 1) { gloadA(BK); gloadB(BK); }
      __syncthreads();
      for (; kt + 2 < nk; ++kt) {
        mma(s[kt & 1]);
        sstoreAB(s[(kt + 1) & 1]);
        gloadA((kt + 2) * BK); gloadB((kt + 2) * BK);
        lds_barrier();
      }
      for (; kt < nk; ++kt) {
        mma(s[kt & 1]);
        if (kt + 1 < nk) sstoreAB(s[(kt + 1) & 1]);
        __syncthreads();
      }
    }
    if (VAR & 4096) {
      const bool rec = (blockIdx.x == 1000) && lane == 0;
      for (; kt + 2 < nk; ++kt) {
        Smem& cur = s[kt & 1];
        Smem& nxt = s[(kt + 1) & 1];
        const unsigned long long t0 = stamp();
        mma(cur);
        const unsigned long long t1 = stamp();
        sstore(nxt);
        const unsigned long long t2 = stamp();
        gload((kt + 2) * BK);
        const unsigned long long t3 = stamp();
        lds_barrier();
        const unsigned long long t4 = stamp();
        if (rec && kt < 64) {
          g_stamps[wave][kt][0] = t0; g_stamps[wave][kt][1] = t1; g_stamps[wave][kt][2] = t2; g_stamps[wave][kt][3] = t3; g_stamps[wave][kt][4] = t4;
        }
      }
    }
    for (; kt + 2 < nk; ++kt) {            // steady state: no branches in the body
      Smem& cur = s[kt & 1];
      Smem& nxt = s[(kt + 1) & 1];
      if (VAR & 32768) {
        // source-level interleave: 8 x { 6 MFMAs ; fence ; split + 2x2 ds_write of one staged float4 pair ; fence }
        bf16x8 af[2][4], bfr[2][2];
        auto frags = [&](int ks) {
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
              bfr[tt][j] = *reinterpret_cast<const bf16x8*>(&cur.b[tt][swz_off(wc * 64 + j * 32 + l31, ks * 16 + 8 * half)]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              af[tt][i] = *reinterpret_cast<const bf16x8*>(&cur.a[tt][swz_off(wr * 128 + i * 32 + l31, ks * 16 + 8 * half)]);
          }
        };
        auto store_one = [&](int p) {
          bf16x4 h, l;
          const int o = swz_off(srow + 64 * p, sc4);
          split4(ra[p], h, l);
          *reinterpret_cast<bf16x4*>(&nxt.a[0][o]) = h;
          *reinterpret_cast<bf16x4*>(&nxt.a[1][o]) = l;
          split4(rb[p], h, l);
          *reinterpret_cast<bf16x4*>(&nxt.b[0][o]) = h;
          *reinterpret_cast<bf16x4*>(&nxt.b[1][o]) = l;
          ra[p] = *reinterpret_cast<const float4*>(Ap + (size_t)(64 * p) * lda + (kt + 2) * BK);
          rb[p] = *reinterpret_cast<const float4*>(Bp + (size_t)(64 * p) * ldb + (kt + 2) * BK);
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          frags(ks);
#pragma unroll
          for (int c = 0; c < 4; ++c) {              // 4 chunks of 6 MFMAs: linear index m = 6c .. 6c+5 over (product, i, j)
#pragma unroll
            for (int m = 6 * c; m < 6 * c + 6; ++m) {
              const int prod = m / 8, ij = m % 8, i = ij >> 1, j = ij & 1;
              const int ta = prod == 1 ? 1 : 0, tb = prod == 0 ? 1 : 0;     // (0,1) (1,0) (0,0)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ta][i], bfr[tb][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (VAR & 131072) {
              // every wave stages its share (all 4 float4 pairs) after a DIFFERENT chunk: the 8 waves' load groups are spread
              // over the K-step instead of arriving at the vector-memory path together
              if (stag_pos == 4 * ks + c) {
                store_one(0);
                store_one(1);
                store_one(2);
                store_one(3);
              }
            } else {
              if (c & 1) store_one(2 * ks + (c >> 1));  // 4 staged (A,B) float4 pairs per K-step: after chunks 1,3 of each ks
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if (VAR & 8192) {
        // both wave groups run the same two blocks; `late` only selects which one comes first (2-trip loop, not unrolled)
#pragma unroll 1
        for (int ph = 0; ph < 2; ++ph) {
          if ((ph != 0) != late) {
            sstore(nxt);
            gload((kt + 2) * BK);
          } else {
            mma(cur);
          }
        }
      } else if (VAR & 512) {
        if (late) {
          sstore(nxt);
          gload((kt + 2) * BK);
          __builtin_amdgcn_sched_barrier(0);
          mma(cur);
        } else {
          mma(cur);
          __builtin_amdgcn_sched_barrier(0);
          sstore(nxt);
          gload((kt + 2) * BK);
        }
      } else if ((VAR & 128) ? (wave >= 4) : (VAR & 1)) {
        sstore(nxt);
        gload((kt + 2) * BK);
        mma(cur);
      } else {
        mma(cur);
        sstore(nxt);
        gload((kt + 2) * BK);
      }
      if (VAR & 64) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
          for (int m = 0; m < 6; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
          }
          __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);     // 2 DS write
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);     // 3 DS read
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM read
        }
      }
      lds_barrier();
    }
    for (; kt < nk; ++kt) {
      Smem& cur = s[kt & 1];
      Smem& nxt = s[(kt + 1) & 1];
      mma(cur);
      if (kt + 1 < nk) sstore(nxt);
      __syncthreads();
    }
  } else {
  for (int kt = 0; kt < nk; ++kt) {
    Smem& cur = s[kt & 1];
    Smem& nxt = s[(kt + 1) & 1];
    if (!(VAR & 1)) {
      mma(cur);
      if (kt + 1 < nk) sstore(nxt);
      if (kt + 2 < nk) gload((kt + 2) * BK);
    } else {
      if (kt + 1 < nk) sstore(nxt);
      if (kt + 2 < nk) gload((kt + 2) * BK);
      mma(cur);
    }
    __syncthreads();
  }
  }
  if (VAR & 8) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
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
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 128 + i * 32 + rowmap(r, half);
        if (VAR & 2048) __builtin_nontemporal_store(fmaxf(acc[i][j][r] + bv, 0.f), &C[(size_t)row * ldc + col]);
        else C[(size_t)row * ldc + col] = fmaxf(acc[i][j][r] + bv, 0.f);
      }
  }
}

static const __bf16* g_Bh = nullptr;
static const __bf16* g_Bl = nullptr;
template <int VAR>
static float run(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int iters) {
  const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
  hipFuncSetAttribute((const void*)gemm256<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sizeof(Smem));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i)
    hipLaunchKernelGGL((gemm256<VAR>), dim3(tiles), dim3(512), 2 * sizeof(Smem), 0, A, K, B, K, C, N, M, N, K, bias, tiles_n, g_Bh, g_Bl);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((gemm256<VAR>), dim3(tiles), dim3(512), 2 * sizeof(Smem), 0, A, K, B, K, C, N, M, N, K, bias, tiles_n, g_Bh, g_Bl);
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
  {
    std::vector<__bf16> hh((size_t)N * K), hl((size_t)N * K);
    for (size_t i = 0; i < hh.size(); ++i) {
      const __bf16 h = (__bf16)hb[i];
      hh[i] = h;
      hl[i] = (__bf16)(hb[i] - (float)h);
    }
    __bf16 *dh, *dl;
    hipMalloc(&dh, hh.size() * 2);
    hipMalloc(&dl, hl.size() * 2);
    hipMemcpy(dh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dl, hl.data(), hl.size() * 2, hipMemcpyHostToDevice);
    g_Bh = dh;
    g_Bl = dl;
  }
  const double fl = 2.0 * M * N * K;
  printf("M=%d N=%d K=%d\n", M, N, K);
  const int only = argc > 4 ? atoi(argv[4]) : -1;
#define RUN(name, V)                                                                                         \
  if (only < 0 || only == V) {                                                                               \
    float us = run<V>(A, B, C, bias, M, N, K, 10);                                                           \
    printf("%-32s %9.1f us  %7.1f TF(alg)  %7.1f TF(exec x3)\n", name, us, fl / us / 1e6, 3 * fl / us / 1e6); \
  }
  RUN("full (mma,store,load)", 0)
  RUN("full (store,load,mma)", 1)
  RUN("peeled+raw barrier (mma,st,ld)", 32)
  RUN("peeled+raw barrier (st,ld,mma)", 33)
  RUN("peeled+raw+sched (mma,st,ld)", 96)
  RUN("peeled+raw+sched (st,ld,mma)", 97)
  RUN("ping-pong by wave>>2", 160)
  RUN("staggered staging, nt stores", 32 + 2048 + 32768 + 131072)
  RUN("staggered staging, no epilogue", 32 + 8 + 32768 + 131072)
  RUN("B pre-split, nt stores", 32 + 2048 + 65536)
  RUN("B pre-split, no epilogue", 32 + 8 + 65536)
  RUN("source interleave, nt stores", 32 + 2048 + 32768)
  RUN("source interleave, no epilogue", 32 + 8 + 32768)
  RUN("static prio, nt stores", 32 + 2048 + 16384)
  RUN("static prio, no epilogue", 32 + 8 + 16384)
  RUN("ping-pong 2-trip loop", 32 + 8192)
  RUN("ping-pong 2-trip loop, nt stores", 32 + 8192 + 2048)
  RUN("ping-pong 2-trip, no epilogue", 32 + 8192 + 8)
  RUN("ping-pong scalar branch", 32 + 512)
  RUN("ping-pong scalar, nt stores", 32 + 512 + 2048)
  RUN("peeled+raw, nontemporal C stores", 32 + 2048)
  RUN("ping-pong, nontemporal C stores", 160 + 2048)
  RUN("ping-pong, loads aliased to tile 0", 160 + 256)
  RUN("ping-pong, aliased, no epilogue", 160 + 256 + 8)
  RUN("ping-pong, no epilogue", 160 + 8)
  RUN("ping-pong, no gload", 160 + 2)
  RUN("ping-pong, no gload no epilogue", 160 + 2 + 8)
  RUN("no gload", 2)
  RUN("no sstore", 4)
  RUN("no epilogue", 8)
  RUN("no mfma", 16)
  RUN("no gload, no epilogue", 10)
  RUN("mfma+dsread only", 14)
  if (only == 4128) {
    run<4128>(A, B, C, bias, M, N, K, 1);
    static unsigned long long h[8][64][5];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h));
    for (int w = 0; w < 8; w += 3) {
      printf("wave %d: per K-step [mma | store | load-issue | barrier] cycles (readcyclecounter ticks)\n", w);
      for (int kt = 2; kt < 12; ++kt)
        printf("  kt %2d: %6llu %6llu %6llu %6llu   total %6llu\n", kt, h[w][kt][1] - h[w][kt][0], h[w][kt][2] - h[w][kt][1],
               h[w][kt][3] - h[w][kt][2], h[w][kt][4] - h[w][kt][3], h[w][kt + 1][0] - h[w][kt][0]);
    }
  }
  for (int pass = 0; pass < 2 && only < 0; ++pass) {
    hipMemset(C, 0, (size_t)M * N * 4);
    if (pass == 0) run<165920>(A, B, C, bias, M, N, K, 1); else run<33>(A, B, C, bias, M, N, K, 1);
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
    printf("check pass %d maxrelerr %.2e\n", pass, maxerr);
  }
  return 0;
}
