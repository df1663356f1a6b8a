// fp32-accurate GEMMs on the bf16 matrix cores ("split-bf16"): the dense projections of the scoring model
// (allrank/models/model.py:35-44 FCModel, transformer.py:193-203 q/k/v/out projections, :221-227 feed-forward).
//
// Why: gfx950 has no TF32/xf32 path and its exact-fp32 MFMA runs at the fp32 vector rate (157 TF), 1/16 of the bf16
// MFMA rate (2.5 PF).  Every fp32 operand is therefore split ON THE FLY, while it is staged into LDS, into two bf16
// terms x = hi + lo (hi = bf16(x), lo = bf16(x - hi); |x - hi - lo| <= 2^-18 |x|) and the product is evaluated as
//        A B^T  ~=  Ahi Bhi^T + Ahi Blo^T + Alo Bhi^T            (three v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// bf16 x bf16 products are exact in fp32; the dropped terms (lo*lo and the split residuals) are <= 3 * 2^-18 relative
// per product with pseudo-random sign, i.e. the same order as the round-off an fp32 accumulation over K = 512 carries
// anyway.  3 MFMAs at the bf16 rate = 833 TF "fp32-equivalent" ceiling, 5.3x the fp32-MFMA roof, and each pair of
// fragment loads feeds 3 MFMAs, so the LDS traffic per MFMA is a third of a plain bf16 GEMM's.
// (NTERMS = 3 adds lo2 and uses 6 MFMAs for a 2^-26 product error: the strict mode.)
//
//   ltrx_gemm_nt : C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]) (+ ReLU)        both operands K-contiguous
//                  -> forward of nn.Linear (B = weight [out,in]) and its input gradient (B = weight^T, kept transposed)
//   ltrx_gemm_tn : C[N',K'] = A[M,N']^T * B[M,K']  (split-K over M, deterministic two-stage reduction)
//                  -> weight gradient dW = dY^T X; both operands are contraction-STRIDED in memory, they are transposed
//                     on the way into LDS (each lane converts a 4(m) x 1(n) register column into one 8-byte LDS write).
//
// Tiling: workgroup = 4 waves (2 x 2), tile 128 x 128 x 32; wave tile 64 x 64 = 2 x 2 MFMA tiles (64 accumulator
// VGPRs).  LDS operand images are [row][k] bf16 with k contiguous and a row stride of 40 elements (80 B): every lane
// fetches its 8-element MFMA fragment with one ds_read_b128, conflict-free.  Global loads of K-tile t+1 are issued
// into registers before the MFMAs of tile t (register double buffering); two workgroups fit per CU (40 KB LDS each).
// Workgroup ids are remapped so that the column tiles of one A row-panel run on the same XCD (shared L2).
#include "ltrx_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BN = 128;   // output-tile columns (rows of B); the row count BM_ and the K-tile depth BK_ are template parameters

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// split 4 floats into hi / lo (/ lo2) bf16 quads
template <int NTERMS>
__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo, bf16x4& lo2) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)x[e];
    const float r1 = x[e] - (float)h;
    const __bf16 l = (__bf16)r1;
    hi[e] = h;
    lo[e] = l;
    if (NTERMS == 3) lo2[e] = (__bf16)(r1 - (float)l);
  }
}

// LDS operand images: [term][row][k] bf16, k contiguous, row stride BK_+8 elements (16-byte aligned rows whose
// stride in dwords is 4 mod 16 -> the 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots).
template <int NTERMS, int BM_, int BK_>
struct Smem {
  static constexpr int LDK = BK_ + 8;
  __bf16 a[NTERMS][BM_ * LDK];
  __bf16 b[NTERMS][BN * LDK];
};

// the MFMA phase over one staged K-tile: wave (wr, wc) owns a 64 x 64 sub-tile = 2 x 2 MFMA tiles
template <int NTERMS, int BM_, int BK_>
__device__ __forceinline__ void mma_tile(const Smem<NTERMS, BM_, BK_>& s, int wr, int wc, f32x16 (&acc)[2][2]) {
  constexpr int LDK = Smem<NTERMS, BM_, BK_>::LDK;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int ks = 0; ks < BK_ / 16; ++ks) {
    bf16x8 af[NTERMS][2], bfr[NTERMS][2];
#pragma unroll
    for (int t = 0; t < NTERMS; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[t][i] = *reinterpret_cast<const bf16x8*>(&s.a[t][(wr * 64 + i * 32 + l31) * LDK + ks * 16 + 8 * half]);
        bfr[t][i] = *reinterpret_cast<const bf16x8*>(&s.b[t][(wc * 64 + i * 32 + l31) * LDK + ks * 16 + 8 * half]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // smallest terms first
        if (NTERMS == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bfr[0][j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
      }
  }
}

// XCD-aware bijective remap of a linear workgroup id (cdna_hip_programming.md §5): ids that are consecutive after
// the remap land on the same XCD, so tiles sharing an operand panel share an L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// NT:  C[M,N] = A[M,K] B[N,K]^T (+bias) (+ReLU | * mask)
// workgroup = 2 * BM_/64 waves: (BM_/64) x 2 grid of 64 x 64 wave tiles over a BM_ x 128 output tile
// ------------------------------------------------------------------------------------------------------------------
template <int NTERMS, int BM_, int BK_>
__global__ void __launch_bounds__(BM_ * 2, 2) ltrx_gemm_nt_kernel(const float* __restrict__ A, int lda,
                                                                  const float* __restrict__ B, int ldb,
                                                                  float* __restrict__ C, int ldc, int M, int N, int K,
                                                                  const float* __restrict__ bias, int act,
                                                                  const float* __restrict__ aux, int ldaux, int tiles_n) {
  constexpr int T = BM_ * 2;                    // threads
  constexpr int C4 = BK_ / 4;                   // float4 per tile row
  constexpr int LDK = Smem<NTERMS, BM_, BK_>::LDK;
  constexpr int PA = BM_ * C4 / T;              // float4 per thread for the A tile
  constexpr int PB = BN * C4 / T;               //                     ... B tile
  __shared__ __attribute__((aligned(16))) Smem<NTERMS, BM_, BK_> s;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (id / tiles_n) * BM_, n0 = (id % tiles_n) * BN;
  const int wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int srow = threadIdx.x / C4;            // staging: thread -> (row, float4 column); passes step T / C4 rows
  const int sc4 = (threadIdx.x % C4) * 4;
  constexpr int RSTEP = T / C4;
  float4 ra[PA], rb[PB];

  auto gload = [&](int k0) {
    const int k = k0 + sc4;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int r = srow + RSTEP * p;
      ra[p] = (m0 + r < M && k < K) ? *reinterpret_cast<const float4*>(A + (size_t)(m0 + r) * lda + k)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int r = srow + RSTEP * p;
      rb[p] = (n0 + r < N && k < K) ? *reinterpret_cast<const float4*>(B + (size_t)(n0 + r) * ldb + k)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&]() {
    bf16x4 h, l, l2;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int r = srow + RSTEP * p;
      split4<NTERMS>(ra[p], h, l, l2);
      *reinterpret_cast<bf16x4*>(&s.a[0][r * LDK + sc4]) = h;
      *reinterpret_cast<bf16x4*>(&s.a[1][r * LDK + sc4]) = l;
      if (NTERMS == 3) *reinterpret_cast<bf16x4*>(&s.a[2][r * LDK + sc4]) = l2;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int r = srow + RSTEP * p;
      split4<NTERMS>(rb[p], h, l, l2);
      *reinterpret_cast<bf16x4*>(&s.b[0][r * LDK + sc4]) = h;
      *reinterpret_cast<bf16x4*>(&s.b[1][r * LDK + sc4]) = l;
      if (NTERMS == 3) *reinterpret_cast<bf16x4*>(&s.b[2][r * LDK + sc4]) = l2;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + BK_ - 1) / BK_;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                 // the previous tile's fragments are consumed
    sstore();
    __syncthreads();
    if (kt + 1 < nk) gload((kt + 1) * BK_);   // in flight during the MFMAs below
    mma_tile<NTERMS, BM_, BK_>(s, wr, wc, acc);
  }

  // epilogue: lane owns column n0 + wc*64 + j*32 + (lane&31); register r is row rowmap(r, half)
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + l31;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + rowmap(r, half);
        if (row < M) {
          float v = acc[i][j][r] + bv;
          if (act == 1) v = fmaxf(v, 0.f);
          if (act == 2) v = (aux[(size_t)row * ldaux + col] > 0.f) ? v : 0.f;    // ReLU backward fused into the dgrad
          C[(size_t)row * ldc + col] = v;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// TN (weight gradient):  C[N',K'] = sum_m A[m][n'] * B[m][k'],  split over m into `splits` slabs
// ------------------------------------------------------------------------------------------------------------------
template <int NTERMS>
__global__ void __launch_bounds__(256, 2) ltrx_gemm_tn_kernel(const float* __restrict__ A, int lda,
                                                              const float* __restrict__ B, int ldb,
                                                              float* __restrict__ slabs, float* __restrict__ bias_slabs,
                                                              int M, int NP, int KP, int tiles_k, int m_per_split) {
  constexpr int BM = 128, BK = 32;
  constexpr int LDK = Smem<NTERMS, BM, BK>::LDK;
  __shared__ __attribute__((aligned(16))) Smem<NTERMS, BM, BK> s;
  const int tile = blockIdx.x, split = blockIdx.y;
  const int n0 = (tile / tiles_k) * BM, k0 = (tile % tiles_k) * BN;     // output tile: rows n', cols k'
  const int wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int lane = threadIdx.x & 63;
  const int mbeg = split * m_per_split, mend = min(M, mbeg + m_per_split);

  // staging map: the K-tile is 32 contraction rows (m) x 128 columns.  A wave covers 8 m-rows (two groups of 4) per
  // pass... each lane owns ONE column (c = lane + 64*cc) and 4 consecutive m rows -> after the split one 8-byte LDS
  // write lands at [c][m..m+3] (k contiguous).  4 waves x 2 column halves x 4 m-groups: thread -> (mg, cc).
  //   thread t: column c = (t & 127), m-group g = t >> 7 (0..1); passes p = 0..3 cover m = 8p + 4g .. +3
  const int scol = threadIdx.x & 127, sg = threadIdx.x >> 7;
  float ra[4][4], rb[4][4];     // [pass][m within group]
  const bool want_bias = (bias_slabs != nullptr) && (tile % tiles_k == 0);   // column sums of A = the bias gradient
  float bsum = 0.f;

  auto gload = [&](int mt) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = mt + 8 * p + 4 * sg + e;
        const bool okm = m < mend;
        ra[p][e] = (okm && n0 + scol < NP) ? A[(size_t)m * lda + n0 + scol] : 0.f;
        rb[p][e] = (okm && k0 + scol < KP) ? B[(size_t)m * ldb + k0 + scol] : 0.f;
      }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int kk = 8 * p + 4 * sg;
      bf16x4 h, l, l2;
      split4<NTERMS>(make_float4(ra[p][0], ra[p][1], ra[p][2], ra[p][3]), h, l, l2);
      *reinterpret_cast<bf16x4*>(&s.a[0][scol * LDK + kk]) = h;
      *reinterpret_cast<bf16x4*>(&s.a[1][scol * LDK + kk]) = l;
      if (NTERMS == 3) *reinterpret_cast<bf16x4*>(&s.a[2][scol * LDK + kk]) = l2;
      split4<NTERMS>(make_float4(rb[p][0], rb[p][1], rb[p][2], rb[p][3]), h, l, l2);
      *reinterpret_cast<bf16x4*>(&s.b[0][scol * LDK + kk]) = h;
      *reinterpret_cast<bf16x4*>(&s.b[1][scol * LDK + kk]) = l;
      if (NTERMS == 3) *reinterpret_cast<bf16x4*>(&s.b[2][scol * LDK + kk]) = l2;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (mbeg < mend) {
    gload(mbeg);
    for (int mt = mbeg; mt < mend; mt += BK) {
      __syncthreads();
      sstore();
      if (want_bias) {
#pragma unroll
        for (int p = 0; p < 4; ++p) bsum += (ra[p][0] + ra[p][1]) + (ra[p][2] + ra[p][3]);
      }
      __syncthreads();
      if (mt + BK < mend) gload(mt + BK);
      mma_tile<NTERMS, BM, BK>(s, wr, wc, acc);
    }
  }
  if (want_bias && n0 + scol < NP) bias_slabs[((size_t)split * 2 + sg) * NP + n0 + scol] = bsum;
  float* slab = slabs + (size_t)split * NP * KP;
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = k0 + wc * 64 + j * 32 + l31;
    if (col >= KP) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n0 + wr * 64 + i * 32 + rowmap(r, half);
        if (row < NP) slab[(size_t)row * KP + col] = acc[i][j][r];
      }
  }
}

// C[i] = sum_s slabs[s][i]   (fixed order; 16-byte accesses when n % 4 == 0, scalar otherwise)
__global__ void __launch_bounds__(256) ltrx_gemm_slab_reduce_kernel(const float* __restrict__ slabs, int splits,
                                                                    size_t n, float* __restrict__ C) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = 0.f;
    for (int sidx = 0; sidx < splits; ++sidx) a += slabs[(size_t)sidx * n + i];
    C[i] = a;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
// tile variant: 0 = auto; 1 = 128x128x32 (4 waves); 2 = 128x128x64; 3 = 256x128x32 (8 waves); 4 = 256x128x64
static int g_nt_variant = 0;
extern "C" void ltrx_gemm_set_variant(int v) { g_nt_variant = v; }

template <int NTERMS, int BM_, int BK_>
static void launch_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                      const float* bias, int act, const float* aux, int ldaux, hipStream_t s) {
  const int tiles_m = (M + BM_ - 1) / BM_, tiles_n = (N + BN - 1) / BN;
  hipLaunchKernelGGL((ltrx_gemm_nt_kernel<NTERMS, BM_, BK_>), dim3(tiles_m * tiles_n), dim3(BM_ * 2), 0, s, A, lda, B, ldb, C,
                     ldc, M, N, K, bias, act, aux, ldaux, tiles_n);
}

extern "C" int ltrx_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                            const float* bias, int act, const float* aux, int ldaux, int strict, ltrx_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return LTRX_EINVAL;
  if (act == 2 && (!aux || ldaux < N)) return LTRX_EINVAL;
  if ((K & 3) || (lda & 3) || (ldb & 3) || lda < K || ldb < K || ldc < N) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int v = g_nt_variant;
  if (v == 0) v = 1;
  if (strict) {
    launch_nt<3, 128, 32>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, s);
  } else {
    switch (v) {
      case 2: launch_nt<2, 128, 64>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, s); break;
      case 3: launch_nt<2, 256, 32>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, s); break;
      case 4: launch_nt<2, 256, 64>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, s); break;
      default: launch_nt<2, 128, 32>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, s); break;
    }
  }
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

static int tn_splits(int M, int tiles) {
  int want = (512 + tiles - 1) / tiles;            // aim for >= 512 workgroups
  int maxs = (M + 4 * 32 - 1) / (4 * 32);          // at least 4 K-tiles per split
  if (want > maxs) want = maxs;
  if (want > 64) want = 64;
  return want < 1 ? 1 : want;
}

extern "C" size_t ltrx_gemm_tn_workspace_bytes(int M, int NP, int KP) {
  if (M <= 0 || NP <= 0 || KP <= 0) return 0;
  const int tiles = ((NP + 127) / 128) * ((KP + BN - 1) / BN);
  const size_t sp = (size_t)tn_splits(M, tiles);
  return (sp * NP * KP + 2 * sp * NP) * sizeof(float);
}

extern "C" int ltrx_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, float* bias_out, int M, int NP,
                            int KP, int strict, void* ws, ltrx_stream_t stream) {
  if (!A || !B || !C || !ws || M <= 0 || NP <= 0 || KP <= 0) return LTRX_EINVAL;
  if (lda < NP || ldb < KP) return LTRX_EUNSUPPORTED;
  const int tiles_n = (NP + 127) / 128, tiles_k = (KP + BN - 1) / BN;
  const int tiles = tiles_n * tiles_k;
  const int splits = tn_splits(M, tiles);
  int mps = (M + splits - 1) / splits;
  mps = (mps + 31) / 32 * 32;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(tiles, splits);
  float* bslabs = bias_out ? (float*)ws + (size_t)splits * NP * KP : nullptr;
  if (strict)
    hipLaunchKernelGGL(ltrx_gemm_tn_kernel<3>, grid, dim3(256), 0, s, A, lda, B, ldb, (float*)ws, bslabs, M, NP, KP, tiles_k, mps);
  else
    hipLaunchKernelGGL(ltrx_gemm_tn_kernel<2>, grid, dim3(256), 0, s, A, lda, B, ldb, (float*)ws, bslabs, M, NP, KP, tiles_k, mps);
  LTRX_LAUNCH_CHECK();
  if (bias_out) {
    hipLaunchKernelGGL(ltrx_gemm_slab_reduce_kernel, dim3((NP + 255) / 256), dim3(256), 0, s, (const float*)bslabs, 2 * splits,
                       (size_t)NP, bias_out);
    LTRX_LAUNCH_CHECK();
  }
  const size_t n = (size_t)NP * KP;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ltrx_gemm_slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)ws, splits, n, C);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
