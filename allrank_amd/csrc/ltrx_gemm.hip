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
// Two kernel families (the host wrappers pick per shape):
//   * 128 x 128 x 32 tiles, 4 waves (2 x 2) of 64 x 64 wave tiles, one LDS stage (32 KB), two workgroups per CU -- every
//     shape, ragged edges, the strict 3-term mode; global loads of K-tile t+1 are issued into registers before the MFMAs of
//     tile t (register double buffering).
//   * 256 x 256 x 32 tiles, 8 waves (2 x 4) of 128 x 64 wave tiles, two LDS stages (128 KB, one workgroup per CU), LDS-only
//     barrier, nontemporal output -- the big projections and their weight gradients (ltrx_gemm_nt256_kernel,
//     ltrx_gemm_tn256_kernel below; 1.3-1.6x the small tile on those shapes, see DESIGN.md section 4 for the measurements).
// LDS operand images are [row][k] bf16 with k contiguous, UNPADDED 64-byte rows and an XOR swizzle of the 16-byte chunks
// (swz_off): every lane fetches its 8-element MFMA fragment with one ds_read_b128, conflict-free in both directions.
// Workgroup ids are remapped so that the column tiles of one A row-panel run on the same XCD (shared L2).
#include "ltrx_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
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

// LDS operand images: [term][row][k] bf16, k contiguous, UNPADDED rows of BK_ elements (64 B at BK_ = 32); the 16-byte
// chunk c of row r is stored at chunk position c ^ swz(r), swz(r) = (r >> 2) & (BK_/8 - 1).  With that swizzle the
// 16-lane groups of a fragment ds_read_b128 (rows {0-3,12-15,20-27}+..., same logical chunk) hit 16 distinct 4-bank
// slots, and the staging ds_write_b64 (8 lanes per row, 2 rows per 16-lane group, rows 64 B apart) covers all 32 write
// banks exactly once: both directions are conflict-free (SQ_LDS_BANK_CONFLICT was 33 % of LDS cycles with padded rows).
template <int NTERMS, int BM_, int BK_>
struct Smem {
  static constexpr int LDK = BK_;
  __bf16 a[NTERMS][BM_ * LDK];
  __bf16 b[NTERMS][BN * LDK];
};

template <int BK_>
__device__ __forceinline__ int swz_off(int row, int k) {       // element offset of (row, k) inside an operand image
  constexpr int CH = BK_ / 8;                                   // 16-byte chunks per row
  const int c = (k >> 3) ^ ((row >> 2) & (CH - 1));
  return row * BK_ + c * 8 + (k & 7);
}

// the MFMA phase over one staged K-tile: wave (wr, wc) owns a 64 x 64 sub-tile = 2 x 2 MFMA tiles
template <int NTERMS, int BM_, int BK_>
__device__ __forceinline__ void mma_tile(const Smem<NTERMS, BM_, BK_>& s, int wr, int wc, f32x16 (&acc)[2][2]) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int ks = 0; ks < BK_ / 16; ++ks) {
    bf16x8 af[NTERMS][2], bfr[NTERMS][2];
#pragma unroll
    for (int t = 0; t < NTERMS; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[t][i] = *reinterpret_cast<const bf16x8*>(&s.a[t][swz_off<BK_>(wr * 64 + i * 32 + l31, ks * 16 + 8 * half)]);
        bfr[t][i] = *reinterpret_cast<const bf16x8*>(&s.b[t][swz_off<BK_>(wc * 64 + i * 32 + l31, ks * 16 + 8 * half)]);
      }
    // term-major order: consecutive MFMAs write DIFFERENT accumulators (no back-to-back dependent issue); the
    // smallest terms are accumulated first.
#define LTRX_MMA_TERM(TA, TB)                                                                                     \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA][i], bfr[TB][j], acc[i][j], 0, 0, 0);
    if (NTERMS == 3) {
      LTRX_MMA_TERM(NTERMS - 2, NTERMS - 2)
      LTRX_MMA_TERM(0, NTERMS - 1)
      LTRX_MMA_TERM(NTERMS - 1, 0)
    }
    if (NTERMS >= 2) {
      LTRX_MMA_TERM(0, NTERMS >= 2 ? 1 : 0)
      LTRX_MMA_TERM(NTERMS >= 2 ? 1 : 0, 0)
    }
    LTRX_MMA_TERM(0, 0)
#undef LTRX_MMA_TERM
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
__global__ void __launch_bounds__(BM_ * 2) __attribute__((amdgpu_waves_per_eu(2, 3))) ltrx_gemm_nt_kernel(const float* __restrict__ A, int lda,
                                                                  const float* __restrict__ B, int ldb,
                                                                  float* __restrict__ C, int ldc, int M, int N, int K,
                                                                  const float* __restrict__ bias, int act,
                                                                  const float* __restrict__ aux, int ldaux, int tiles_n,
                                                                  ltrx::DropSpec drop, const uint32_t* __restrict__ drop_step) {
  constexpr int T = BM_ * 2;                    // threads
  constexpr int C4 = BK_ / 4;                   // float4 per tile row
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
      const int o = swz_off<BK_>(srow + RSTEP * p, sc4);
      split4<NTERMS>(ra[p], h, l, l2);
      *reinterpret_cast<bf16x4*>(&s.a[0][o]) = h;
      if (NTERMS >= 2) *reinterpret_cast<bf16x4*>(&s.a[NTERMS >= 2 ? 1 : 0][o]) = l;
      if (NTERMS == 3) *reinterpret_cast<bf16x4*>(&s.a[NTERMS - 1][o]) = l2;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int o = swz_off<BK_>(srow + RSTEP * p, sc4);
      split4<NTERMS>(rb[p], h, l, l2);
      *reinterpret_cast<bf16x4*>(&s.b[0][o]) = h;
      if (NTERMS >= 2) *reinterpret_cast<bf16x4*>(&s.b[NTERMS >= 2 ? 1 : 0][o]) = l;
      if (NTERMS == 3) *reinterpret_cast<bf16x4*>(&s.b[NTERMS - 1][o]) = l2;
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
  ltrx::DropSpec dsp = drop;
  if (drop_step) dsp.seed ^= drop_step[0] * 0x9E3779B9u;
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
          if (act == 2) v = (aux[(size_t)row * ldaux + col] > 0.f) ? v * drop.inv_keep : 0.f;   // ReLU(+dropout) backward
          else if (drop.thresh != 0u) v *= ltrx::drop_keep_scale(dsp, (uint64_t)row * (uint64_t)N + (uint64_t)col);
          if (act == 3) v += aux[(size_t)row * ldaux + col];                                    // residual stream + drop(branch)
          C[(size_t)row * ldc + col] = v;
        }
      }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// NT, large-tile variant for the big projections: 256 x 256 x 32 tile, 8 waves (2 x 4), wave tile 128 x 64 (128
// accumulator VGPRs), two LDS stages (128 KB: one workgroup per CU) and ONE LDS-only barrier per K-step.
// Why (tools/lab/gemm_ablate.hip, gemm256.hip; MI355X, M 61440 x N 2048 x K 512): in the 128 x 128 kernel the memory side
// (global loads -> split -> LDS) alone takes 340 us and the MFMAs alone 185 us, and the two do not overlap (567 us):
//   * per MFMA the 256 x 256 tile needs half the global-load and LDS-write traffic and 3/4 of the LDS fragment reads;
//   * __syncthreads() drains vmcnt(0), i.e. it exposes the latency of the prefetch issued just before it on every K-step;
//     the barrier here orders LDS only (release/acquire fences on the "local" address space + s_barrier), so the global
//     loads of tile t+2 stay in flight across it;
//   * the steady-state loop is branch-free, which lets the scheduler interleave the split/ds_write VALU work of tile t+1
//     with the MFMAs of tile t;
//   * the output tile is streamed with nontemporal stores: C (0.5 GB for the FFN) no longer evicts the operand panels
//     that the other tiles of the same XCD are about to re-read from L2.
// Measured: 567 -> 365 us (FFN1), 560 -> 345 us (FFN2).  Used when N is a multiple of 256, K of 32, and the grid fills
// the chip (a ragged last row tile is handled by clamped loads and guarded stores); everything else (and the strict
// 3-term mode, whose LDS images do not fit twice) stays on the kernel above.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_only_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// BM = 256 (wave tile 128 x 64) or 128 (wave tile 64 x 64, 96 KB of LDS): the 128-row variant is for shapes whose 256-row
// tiling would leave half of the CUs without a tile (M 15360 x N 512: 120 tiles of 256 x 256, 240 of 128 x 256)
// NT = number of bf16 terms per operand: 2 = hi + lo (three products), 1 = hi only (ONE product: the plain-bf16
// throughput mode, precision code 2 of the host wrappers -- NOT the parity arithmetic)
template <int BM, int NT = 2>
struct SmemNT {
  __bf16 a[NT][BM * 32];
  __bf16 b[NT][256 * 32];
};

// IMG >= 1: B points at a pre-split IMAGE of the operand instead of fp32 values -- per 4 consecutive k the 16 bytes {hi0..hi3, lo0..lo3}
// (bf16) in place of the 4 floats, same addressing (ltrx_split_image; the weights and their transposes, refreshed once per optimizer
// step).  The staging of B is then a plain copy: half of the kernel's split work (VALU, and the power it draws) is gone, the bits
// that reach LDS -- and the results -- are identical.
// IMG == 2 (round 5): A is an image too -- an ACTIVATION written that way by its producer (the LayerNorm forward, the feed-forward
// GEMM's own epilogue: `cimg`), which has no fp32 reader: the loop then stages both operands with plain copies, no split at all.
// cimg: the output tile leaves as an image (each 16-byte store = 4 consecutive columns of one row, split in the epilogue).
template <bool TAIL, int BM, int NT, int IMG>
__device__ __forceinline__ void nt256_body(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                           int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                           const float* __restrict__ bias, int act,
                                           const float* __restrict__ aux, int ldaux, int tiles_n,
                                           ltrx::DropSpec drop, const uint32_t* __restrict__ drop_step, int cimg) {
  constexpr int BK_ = 32;
  constexpr bool BIMG = IMG >= 1, AIMG = IMG == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef SmemNT<BM, NT> SmemT;
  constexpr int L1 = NT - 1;           // index of the lo image (aliases hi when there is none; never touched then)
  constexpr int RI = BM / 64;          // 32-row accumulator blocks per wave (two wave rows)
  SmemT* s = reinterpret_cast<SmemT*>(smem_raw);
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * 256;
  const int wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int srow = threadIdx.x >> 3, sc4 = (threadIdx.x & 7) * 4;        // staging: 64 rows x 8 float4 per pass, 4 passes
  float4 ra[RI], rb[4];
  // rows of A beyond M (the last row tile of a batch whose size is not a multiple of 256) are clamped to row M-1: they
  // are read but their results are never stored
  // (TAIL: only instantiated for a batch whose row count is not a multiple of 256 -- the extra address registers cost the
  // exact-multiple kernel 20 % when they are always there)
  const int ar0 = TAIL ? min(m0 + srow, M - 1) : m0 + srow;
  const float* Ap = A + (size_t)ar0 * lda + sc4;
  const float* Bp = B + (size_t)(n0 + srow) * ldb + sc4;
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < RI; ++p) {
      if constexpr (TAIL)
        ra[p] = *reinterpret_cast<const float4*>(Ap + (size_t)min(64 * p, M - 1 - ar0) * lda + k0);
      else
        ra[p] = *reinterpret_cast<const float4*>(Ap + (size_t)(64 * p) * lda + k0);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const float4*>(Bp + (size_t)(64 * p) * ldb + k0);
  };
  auto sstore = [&](SmemT& d) {
    bf16x4 h, l, l2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int o = swz_off<BK_>(srow + 64 * p, sc4);
      if (p < RI) {
        if constexpr (AIMG) {
          *reinterpret_cast<float2*>(&d.a[0][o]) = make_float2(ra[p < RI ? p : 0].x, ra[p < RI ? p : 0].y);
          if (NT == 2) *reinterpret_cast<float2*>(&d.a[L1][o]) = make_float2(ra[p < RI ? p : 0].z, ra[p < RI ? p : 0].w);
        } else {
          split4<2>(ra[p < RI ? p : 0], h, l, l2);
          *reinterpret_cast<bf16x4*>(&d.a[0][o]) = h;
          if (NT == 2) *reinterpret_cast<bf16x4*>(&d.a[L1][o]) = l;
        }
      }
      if constexpr (BIMG) {
        *reinterpret_cast<float2*>(&d.b[0][o]) = make_float2(rb[p].x, rb[p].y);
        if (NT == 2) *reinterpret_cast<float2*>(&d.b[L1][o]) = make_float2(rb[p].z, rb[p].w);
      } else {
        split4<2>(rb[p], h, l, l2);
        *reinterpret_cast<bf16x4*>(&d.b[0][o]) = h;
        if (NT == 2) *reinterpret_cast<bf16x4*>(&d.b[L1][o]) = l;
      }
    }
  };
  f32x16 acc[RI][2];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mma = [&](const SmemT& t) {
#pragma unroll
    for (int ks = 0; ks < BK_ / 16; ++ks) {
      bf16x8 af[NT][RI], bfr[NT][2];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bfr[tt][j] = *reinterpret_cast<const bf16x8*>(&t.b[tt][swz_off<BK_>(wc * 64 + j * 32 + l31, ks * 16 + 8 * half)]);
#pragma unroll
        for (int i = 0; i < RI; ++i)
          af[tt][i] = *reinterpret_cast<const bf16x8*>(&t.a[tt][swz_off<BK_>(wr * (BM / 2) + i * 32 + l31, ks * 16 + 8 * half)]);
      }
#define LTRX_MMA256(TA, TB)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < RI; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA][i], bfr[TB][j], acc[i][j], 0, 0, 0);
      if (NT == 2) {
        LTRX_MMA256(0, L1)
        LTRX_MMA256(L1, 0)
      }
      LTRX_MMA256(0, 0)
#undef LTRX_MMA256
    }
  };

  const int nk = K / BK_;
  gload(0);
  sstore(s[0]);
  if (nk > 1) gload(BK_);
  __syncthreads();
  int kt = 0;
  for (; kt + 2 < nk; ++kt) {                 // steady state, branch-free: MFMAs of tile kt | stage tile kt+1 | prefetch kt+2
    mma(s[kt & 1]);
    sstore(s[(kt + 1) & 1]);
    gload((kt + 2) * BK_);
    lds_only_barrier();
  }
  for (; kt < nk; ++kt) {
    mma(s[kt & 1]);
    if (kt + 1 < nk) sstore(s[(kt + 1) & 1]);
    __syncthreads();
  }

  ltrx::DropSpec dsp = drop;
  if (drop_step) dsp.seed ^= drop_step[0] * 0x9E3779B9u;
  // Epilogue with 16-byte stores.  An accumulator block keeps one COLUMN per lane (16 rows down the registers); a 4x4
  // transpose inside each lane quad (two DPP butterflies: quad_perm [1,0,3,2] then [2,3,0,1]) turns registers 4g..4g+3 of
  // the four lanes of a quad into 4 consecutive columns of ONE row per lane, so the tile leaves the CU as 32 dwordx4 stores
  // per lane instead of 128 dword stores -- the vector-memory issue path is what bounds this kernel, and the saved
  // activation of the ReLU backward (act 2) and the bias are read as float4 the same way.
  const int q = lane & 3;
  const bool b0 = q & 1, b1 = q & 2;
  const int cq = l31 & ~3;
  // act 4 / 5: the ReLU(+dropout) mask as ONE BIT per element instead of the saved fp32 activation.  The lane that produces an
  // element in the forward launch (act 4) is the lane that needs its mask in the input-gradient launch (act 5: same M, N, tiling),
  // so the 128 bits of a lane's 32 row-quads travel as one private 16-byte word per (tile, thread): 1/32 of the bytes act 2 reads.
  unsigned int mbits[4] = {0u, 0u, 0u, 0u};
  uint4* const mask_words = reinterpret_cast<uint4*>(const_cast<float*>(aux)) + (size_t)id * 512 + threadIdx.x;
  if (act == 5) {
    const uint4 w = *mask_words;
    mbits[0] = w.x; mbits[1] = w.y; mbits[2] = w.z; mbits[3] = w.w;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + cq;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      f32x4 ax[4];
      if (act == 2 || act == 3) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = m0 + wr * (BM / 2) + i * 32 + 8 * g + 4 * half + q;
          ax[g] = (!TAIL || row < M) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(aux + (size_t)row * ldaux + col))
                                     : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float a0 = acc[i][j][4 * g + 0], a1 = acc[i][j][4 * g + 1], a2 = acc[i][j][4 * g + 2], a3 = acc[i][j][4 * g + 3];
        // stage 1: exchange across lane bit 0
        const float r_lo = LTRX_DPP_F(0.f, b0 ? a0 : a1, 0xB1, 0xF, true);
        const float r_hi = LTRX_DPP_F(0.f, b0 ? a2 : a3, 0xB1, 0xF, true);
        const float c0 = b0 ? r_lo : a0, c1 = b0 ? a1 : r_lo, c2 = b0 ? r_hi : a2, c3 = b0 ? a3 : r_hi;
        // stage 2: exchange across lane bit 1
        const float r_a = LTRX_DPP_F(0.f, b1 ? c0 : c2, 0x4E, 0xF, true);
        const float r_b = LTRX_DPP_F(0.f, b1 ? c1 : c3, 0x4E, 0xF, true);
        f32x4 v = {b1 ? r_a : c0, b1 ? r_b : c1, b1 ? c2 : r_a, b1 ? c3 : r_b};   // row q, columns col..col+3
        const int row = m0 + wr * (BM / 2) + i * 32 + 8 * g + 4 * half + q;
        if (TAIL && row >= M) continue;
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (act == 1 || act == 4) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (act == 5) {
          const unsigned int nb_ = mbits[((j * RI + i) * 4 + g) >> 3] >> ((((j * RI + i) * 4 + g) & 7) * 4);
          v.x = (nb_ & 1u) ? v.x * drop.inv_keep : 0.f;
          v.y = (nb_ & 2u) ? v.y * drop.inv_keep : 0.f;
          v.z = (nb_ & 4u) ? v.z * drop.inv_keep : 0.f;
          v.w = (nb_ & 8u) ? v.w * drop.inv_keep : 0.f;
        } else if (act == 2) {
          v.x = (ax[g].x > 0.f) ? v.x * drop.inv_keep : 0.f;
          v.y = (ax[g].y > 0.f) ? v.y * drop.inv_keep : 0.f;
          v.z = (ax[g].z > 0.f) ? v.z * drop.inv_keep : 0.f;
          v.w = (ax[g].w > 0.f) ? v.w * drop.inv_keep : 0.f;
        } else if (drop.thresh != 0u) {
          const uint64_t e = (uint64_t)row * (uint64_t)N + (uint64_t)col;
          v.x *= ltrx::drop_keep_scale(dsp, e);
          v.y *= ltrx::drop_keep_scale(dsp, e + 1);
          v.z *= ltrx::drop_keep_scale(dsp, e + 2);
          v.w *= ltrx::drop_keep_scale(dsp, e + 3);
        }
        if (act == 3) {      // SublayerConnection: x + dropout(sublayer(norm(x))) (transformer.py:98-106), the stream read here
          v.x += ax[g].x; v.y += ax[g].y; v.z += ax[g].z; v.w += ax[g].w;
        }
        if (act == 4)        // (after the dropout scaling: a dropped unit is 0 in the saved activation and passes no gradient)
          mbits[((j * RI + i) * 4 + g) >> 3] |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u))
                                                << ((((j * RI + i) * 4 + g) & 7) * 4);
        if (cimg) {          // the consumer GEMMs stage this activation as an operand image: split once, here
          const float4 im = ltrx_split_image4(make_float4(v.x, v.y, v.z, v.w));
          v = f32x4{im.x, im.y, im.z, im.w};
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col));
      }
    }
  }
  if (act == 4) *mask_words = make_uint4(mbits[0], mbits[1], mbits[2], mbits[3]);
}

template <bool TAIL, int BM, int NT, int IMG>
__global__ void __launch_bounds__(512) ltrx_gemm_nt256_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                              int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                              const float* __restrict__ bias, int act,
                                                              const float* __restrict__ aux, int ldaux, int tiles_n,
                                                              ltrx::DropSpec drop, const uint32_t* __restrict__ drop_step, int cimg) {
  nt256_body<TAIL, BM, NT, IMG>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, tiles_n, drop, drop_step, cimg);
}
// 64-row tiles, TWO workgroups per CU (2 x 80 KB of LDS, 128 registers per lane): for the shapes whose 128-row tiling is a single,
// not even full round of workgroups (N = 512 projections at 64 slates per GPU: 240 tiles) -- twice the workgroups, four waves per
// SIMD to hide the staging latency that two waves leave exposed
template <bool TAIL, int NT, int IMG>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
ltrx_gemm_nt64_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc, int M,
                      int N, int K, const float* __restrict__ bias, int act, const float* __restrict__ aux, int ldaux, int tiles_n,
                      ltrx::DropSpec drop, const uint32_t* __restrict__ drop_step, int cimg) {
  nt256_body<TAIL, 64, NT, IMG>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, tiles_n, drop, drop_step, cimg);
}

// ------------------------------------------------------------------------------------------------------------------
// TN (weight gradient):  C[N',K'] = sum_m A[m][n'] * B[m][k'],  split over m into `splits` slabs
// ------------------------------------------------------------------------------------------------------------------
template <int NTERMS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) ltrx_gemm_tn_kernel(const float* __restrict__ A, int lda,
                                                              const float* __restrict__ B, int ldb,
                                                              float* __restrict__ slabs, float* __restrict__ bias_slabs,
                                                              int M, int NP, int KP, int tiles_k, int m_per_split) {
  constexpr int BM = 128, BK = 32;
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

  const bool cola = n0 + scol < NP, colb = k0 + scol < KP;
  const float* pA = A + (cola ? n0 + scol : NP - 1);
  const float* pB = B + (colb ? k0 + scol : KP - 1);
  int mt_held = 0;                   // first contraction row of the tile currently held in ra/rb
  auto gload = [&](int mt) {         // raw loads (clamped rows); zero-select deferred to sstore
    mt_held = mt;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = min(mt + 8 * p + 4 * sg + e, mend - 1);
        ra[p][e] = pA[(size_t)m * lda];
        rb[p][e] = pB[(size_t)m * ldb];
      }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool okm = mt_held + 8 * p + 4 * sg + e < mend;
        ra[p][e] = (okm && cola) ? ra[p][e] : 0.f;
        rb[p][e] = (okm && colb) ? rb[p][e] : 0.f;
      }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int o = swz_off<BK>(scol, 8 * p + 4 * sg);
      bf16x4 h, l, l2;
      split4<NTERMS>(make_float4(ra[p][0], ra[p][1], ra[p][2], ra[p][3]), h, l, l2);
      *reinterpret_cast<bf16x4*>(&s.a[0][o]) = h;
      if (NTERMS >= 2) *reinterpret_cast<bf16x4*>(&s.a[NTERMS >= 2 ? 1 : 0][o]) = l;
      if (NTERMS == 3) *reinterpret_cast<bf16x4*>(&s.a[NTERMS - 1][o]) = l2;
      split4<NTERMS>(make_float4(rb[p][0], rb[p][1], rb[p][2], rb[p][3]), h, l, l2);
      *reinterpret_cast<bf16x4*>(&s.b[0][o]) = h;
      if (NTERMS >= 2) *reinterpret_cast<bf16x4*>(&s.b[NTERMS >= 2 ? 1 : 0][o]) = l;
      if (NTERMS == 3) *reinterpret_cast<bf16x4*>(&s.b[NTERMS - 1][o]) = l2;
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

// ------------------------------------------------------------------------------------------------------------------
// TN, large-tile variant (weight gradients of the big projections): 256 x 256 output tile, 8 waves (wave tile 128 x 64),
// K-step = 32 contraction rows, two LDS stages + LDS-only barrier, split-K slabs as above.
// Staging: 8 consecutive lanes fetch 128 contiguous bytes of one contraction row (16-byte loads, 4x fewer VMEM
// instructions than the dword loads above); a thread owns a 4 (m) x 4 (column) block of A and of B, i.e. after the split
// one 8-byte LDS write per column lands at [column][m .. m+3].  Rows r and r + 16 of an image swap bank halves (swz_t) so
// that the 16 lanes of a store group (8 column groups x 2 m-groups) hit 16 distinct bank pairs; the fragment reads stay
// conflict-free under that row permutation.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz_t(int row, int k) {
  const int c = (k >> 3) ^ ((row >> 2) & 3);
  return (row ^ ((row >> 4) & 1)) * 32 + c * 8 + (k & 7);
}

// Up to LTRX_TN_GROUP weight gradients over the SAME rows (the four projections of an encoder layer: dW = dY^T X with their own dY, X
// and output shape) in ONE launch: the tiles of all problems form one grid, so the chip is filled with total_tiles x splits workgroups
// instead of tiles x splits per problem -- 4x fewer splits, i.e. 4x fewer partial slabs to write and to reduce (round 4: the slabs were
// 550 MB per step whatever the batch; ltrx_gemm_tn_group).
#define LTRX_TN_GROUP 4
struct TnGroup {
  const float* A[LTRX_TN_GROUP];
  const float* B[LTRX_TN_GROUP];
  float* slabs[LTRX_TN_GROUP];
  float* bias_slabs[LTRX_TN_GROUP];
  int lda[LTRX_TN_GROUP], ldb[LTRX_TN_GROUP], NP[LTRX_TN_GROUP], KP[LTRX_TN_GROUP], tiles_k[LTRX_TN_GROUP];
  int kv[LTRX_TN_GROUP];             // columns of B that exist (= row length of the slabs and of C); KP = kv rounded up to the tile
  int tile_start[LTRX_TN_GROUP + 1];
  int nprob;
  unsigned char bimg[LTRX_TN_GROUP];   // operand B of the problem is a pre-split image (an activation written that way by its producer)
  // grouped launches (nprob > 1, at most 256 workgroups): workgroup id -> (tile, split), built by the host so that the tiles of one
  // (problem, split) -- which share their two operand row slabs -- sit on ONE XCD (workgroup w is dispatched to XCD w % 8)
  unsigned char map_tile[256];
  unsigned char map_split[256];
};

template <int NT>
__global__ void __launch_bounds__(512) ltrx_gemm_tn256_kernel(const TnGroup grp, int M, int m_per_split) {
  constexpr int BK_ = 32;
  constexpr int L1 = NT - 1;
  typedef SmemNT<256, NT> SmemT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SmemT* s = reinterpret_cast<SmemT*>(smem_raw);
  // Workgroup -> (split, tile) in split-major order through the XCD remap: all tiles of one m-slab run on ONE XCD, so the dY
  // slab (shared by the tiles of a tile row) and the X slab (shared by the tiles of a tile column) are fetched from HBM once
  // and served to the other tiles by that XCD's L2.  PMC at the FFN shape (profiles/r01_pmc_tn256_xcd.md): L2 hit rate
  // 27 % -> 68 %, HBM-side reads 1.5 GB -> 0.63 GB per launch (= the algorithmic bytes); the duration does not change (the
  // kernel is bound by its staging path, not by HBM), the freed bandwidth is what the overlapped all-reduce needs.
  const int n_tiles = gridDim.x;
  int gtile, split;
  if (grp.nprob > 1) {
    const int w = blockIdx.x + n_tiles * blockIdx.y;
    gtile = grp.map_tile[w];
    split = grp.map_split[w];
  } else {
    const int wg = xcd_remap(blockIdx.x + n_tiles * blockIdx.y, n_tiles * gridDim.y);
    gtile = wg % n_tiles;
    split = wg / n_tiles;
  }
  int pi = 0;                                                          // (workgroup-uniform)
  while (pi + 1 < grp.nprob && gtile >= grp.tile_start[pi + 1]) ++pi;
  const float* __restrict__ A = grp.A[pi];
  const float* __restrict__ B = grp.B[pi];
  float* __restrict__ slabs = grp.slabs[pi];
  float* __restrict__ bias_slabs = grp.bias_slabs[pi];
  const int lda = grp.lda[pi], ldb = grp.ldb[pi], NP = grp.NP[pi], tiles_k = grp.tiles_k[pi], kv = grp.kv[pi];
  const bool bimg = grp.bimg[pi] != 0;                                  // (workgroup-uniform)
  const int tile = gtile - grp.tile_start[pi];
  const int n0 = (tile / tiles_k) * 256, k0 = (tile % tiles_k) * 256;
  const int wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int mbeg = split * m_per_split, mend = min(M, mbeg + m_per_split);
  const int cg = (threadIdx.x & 7) + 8 * wave, mg = (threadIdx.x >> 3) & 7;
  const float* PA = A + n0 + (size_t)(4 * mg) * lda + 4 * cg;
  const float* PB = B + k0 + (size_t)(4 * mg) * ldb + 4 * cg;
  float4 r[2][4];                    // [operand][m row]
  auto gload1 = [&](int mt, int g) {          // operand g of the m-slab starting at row mt
#pragma unroll
    for (int e = 0; e < 4; ++e)
      r[g][e] = g ? *reinterpret_cast<const float4*>(PB + (size_t)(mt + e) * ldb)
                  : *reinterpret_cast<const float4*>(PA + (size_t)(mt + e) * lda);
  };
  auto gload = [&](int mt) {
    gload1(mt, 0);
    gload1(mt, 1);
  };
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_bias = bias_slabs != nullptr && (tile % tiles_k) == 0;      // column sums of A = the bias gradient
  auto sstore1 = [&](SmemT& d, int g) {
    if (g == 1 && bimg) {
      // B is an image: row e of the thread's 4 (m) x 4 (column) block arrives as {hi(c0,c1), hi(c2,c3), lo(c0,c1), lo(c2,c3)} -- four
      // dwords of two bf16 each.  The LDS image wants [column][m .. m+3]: a 4 x 4 transpose of 16-bit values, two v_perm_b32 per
      // (column, term) instead of the split's ~6 VALU operations per element.
      unsigned int wv[4][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        wv[e][0] = __float_as_uint(r[1][e].x); wv[e][1] = __float_as_uint(r[1][e].y);
        wv[e][2] = __float_as_uint(r[1][e].z); wv[e][3] = __float_as_uint(r[1][e].w);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned int sel = (c & 1) ? 0x07060302u : 0x05040100u;     // high / low halves of the two source dwords
        const int hw = c >> 1;                                            // dword holding the hi pair of columns (c, c^1); lo pair: + 2
        uint2 hq, lq;
        hq.x = __builtin_amdgcn_perm(wv[1][hw], wv[0][hw], sel);
        hq.y = __builtin_amdgcn_perm(wv[3][hw], wv[2][hw], sel);
        lq.x = __builtin_amdgcn_perm(wv[1][hw + 2], wv[0][hw + 2], sel);
        lq.y = __builtin_amdgcn_perm(wv[3][hw + 2], wv[2][hw + 2], sel);
        const int o = swz_t(4 * cg + c, 4 * mg);
        *reinterpret_cast<uint2*>(&d.b[0][o]) = hq;
        if (NT == 2) *reinterpret_cast<uint2*>(&d.b[L1][o]) = lq;
      }
    } else {
      __bf16* img0 = g ? d.b[0] : d.a[0];
      __bf16* img1 = g ? d.b[L1] : d.a[L1];
      const float cx[4][4] = {{r[g][0].x, r[g][1].x, r[g][2].x, r[g][3].x}, {r[g][0].y, r[g][1].y, r[g][2].y, r[g][3].y},
                              {r[g][0].z, r[g][1].z, r[g][2].z, r[g][3].z}, {r[g][0].w, r[g][1].w, r[g][2].w, r[g][3].w}};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x4 h, l, l2;
        split4<2>(make_float4(cx[c][0], cx[c][1], cx[c][2], cx[c][3]), h, l, l2);
        const int o = swz_t(4 * cg + c, 4 * mg);
        *reinterpret_cast<bf16x4*>(&img0[o]) = h;
        if (NT == 2) *reinterpret_cast<bf16x4*>(&img1[o]) = l;
        if (g == 0 && want_bias) bsum[c] += (cx[c][0] + cx[c][1]) + (cx[c][2] + cx[c][3]);
      }
    }
  };
  auto sstore = [&](SmemT& d) {
    sstore1(d, 0);
    sstore1(d, 1);
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  auto mma1 = [&](const SmemT& t, int ks) {
    {
      bf16x8 af[NT][4], bfr[NT][2];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bfr[tt][j] = *reinterpret_cast<const bf16x8*>(&t.b[tt][swz_t(wc * 64 + j * 32 + l31, ks * 16 + 8 * half)]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          af[tt][i] = *reinterpret_cast<const bf16x8*>(&t.a[tt][swz_t(wr * 128 + i * 32 + l31, ks * 16 + 8 * half)]);
      }
#define LTRX_MMA256(TA, TB)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA][i], bfr[TB][j], acc[i][j], 0, 0, 0);
      if (NT == 2) {
        LTRX_MMA256(0, L1)
        LTRX_MMA256(L1, 0)
      }
      LTRX_MMA256(0, 0)
#undef LTRX_MMA256
    }
  };
  auto mma = [&](const SmemT& t) {
    mma1(t, 0);
    mma1(t, 1);
  };
  const int nk = (mend - mbeg) / BK_;            // M and m_per_split are multiples of 32 (host)
  if (nk > 0) {
    gload(mbeg);
    sstore(s[0]);
    if (nk > 1) gload(mbeg + BK_);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
      sstore(s[(kt + 1) & 1]);
      // pinned: left to itself the scheduler sinks these loads below the MFMA block (shorter live ranges), i.e. to a few
      // hundred cycles before the wait at the top of the next iteration, and every K-step eats the full memory latency
      __builtin_amdgcn_sched_barrier(0);
      gload(mbeg + (kt + 2) * BK_);
      __builtin_amdgcn_sched_barrier(0);
      mma(s[kt & 1]);
      lds_only_barrier();
    }
    for (; kt < nk; ++kt) {
      mma(s[kt & 1]);
      if (kt + 1 < nk) sstore(s[(kt + 1) & 1]);
      __syncthreads();
    }
  }
  if (want_bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = bsum[c];
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (mg == 0) bias_slabs[(size_t)split * NP + n0 + 4 * cg + c] = v;
    }
  }
  // (kv < the tile's column range: B's row stride covers the whole tile -- the host checks -- but only kv columns exist; the
  //  others were read from the padding and their products are dropped here)
  float* slab = slabs + (size_t)split * NP * kv;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = k0 + wc * 64 + j * 32 + l31;
    if (col < kv) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = n0 + wr * 128 + i * 32 + rowmap(q, half);
          slab[(size_t)row * kv + col] = acc[i][j][q];
        }
    }
  }
}

// Fixed-order sums of partial slabs, as device functions shared by the per-GEMM launch and the grouped launch (same order of
// additions in both: a result does not depend on which launch produced it).
// columns kind: out[c] = sum_s src[s * row_stride + c] for 64 columns per workgroup; its 4 waves take every 4th slab with 8 loads in
// flight, partials combined through LDS in a fixed order (one thread per column looping over all slabs was up to 128
// dependent-latency iterations in a handful of workgroups: the critical path of the launch)
__device__ __forceinline__ void reduce_cols(const float* __restrict__ src, int splits, size_t n, size_t row_stride,
                                            float* __restrict__ out, int lb, float (*sh)[64]) {
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const size_t c = (size_t)lb * 64 + cl;
  float a = 0.f;
  if (c < n) {
    int sidx = rg;
    for (; sidx + 28 < splits; sidx += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(sidx + 4 * u) * row_stride + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; sidx < splits; sidx += 4) a += src[(size_t)sidx * row_stride + c];
  }
  sh[rg][cl] = a;
  __syncthreads();
  if (rg == 0 && c < n) out[c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
// wide kind: C[i] = sum_s slabs[s * n + i], grid-stride over `nb` workgroups (this one is `lb`)
__device__ __forceinline__ void reduce_wide(const float* __restrict__ slabs, int splits, size_t n, float* __restrict__ C, int lb, int nb) {
  const size_t stride = (size_t)nb * blockDim.x;
  if ((n & 3) == 0 && (((uintptr_t)slabs | (uintptr_t)C) & 15) == 0) {
    // 16-byte accesses, eight slabs in flight per thread; the additions keep the slab order (same bits as the scalar loop)
    const size_t n4 = n >> 2;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(slabs);
    for (size_t i = (size_t)lb * blockDim.x + threadIdx.x; i < n4; i += stride) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      int sidx = 0;
      for (; sidx + 8 <= splits; sidx += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(s4 + (size_t)(sidx + u) * n4 + i);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
      }
      for (; sidx < splits; ++sidx) a += __builtin_nontemporal_load(s4 + (size_t)sidx * n4 + i);
      reinterpret_cast<f32x4*>(C)[i] = a;
    }
    return;
  }
  for (size_t i = (size_t)lb * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = 0.f;
    for (int sidx = 0; sidx < splits; ++sidx) a += slabs[(size_t)sidx * n + i];
    C[i] = a;
  }
}

// One launch serves the weight gradient (workgroups 0 .. main_blocks-1) and, when given, the bias gradient's column-sum slabs
// (the remaining workgroups).
__global__ void __launch_bounds__(256) ltrx_gemm_slab_reduce_kernel(const float* __restrict__ slabs, int splits,
                                                                    size_t n, float* __restrict__ C, int main_blocks,
                                                                    const float* __restrict__ slabs2, int splits2, size_t n2,
                                                                    float* __restrict__ C2) {
  __shared__ float sh[4][64];
  if ((int)blockIdx.x >= main_blocks) {
    reduce_cols(slabs2, splits2, n2, n2, C2, (int)blockIdx.x - main_blocks, sh);
    return;
  }
  reduce_wide(slabs, splits, n, C, (int)blockIdx.x, main_blocks);
}

static size_t slab_reduce_blocks(size_t n) {
  size_t blocks = ((n & 3) == 0 ? n / 4 : n) / 256 + 1;
  return blocks > 2048 ? 2048 : blocks;
}
static void launch_slab_reduce(const float* slabs, int splits, size_t n, float* C, const float* bslabs, int bsplits, size_t nb,
                               float* bias_out, hipStream_t s) {
  const size_t blocks = slab_reduce_blocks(n);
  const size_t bblocks = bias_out ? (nb + 63) / 64 : 0;
  hipLaunchKernelGGL(ltrx_gemm_slab_reduce_kernel, dim3((unsigned)(blocks + bblocks)), dim3(256), 0, s, slabs, splits, n, C,
                     (int)blocks, bslabs, bsplits, nb, bias_out);
}

// Several independent fixed-order reductions in ONE launch (ltrx_reduce_group): the partial slabs of a grouped weight-gradient GEMM,
// their bias column sums, and the parameter-gradient partials of the LayerNorm backward kernels of the same encoder layer -- eleven
// launches of 5-12 us per layer become one.  Entry e owns workgroups blk_start[e] .. blk_start[e+1]-1.
#define LTRX_REDUCE_GROUP 16
struct RedGroup {
  const float* src[LTRX_REDUCE_GROUP];
  float* dst[LTRX_REDUCE_GROUP];
  unsigned long long n[LTRX_REDUCE_GROUP];
  unsigned long long row_stride[LTRX_REDUCE_GROUP];
  int splits[LTRX_REDUCE_GROUP];
  int wide[LTRX_REDUCE_GROUP];
  int blk_start[LTRX_REDUCE_GROUP + 1];
  int nent;
};
__global__ void __launch_bounds__(256) ltrx_reduce_group_kernel(const RedGroup g) {
  __shared__ float sh[4][64];
  int e = 0;                                                            // (workgroup-uniform)
  while (e + 1 < g.nent && (int)blockIdx.x >= g.blk_start[e + 1]) ++e;
  const int lb = (int)blockIdx.x - g.blk_start[e];
  if (g.wide[e])
    reduce_wide(g.src[e], g.splits[e], (size_t)g.n[e], g.dst[e], lb, g.blk_start[e + 1] - g.blk_start[e]);
  else
    reduce_cols(g.src[e], g.splits[e], (size_t)g.n[e], (size_t)g.row_stride[e], g.dst[e], lb, sh);
}

// dst[i][c] = sum_{s < splits[i]} src[i][s * row_stride[i] + c]  for c < cols[i], i < n: every sum in a fixed order (deterministic;
// an entry gives the same bits as ltrx_gemm_tn's own reduction of the same slabs).  Entries with splits[i] <= 0 are skipped.
extern "C" int ltrx_reduce_group(int n, const float* const* src, const int* splits, const size_t* row_stride, const size_t* cols,
                                 float* const* dst, ltrx_stream_t stream) {
  if (n < 0 || n > LTRX_REDUCE_GROUP || (n > 0 && (!src || !splits || !row_stride || !cols || !dst))) return LTRX_EINVAL;
  RedGroup g = {};
  int ne = 0, blk = 0;
  for (int i = 0; i < n; ++i) {
    if (splits[i] <= 0 || cols[i] == 0) continue;
    if (!src[i] || !dst[i] || row_stride[i] < cols[i]) return LTRX_EINVAL;
    g.src[ne] = src[i];
    g.dst[ne] = dst[i];
    g.n[ne] = cols[i];
    g.row_stride[ne] = row_stride[i];
    g.splits[ne] = splits[i];
    g.wide[ne] = (row_stride[i] == cols[i] && cols[i] >= 16384) ? 1 : 0;
    g.blk_start[ne] = blk;
    blk += (int)(g.wide[ne] ? slab_reduce_blocks(cols[i]) : (cols[i] + 63) / 64);
    ++ne;
  }
  g.blk_start[ne] = blk;
  g.nent = ne;
  if (ne == 0) return LTRX_OK;
  hipLaunchKernelGGL(ltrx_reduce_group_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, g);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// pre-split operand image: every 4 consecutive floats of src become the 16 bytes {hi0..hi3, lo0..lo3} (bf16) of dst -- the same
// split4 the kernels apply while staging, so a GEMM fed from the image is bit-identical to one fed from the fp32 values
__global__ void __launch_bounds__(256) ltrx_split_image_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    bf16x4 h, l, l2;
    split4<2>(src[i], h, l, l2);
    float4 o;
    *reinterpret_cast<bf16x4*>(&o.x) = h;
    *reinterpret_cast<bf16x4*>(&o.z) = l;
    dst[i] = o;
  }
}

extern "C" int ltrx_split_image(const float* src, void* dst, size_t n, ltrx_stream_t stream) {
  if (!src || !dst || (n & 3) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return LTRX_EINVAL;
  if (n == 0) return LTRX_OK;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ltrx_split_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n / 4);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
// `tile` argument of ltrx_gemm_nt / ltrx_gemm_tn (a per-call tuning argument, no process state): 0 = auto; 1 = 128x128x32
// (4 waves); 2 = 128x128x64; 3 = 256x128x32 (8 waves); 4 = 256x128x64; 6 = 256x256x32 large-tile kernel (auto picks it for
// exact multiples with >= 360 tiles); 7 = its 128x256x32 form; 8 = its 64x256x32 form with two workgroups per CU (small batches);
// +100 = tuning experiment (every operand row aliases row 0)
// (a PERSISTENT form of the 256x256x32 kernel -- one workgroup per CU walking its tiles, next tile's first K-steps prefetched
//  behind the epilogue -- was built and measured in round 3: bit-identical results, 0 ... -13 % in speed; tools/lab/gemm_persist.inc,
//  profiles/r03_gemm_persist_ab.md)

template <int NTERMS, int BM_, int BK_>
static void launch_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                      const float* bias, int act, const float* aux, int ldaux, ltrx::DropSpec drop, const uint32_t* drop_step,
                      hipStream_t s) {
  const int tiles_m = (M + BM_ - 1) / BM_, tiles_n = (N + BN - 1) / BN;
  hipLaunchKernelGGL((ltrx_gemm_nt_kernel<NTERMS, BM_, BK_>), dim3(tiles_m * tiles_n), dim3(BM_ * 2), 0, s, A, lda, B, ldb, C,
                     ldc, M, N, K, bias, act, aux, ldaux, tiles_n, drop, drop_step);
}

// bytes of the one-bit ReLU mask of an [M, N] activation (acts 4 / 5 of ltrx_gemm_nt), 0 where that form does not apply: the mask is
// kept in the 256 x 256 tile kernel's (tile, thread) order, so both launches must take that kernel whatever their K, dropout and
// epilogue -- N a multiple of 256, K a multiple of the kernel's 32-column step (K of the launch asked about: the forward and the
// input-gradient launch of one mask may contract over different widths -- ask for both) and a tile count the dispatch below sends
// there unconditionally.  One predicate mirrors the dispatch: callers fall back to acts 1 / 2 where it returns 0.
extern "C" size_t ltrx_gemm_nt_relu_bits_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || (N % 256) || (K % 32)) return 0;
  const size_t t = (size_t)((M + 255) / 256) * (N / 256);
  if (!(t >= 380 || (t >= 168 && t <= 256))) return 0;
  return t * 512 * 16;
}

// Would ltrx_gemm_nt_img accept operand / output IMAGES for this shape -- i.e. does the automatic tile choice land on the large-tile
// kernel family (256 / 128 / 64-row tiles) for every epilogue?  A producer that writes an activation as an image asks this for each
// GEMM that will consume it, BEFORE writing it (the image has no fp32 copy).  Mirrors the dispatch of ltrx_gemm_nt_img below.
extern "C" int ltrx_gemm_nt_image_ok(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || (N % 256) || (K % 32)) return 0;
  const size_t t = (size_t)((M + 255) / 256) * (N / 256);
  if (t > 256 && t < 360) return 0;                    // (with dropout in the epilogue this range takes the small-tile kernel)
  if (t >= 360 && t < 380 && N / 256 <= 256) {         // may run as one full round of large tiles + the remaining rows
    const int m1 = (256 / (N / 256)) * 256;
    if (M - m1 > 0 && !ltrx_gemm_nt_image_ok(M - m1, N, K)) return 0;
  }
  if (t >= 360 || (t >= 168 && t <= 256)) return 1;
  if (t >= 136 && t < 168 && (size_t)((M + 127) / 128) * (N / 256) > 256) return 1;
  const size_t t128 = (size_t)((M + 127) / 128) * (N / 256);
  if (t128 >= 176 && t128 <= 256) return 1;
  if (t128 < 176) {
    const size_t t64 = (size_t)((M + 63) / 64) * (N / 256);
    if (t64 >= 176 && t64 <= 512) return 1;
  }
  return 0;
}

extern "C" int ltrx_gemm_nt_img(const float* A, int lda, const float* B, int ldb, const void* B_image, float* C, int ldc, int M, int N, int K,
                                const float* bias, int act, const float* aux, int ldaux, float drop_p, uint32_t drop_seed,
                                const uint32_t* drop_step, int strict, int tile, int operand_flags, ltrx_stream_t stream);

extern "C" int ltrx_gemm_nt(const float* A, int lda, const float* B, int ldb, const void* B_image, float* C, int ldc, int M, int N, int K,
                            const float* bias, int act, const float* aux, int ldaux, float drop_p, uint32_t drop_seed,
                            const uint32_t* drop_step, int strict, int tile, ltrx_stream_t stream) {
  return ltrx_gemm_nt_img(A, lda, B, ldb, B_image, C, ldc, M, N, K, bias, act, aux, ldaux, drop_p, drop_seed, drop_step, strict, tile, 0,
                          stream);
}

extern "C" int ltrx_gemm_nt_img(const float* A, int lda, const float* B, int ldb, const void* B_image, float* C, int ldc, int M, int N, int K,
                                const float* bias, int act, const float* aux, int ldaux, float drop_p, uint32_t drop_seed,
                                const uint32_t* drop_step, int strict, int tile, int operand_flags, ltrx_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 5 || tile < 0) return LTRX_EINVAL;
  if (operand_flags & ~(LTRX_GEMM_A_IS_IMAGE | LTRX_GEMM_C_AS_IMAGE)) return LTRX_EINVAL;
  const bool aimg = (operand_flags & LTRX_GEMM_A_IS_IMAGE) != 0;
  const int cimg = (operand_flags & LTRX_GEMM_C_AS_IMAGE) ? 1 : 0;
  // images exist in the large-tile kernel family only; an image of A goes with an image of B (the engine's weights always have one)
  if (operand_flags && (strict == 1 || ((uintptr_t)A & 15) || ((uintptr_t)C & 15))) return LTRX_EUNSUPPORTED;
  if (aimg && (!B_image || (((uintptr_t)B_image) & 15))) return LTRX_EUNSUPPORTED;
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return LTRX_EINVAL;
  const ltrx::DropSpec drop = ltrx_make_drop(drop_p, drop_seed);
  if ((act == 2 || act == 3) && (!aux || ldaux < N)) return LTRX_EINVAL;
  if (act == 4 || act == 5) {        // the one-bit mask lives in the large-tile kernel's own (tile, thread) order: only where it runs
    if (!aux || ((uintptr_t)aux & 15) || tile != 0 || strict == 1 || ltrx_gemm_nt_relu_bits_bytes(M, N, K) == 0) return LTRX_EUNSUPPORTED;
    ldaux = 0;
  }
  if ((K & 3) || (lda & 3) || (ldb & 3) || lda < K || ldb < K || ldc < N) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int v = tile;
  if (v >= 100) {            // tuning experiment: all rows alias row 0 -> every operand load is an L2 hit (results are garbage)
    v -= 100;
    lda = 0;
    ldb = 0;
  }
  // large-tile kernel: exact multiples only, and enough tiles to cover the 256 CUs at least ~1.4 times
  const bool vec_epi = (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
                       (!aux || ((ldaux & 3) == 0 && ((uintptr_t)aux & 15) == 0));      // 16-byte epilogue accesses
  if ((act == 4 || act == 5) && (!vec_epi || (K % 32))) return LTRX_EUNSUPPORTED;
  const bool plain = strict == 2;                     // precision code: 0 = three products, 1 = six (strict), 2 = one (plain bf16)
  if (strict == 2) strict = 0;
  if (v == 0 && !strict && (N % 256) == 0 && (K % 32) == 0 && vec_epi) {
    // one workgroup per CU: a grid that fills 3/4 .. 1 round, or at least ~1.4 rounds (measured, tools/gemm_variants.py:
    // 240 tiles 53 vs 77 us, 360 tiles parity, 120 tiles parity, 480 tiles 102 vs 132 us)
    const size_t t = (size_t)((M + 255) / 256) * (N / 256);
    if (t > 256 && t < 380 && (drop.thresh == 0u || act == 2) && N / 256 <= 256) {
      // between one and ~1.5 rounds of 256-row tiles: a second round for a handful of tiles costs as much as the first
      // (M 33000 x N 512: 95 us against 54 us for M 30720).  One exact round of large tiles first, the remaining rows as
      // their own (smaller) problem: 5-25 % faster over that range (tools/gemm_split_probe.py).  Only without dropout
      // generated in the epilogue -- its hash is indexed by the row of THIS launch.
      const int m1 = (256 / (N / 256)) * 256;
      const int prec = plain ? 2 : strict;
      int rc = ltrx_gemm_nt_img(A, lda, B, ldb, B_image, C, ldc, m1, N, K, bias, act, aux, ldaux, drop_p, drop_seed, drop_step, prec, 0,
                                operand_flags, stream);
      if (rc != LTRX_OK) return rc;
      return ltrx_gemm_nt_img(A + (size_t)m1 * lda, lda, B, ldb, B_image, C + (size_t)m1 * ldc, ldc, M - m1, N, K, bias, act,
                              aux ? aux + (size_t)m1 * ldaux : nullptr, ldaux, drop_p, drop_seed, drop_step, prec, 0, operand_flags, stream);
    }
    if (t >= 360 || (t >= 168 && t <= 256)) v = 6;
    else if (t >= 136 && t < 168 && (size_t)((M + 127) / 128) * (N / 256) > 256) v = 6;   // one partial round still beats two rounds of smaller tiles
    else {                                           // 128-row tiles when they make exactly one well-filled round
      const size_t t128 = (size_t)((M + 127) / 128) * (N / 256);
      if (t128 >= 176 && t128 <= 256) v = 7;
      else if (t128 < 176) {                         // small batches (32 slates x 240 per GPU: 120 tiles of 128 rows): 64-row tiles, two
        const size_t t64 = (size_t)((M + 63) / 64) * (N / 256);   // workgroups per CU -- 55 vs 77 (128-row) vs 108 us (small-tile kernel)
        if (t64 >= 176 && t64 <= 512) v = 8;         // at M 7680 x N 512 x K 2048 (tools/gemm_nt64_ab.py); same bits as every other tile
      }
    }
  }
  if (v == 0) v = 1;
  // (the mask's (tile, thread) order is the 256-row kernel's: refuse rather than run any other tiling if the predicate above and this
  //  dispatch ever disagree)
  if ((act == 4 || act == 5) && v != 6) return LTRX_EUNSUPPORTED;
  if (v == 8) {              // 64-row tiles, two workgroups per CU
    if ((N % 256) || (K % 32) || strict || !vec_epi || act == 4 || act == 5) return LTRX_EUNSUPPORTED;
    static std::atomic<uint64_t> attr64_done{0};
    const int arc = ltrx_once_per_device(attr64_done, []() {
#define LTRX_NT64_ATTR(TAIL_, NT_, IMG_)                                                                                  \
  (hipFuncSetAttribute((const void*)ltrx_gemm_nt64_kernel<TAIL_, NT_, IMG_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                       (int)(2 * sizeof(SmemNT<64, NT_>))) != hipSuccess)
      if (LTRX_NT64_ATTR(false, 2, 0) || LTRX_NT64_ATTR(false, 2, 1) || LTRX_NT64_ATTR(false, 2, 2) || LTRX_NT64_ATTR(true, 2, 0) ||
          LTRX_NT64_ATTR(true, 2, 1) || LTRX_NT64_ATTR(true, 2, 2) || LTRX_NT64_ATTR(false, 1, 0) || LTRX_NT64_ATTR(false, 1, 1) ||
          LTRX_NT64_ATTR(false, 1, 2) || LTRX_NT64_ATTR(true, 1, 0) || LTRX_NT64_ATTR(true, 1, 1) || LTRX_NT64_ATTR(true, 1, 2))
        return LTRX_EHIP;
#undef LTRX_NT64_ATTR
      return LTRX_OK;
    });
    if (arc != LTRX_OK) return arc;
    const int tiles_n = N / 256;
    const dim3 grid(((M + 63) / 64) * tiles_n);
    const bool bimg = B_image != nullptr && (((uintptr_t)B_image) & 15) == 0;
    const float* Bk = bimg ? reinterpret_cast<const float*>(B_image) : B;
#define LTRX_NT64_(TAIL_, NT_, IMG_)                                                                                        \
  hipLaunchKernelGGL((ltrx_gemm_nt64_kernel<TAIL_, NT_, IMG_>), grid, dim3(512), 2 * sizeof(SmemNT<64, NT_>), s, A, lda, Bk, ldb, C, \
                     ldc, M, N, K, bias, act, aux, ldaux, tiles_n, drop, drop_step, cimg)
#define LTRX_NT64(TAIL_)                                                                                                    \
  do {                                                                                                                      \
    if (plain) {                                                                                                            \
      if (aimg) LTRX_NT64_(TAIL_, 1, 2); else if (bimg) LTRX_NT64_(TAIL_, 1, 1); else LTRX_NT64_(TAIL_, 1, 0);              \
    } else {                                                                                                                \
      if (aimg) LTRX_NT64_(TAIL_, 2, 2); else if (bimg) LTRX_NT64_(TAIL_, 2, 1); else LTRX_NT64_(TAIL_, 2, 0);              \
    }                                                                                                                       \
  } while (0)
    if (M % 64) LTRX_NT64(true); else LTRX_NT64(false);
#undef LTRX_NT64_
#undef LTRX_NT64
    LTRX_LAUNCH_CHECK();
    return LTRX_OK;
  }
  if (v == 6 || v == 7) {
    if ((N % 256) || (K % 32) || strict || !vec_epi) return LTRX_EUNSUPPORTED;
    static std::atomic<uint64_t> attr_done{0};
    const int arc = ltrx_once_per_device(attr_done, []() {
#define LTRX_NT256_ATTR(TAIL_, BM_, NT_)                                                                                     \
  (hipFuncSetAttribute((const void*)ltrx_gemm_nt256_kernel<TAIL_, BM_, NT_, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                       (int)(2 * sizeof(SmemNT<BM_, NT_>))) != hipSuccess ||                                               \
   hipFuncSetAttribute((const void*)ltrx_gemm_nt256_kernel<TAIL_, BM_, NT_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                       (int)(2 * sizeof(SmemNT<BM_, NT_>))) != hipSuccess ||                                               \
   hipFuncSetAttribute((const void*)ltrx_gemm_nt256_kernel<TAIL_, BM_, NT_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                       (int)(2 * sizeof(SmemNT<BM_, NT_>))) != hipSuccess)
      if (LTRX_NT256_ATTR(false, 256, 2) || LTRX_NT256_ATTR(true, 256, 2) || LTRX_NT256_ATTR(false, 128, 2) ||
          LTRX_NT256_ATTR(true, 128, 2) || LTRX_NT256_ATTR(false, 256, 1) || LTRX_NT256_ATTR(true, 256, 1) ||
          LTRX_NT256_ATTR(false, 128, 1) || LTRX_NT256_ATTR(true, 128, 1))
        return LTRX_EHIP;
#undef LTRX_NT256_ATTR
      return LTRX_OK;
    });
    if (arc != LTRX_OK) return arc;
    const int tiles_n = N / 256;
    const int bm = (v == 6) ? 256 : 128;
    const dim3 grid(((M + bm - 1) / bm) * tiles_n);
    // the pre-split image of B (same addressing as B, 16-byte aligned) replaces the fp32 operand in this kernel family only
    const bool bimg = B_image != nullptr && (((uintptr_t)B_image) & 15) == 0;
    const float* Bk = bimg ? reinterpret_cast<const float*>(B_image) : B;
#define LTRX_NT256_(TAIL_, BM_, NT_, IMG_)                                                                                   \
  hipLaunchKernelGGL((ltrx_gemm_nt256_kernel<TAIL_, BM_, NT_, IMG_>), grid, dim3(512), 2 * sizeof(SmemNT<BM_, NT_>), s, A, lda, Bk, \
                     ldb, C, ldc, M, N, K, bias, act, aux, ldaux, tiles_n, drop, drop_step, cimg)
#define LTRX_NT256(TAIL_, BM_)                                                                                               \
  do {                                                                                                                       \
    if (plain) {                                                                                                             \
      if (aimg) LTRX_NT256_(TAIL_, BM_, 1, 2); else if (bimg) LTRX_NT256_(TAIL_, BM_, 1, 1); else LTRX_NT256_(TAIL_, BM_, 1, 0); \
    } else {                                                                                                                 \
      if (aimg) LTRX_NT256_(TAIL_, BM_, 2, 2); else if (bimg) LTRX_NT256_(TAIL_, BM_, 2, 1); else LTRX_NT256_(TAIL_, BM_, 2, 0); \
    }                                                                                                                        \
  } while (0)
    if (v == 6) {
      if (M % 256) LTRX_NT256(true, 256); else LTRX_NT256(false, 256);
    } else {
      if (M % 128) LTRX_NT256(true, 128); else LTRX_NT256(false, 128);
    }
#undef LTRX_NT256_
#undef LTRX_NT256
    LTRX_LAUNCH_CHECK();
    return LTRX_OK;
  }
  if (operand_flags) return LTRX_EUNSUPPORTED;        // (ltrx_gemm_nt_image_ok said so: no image form in the small-tile kernels)
  if (strict) {
    launch_nt<3, 128, 32>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, drop, drop_step, s);
  } else if (plain) {
    launch_nt<1, 128, 32>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, drop, drop_step, s);
  } else {
    switch (v) {
      case 2: launch_nt<2, 128, 64>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, drop, drop_step, s); break;
      case 3: launch_nt<2, 256, 32>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, drop, drop_step, s); break;
      case 4: launch_nt<2, 256, 64>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, drop, drop_step, s); break;
      default: launch_nt<2, 128, 32>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, drop, drop_step, s); break;
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

// large-tile wgrad kernel: exact multiples only
static bool tn256_ok(int M, int NP, int KP) { return (NP % 256) == 0 && (KP % 256) == 0 && (M % 32) == 0 && M >= 2048; }
static void tn256_plan(int M, int NP, int KP, int* splits, int* mps) {
  const int tiles = (NP / 256) * (KP / 256);
  int sp = 256 / tiles;                               // one workgroup per CU and ONE round: never more than 256 workgroups
  if (sp > M / 128) sp = M / 128;                     // at least 4 K-steps per split
  if (sp < 1) sp = 1;
  int m = ((M + sp - 1) / sp + 31) / 32 * 32;
  *mps = m;
  *splits = (M + m - 1) / m;
}

// splits the dispatch below will use for (M, NP, KP) (exported for the sizing test: the workspace must cover every row count
// <= the M it was sized for, because variable-length batches call ltrx_gemm_tn with a different M every step)
extern "C" int ltrx_gemm_tn_splits(int M, int NP, int KP) {
  if (M <= 0 || NP <= 0 || KP <= 0) return 0;
  if (tn256_ok(M, NP, KP)) {
    int s2, mps;
    tn256_plan(M, NP, KP, &s2, &mps);
    return s2;
  }
  return tn_splits(M, ((NP + 127) / 128) * ((KP + BN - 1) / BN));
}

// Upper bound over ALL row counts m <= M (not only M itself): the small-tile plan is monotone in m (tn_splits), the
// large-tile plan never uses more than min(256 / tiles256, m / 128) splits but is not monotone (m is rounded to 32-row
// slabs and the kernel switches on at m >= 2048, m % 32 == 0), so its bound is taken whenever the shape could select it.
extern "C" size_t ltrx_gemm_tn_workspace_bytes(int M, int NP, int KP) {
  if (M <= 0 || NP <= 0 || KP <= 0) return 0;
  const int tiles = ((NP + 127) / 128) * ((KP + BN - 1) / BN);
  size_t sp = (size_t)tn_splits(M, tiles);
  if ((NP % 256) == 0 && M >= 2048) {                 // (KP % 256 != 0: the large tile over a padded B, when ldb allows it)
    int s2 = 256 / ((NP / 256) * ((KP + 255) / 256));
    if (s2 > M / 128) s2 = M / 128;
    if (s2 < 1) s2 = 1;
    if ((size_t)s2 > sp) sp = (size_t)s2;
  }
  return (sp * NP * KP + 2 * sp * NP) * sizeof(float);
}

extern "C" int ltrx_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, float* bias_out, int M, int NP,
                            int KP, int strict, int tile, void* ws, ltrx_stream_t stream) {
  if (!A || !B || !C || !ws || M <= 0 || NP <= 0 || KP <= 0 || tile < 0) return LTRX_EINVAL;
  if (lda < NP || ldb < KP) return LTRX_EUNSUPPORTED;
  const bool plain = strict == 2;                     // precision code as in ltrx_gemm_nt
  if (strict == 2) strict = 0;
  // the large tile also serves a B whose column count is not a multiple of 256 when its ROW STRIDE covers the last tile (the
  // engine pads the input features to 256 floats per row): the surplus columns are computed from the padding and dropped
  // (opt-in, tile = 9: the kernel reads B[m][KP .. KPr) -- the caller states that every row, the last one included, is readable there)
  const int KPr = (KP + 255) / 256 * 256;
  if (tile == 9 && (KPr == KP || ldb < KPr)) tile = 0;
  if (!strict && tile != 1 && (KPr == KP || tile == 9) && tn256_ok(M, NP, KPr) && ldb >= KPr && (lda & 3) == 0 && (ldb & 3) == 0) {
    const int kv = KP;
    KP = KPr;
    hipStream_t s = (hipStream_t)stream;
    int splits, mps;
    tn256_plan(M, NP, KP, &splits, &mps);
    static std::atomic<uint64_t> attr_done{0};
    const int arc = ltrx_once_per_device(attr_done, []() {
      if (hipFuncSetAttribute((const void*)ltrx_gemm_tn256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(2 * sizeof(SmemNT<256, 2>))) != hipSuccess ||
          hipFuncSetAttribute((const void*)ltrx_gemm_tn256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(2 * sizeof(SmemNT<256, 1>))) != hipSuccess)
        return LTRX_EHIP;
      return LTRX_OK;
    });
    if (arc != LTRX_OK) return arc;
    float* bslabs = bias_out ? (float*)ws + (((size_t)splits * NP * kv + 3) & ~(size_t)3) : nullptr;
    TnGroup g = {};
    g.kv[0] = kv;
    g.A[0] = A;
    g.B[0] = B;
    g.slabs[0] = (float*)ws;
    g.bias_slabs[0] = bslabs;
    g.lda[0] = lda;
    g.ldb[0] = ldb;
    g.NP[0] = NP;
    g.KP[0] = KP;
    g.tiles_k[0] = KP / 256;
    g.tile_start[0] = 0;
    g.tile_start[1] = (NP / 256) * (KP / 256);
    g.nprob = 1;
    if (plain)
      hipLaunchKernelGGL(ltrx_gemm_tn256_kernel<1>, dim3((NP / 256) * (KP / 256), splits), dim3(512), 2 * sizeof(SmemNT<256, 1>), s, g, M, mps);
    else
      hipLaunchKernelGGL(ltrx_gemm_tn256_kernel<2>, dim3((NP / 256) * (KP / 256), splits), dim3(512), 2 * sizeof(SmemNT<256, 2>), s, g, M, mps);
    LTRX_LAUNCH_CHECK();
    launch_slab_reduce((const float*)ws, splits, (size_t)NP * kv, C, bslabs, splits, (size_t)NP, bias_out, s);
    LTRX_LAUNCH_CHECK();
    return LTRX_OK;
  }
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
  else if (plain)
    hipLaunchKernelGGL(ltrx_gemm_tn_kernel<1>, grid, dim3(256), 0, s, A, lda, B, ldb, (float*)ws, bslabs, M, NP, KP, tiles_k, mps);
  else
    hipLaunchKernelGGL(ltrx_gemm_tn_kernel<2>, grid, dim3(256), 0, s, A, lda, B, ldb, (float*)ws, bslabs, M, NP, KP, tiles_k, mps);
  LTRX_LAUNCH_CHECK();
  launch_slab_reduce((const float*)ws, splits, (size_t)NP * KP, C, bslabs, 2 * splits, (size_t)NP, bias_out, s);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// ---- grouped weight gradients (see TnGroup) ----
static void tn_group_plan(int nprob, int M, const int* NP, const int* KP, int* total_tiles, int* splits, int* mps) {
  int t = 0;
  for (int p = 0; p < nprob; ++p) t += (NP[p] / 256) * (KP[p] / 256);
  int sp = t > 0 ? 256 / t : 1;                       // one workgroup per CU and ONE round
  if (sp > M / 128) sp = M / 128;                     // at least 4 K-steps per split
  if (sp < 1) sp = 1;
  const int m = ((M + sp - 1) / sp + 31) / 32 * 32;
  *total_tiles = t;
  *mps = m;
  *splits = (M + m - 1) / m;
}
static bool tn_group_ok(int nprob, int M, const int* NP, const int* KP, const int* lda, const int* ldb, int strict) {
  if (nprob < 1 || nprob > LTRX_TN_GROUP || strict == 1) return false;
  int t = 0;
  for (int p = 0; p < nprob; ++p) {
    if (!tn256_ok(M, NP[p], KP[p]) || (lda && ((lda[p] & 3) || lda[p] < NP[p])) || (ldb && ((ldb[p] & 3) || ldb[p] < KP[p]))) return false;
    t += (NP[p] / 256) * (KP[p] / 256);
  }
  return t <= 256;
}

// (problem, split) groups -> XCDs: fills g.map_tile / g.map_split for the total * splits (<= 256) workgroups of a grouped launch
static void tn_group_map(TnGroup& g, int total, int splits) {
  const int nprob = g.nprob;
  // (problem, split) groups, largest first, into the 8 XCDs (first fit, capacity = an even share of the grid; a group that fits
  // nowhere whole is cut); XCD x owns the workgroup ids x, x + 8, x + 16, ...
  const int nwg = total * splits;                        // <= 256 (tn_group_plan)
  const int cap = (nwg + 7) / 8;
  int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int order[LTRX_TN_GROUP];
  for (int p = 0; p < nprob; ++p) order[p] = p;
  for (int a = 0; a < nprob; ++a)
    for (int b = a + 1; b < nprob; ++b)
      if (g.tile_start[order[b] + 1] - g.tile_start[order[b]] > g.tile_start[order[a] + 1] - g.tile_start[order[a]]) {
        const int t_ = order[a];
        order[a] = order[b];
        order[b] = t_;
      }
  auto slot_count = [&](int x) { return (nwg - x + 7) / 8; };      // ids x, x+8, ... < nwg
  for (int oi = 0; oi < nprob; ++oi) {
    const int p = order[oi], nt = g.tile_start[p + 1] - g.tile_start[p];
    for (int sp_ = 0; sp_ < splits; ++sp_) {
      int left = nt, next = 0;
      while (left > 0) {
        int best = -1;
        for (int x = 0; x < 8; ++x)                      // an XCD that takes the rest whole, else the emptiest one
          if (fill[x] + left <= (cap < slot_count(x) ? cap : slot_count(x))) {
            best = x;
            break;
          }
        if (best < 0) {
          int room = -1;
          for (int x = 0; x < 8; ++x) {
            const int r_ = slot_count(x) - fill[x];
            if (r_ > room) {
              room = r_;
              best = x;
            }
          }
        }
        int take = slot_count(best) - fill[best];
        if (take > left) take = left;
        for (int q = 0; q < take; ++q) {
          const int w = best + 8 * (fill[best] + q);
          g.map_tile[w] = (unsigned char)(g.tile_start[p] + next + q);
          g.map_split[w] = (unsigned char)sp_;
        }
        fill[best] += take;
        next += take;
        left -= take;
      }
    }
  }
  }

// bytes for ltrx_gemm_tn_group: an upper bound over every row count m <= M (variable-length batches): per problem at most
// min(256 / total_tiles, m / 128) + 1 slabs of NP x KP (+ NP for the bias), and never less than the single-problem calls need (the
// group call falls back to them when a shape does not qualify)
extern "C" size_t ltrx_gemm_tn_group_workspace_bytes(int nprob, int M, const int* NP, const int* KP) {
  if (nprob < 1 || nprob > LTRX_TN_GROUP || M <= 0 || !NP || !KP) return 0;
  size_t single = 0;
  int t = 0;
  for (int p = 0; p < nprob; ++p) {
    const size_t b = ltrx_gemm_tn_workspace_bytes(M, NP[p], KP[p]);
    if (b > single) single = b;
    t += ((NP[p] + 255) / 256) * ((KP[p] + 255) / 256);
  }
  size_t sp = t > 0 ? (size_t)(256 / t) : 1;
  if (sp < 1) sp = 1;
  sp += 1;
  size_t grouped = 0;
  for (int p = 0; p < nprob; ++p) grouped += (sp * NP[p] * KP[p] + sp * NP[p] + 4) * sizeof(float);
  return grouped > single ? grouped : single;
}

extern "C" int ltrx_gemm_tn_group_img(int nprob, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                                      float* const* bias_out, int M, const int* NP, const int* KP, int strict, void* ws, size_t ws_bytes,
                                      const float** slabs_out, const float** bias_slabs_out, int* splits_out, const int* b_is_image,
                                      ltrx_stream_t stream);

extern "C" int ltrx_gemm_tn_group(int nprob, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                                  float* const* bias_out, int M, const int* NP, const int* KP, int strict, void* ws, size_t ws_bytes,
                                  const float** slabs_out, const float** bias_slabs_out, int* splits_out, ltrx_stream_t stream) {
  return ltrx_gemm_tn_group_img(nprob, A, lda, B, ldb, C, bias_out, M, NP, KP, strict, ws, ws_bytes, slabs_out, bias_slabs_out, splits_out,
                                nullptr, stream);
}

extern "C" int ltrx_gemm_tn_group_img(int nprob, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                                      float* const* bias_out, int M, const int* NP, const int* KP, int strict, void* ws, size_t ws_bytes,
                                      const float** slabs_out, const float** bias_slabs_out, int* splits_out, const int* b_is_image,
                                      ltrx_stream_t stream) {
  const bool defer = slabs_out && bias_slabs_out && splits_out;   // the caller sums the slabs (ltrx_reduce_group)
  if ((slabs_out || bias_slabs_out || splits_out) && !defer) return LTRX_EINVAL;
  if (defer) *splits_out = 0;
  if (nprob < 1 || nprob > LTRX_TN_GROUP || !A || !lda || !B || !ldb || !C || !bias_out || !NP || !KP || !ws || M <= 0) return LTRX_EINVAL;
  for (int p = 0; p < nprob; ++p)
    if (!A[p] || !B[p] || !C[p] || NP[p] <= 0 || KP[p] <= 0) return LTRX_EINVAL;
  const bool plain = strict == 2;
  int total = 0, splits = 1, mps = M;
  bool grouped = tn_group_ok(nprob, M, NP, KP, lda, ldb, strict);
  size_t need = 0;
  if (grouped) {
    tn_group_plan(nprob, M, NP, KP, &total, &splits, &mps);
    for (int p = 0; p < nprob; ++p) need += ((size_t)splits * NP[p] * KP[p] + (size_t)splits * NP[p] + 4) * sizeof(float);
    if (need > ws_bytes) grouped = false;
  }
  if (!grouped) {                                     // shapes outside the large-tile kernel: one call per problem
    for (int p = 0; p < nprob; ++p)
      if (b_is_image && b_is_image[p]) return LTRX_EUNSUPPORTED;       // (operand images exist in the grouped large-tile kernel only)
    for (int p = 0; p < nprob; ++p) {
      if (ltrx_gemm_tn_workspace_bytes(M, NP[p], KP[p]) > ws_bytes) return LTRX_EINVAL;
      const int rc = ltrx_gemm_tn(A[p], lda[p], B[p], ldb[p], C[p], bias_out[p], M, NP[p], KP[p], strict, 0, ws, stream);
      if (rc != LTRX_OK) return rc;
    }
    return LTRX_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  static std::atomic<uint64_t> attr_done{0};
  const int arc = ltrx_once_per_device(attr_done, []() {
    if (hipFuncSetAttribute((const void*)ltrx_gemm_tn256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(2 * sizeof(SmemNT<256, 2>))) != hipSuccess ||
        hipFuncSetAttribute((const void*)ltrx_gemm_tn256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(2 * sizeof(SmemNT<256, 1>))) != hipSuccess)
      return LTRX_EHIP;
    return LTRX_OK;
  });
  if (arc != LTRX_OK) return arc;
  TnGroup g = {};
  g.nprob = nprob;
  float* w = (float*)ws;
  int t0 = 0;
  for (int p = 0; p < nprob; ++p) {
    g.A[p] = A[p];
    g.B[p] = B[p];
    g.lda[p] = lda[p];
    g.ldb[p] = ldb[p];
    g.NP[p] = NP[p];
    g.KP[p] = KP[p];
    g.kv[p] = KP[p];
    g.bimg[p] = (b_is_image && b_is_image[p]) ? 1 : 0;
    if (g.bimg[p] && (((uintptr_t)B[p]) & 15)) return LTRX_EUNSUPPORTED;
    g.tiles_k[p] = KP[p] / 256;
    g.tile_start[p] = t0;
    t0 += (NP[p] / 256) * (KP[p] / 256);
    g.slabs[p] = w;
    w += (size_t)splits * NP[p] * KP[p];
    g.bias_slabs[p] = bias_out[p] ? w : nullptr;
    w += ((size_t)splits * NP[p] + 3) & ~(size_t)3;
  }
  g.tile_start[nprob] = t0;
  tn_group_map(g, total, splits);
  if (plain)
    hipLaunchKernelGGL(ltrx_gemm_tn256_kernel<1>, dim3(total, splits), dim3(512), 2 * sizeof(SmemNT<256, 1>), s, g, M, mps);
  else
    hipLaunchKernelGGL(ltrx_gemm_tn256_kernel<2>, dim3(total, splits), dim3(512), 2 * sizeof(SmemNT<256, 2>), s, g, M, mps);
  LTRX_LAUNCH_CHECK();
  if (defer) {
    for (int p = 0; p < nprob; ++p) {
      slabs_out[p] = g.slabs[p];
      bias_slabs_out[p] = g.bias_slabs[p];
    }
    *splits_out = splits;
    return LTRX_OK;
  }
  for (int p = 0; p < nprob; ++p) {
    launch_slab_reduce(g.slabs[p], splits, (size_t)NP[p] * KP[p], C[p], g.bias_slabs[p], splits, (size_t)NP[p], bias_out[p], s);
    LTRX_LAUNCH_CHECK();
  }
  return LTRX_OK;
}

// test hook (no GPU needed): the workgroup -> (tile, split) table ltrx_gemm_tn_group would launch with for these shapes; returns the
// workgroup count (0 when the shapes take the per-problem path)
extern "C" int ltrx_debug_tn_group_map(int nprob, int M, const int* NP, const int* KP, unsigned char* tile_out, unsigned char* split_out) {
  if (nprob < 1 || nprob > LTRX_TN_GROUP || !NP || !KP || !tile_out || !split_out || M <= 0) return 0;
  if (!tn_group_ok(nprob, M, NP, KP, nullptr, nullptr, 0)) return 0;
  int total = 0, splits = 1, mps = M;
  tn_group_plan(nprob, M, NP, KP, &total, &splits, &mps);
  TnGroup g = {};
  g.nprob = nprob;
  int t0 = 0;
  for (int p = 0; p < nprob; ++p) {
    g.tile_start[p] = t0;
    t0 += (NP[p] / 256) * (KP[p] / 256);
  }
  g.tile_start[nprob] = t0;
  tn_group_map(g, total, splits);
  for (int w = 0; w < total * splits; ++w) {
    tile_out[w] = g.map_tile[w];
    split_out[w] = g.map_split[w];
  }
  return total * splits;
}
