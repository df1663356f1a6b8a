// fp32-accurate GEMMs on the bf16 matrix cores whose operands ARRIVE PRE-SPLIT ("HL16 images") and are staged by LDS-DMA.
// Same arithmetic as ltrx_gemm.hip (A B^T ~= Ah Bh^T + Ah Bl^T + Al Bh^T, bf16 MFMA, fp32 accumulate) for the dense
// projections of the encoder (allrank/models/transformer.py:193-203 q/k/v/out, :221-227 feed-forward) and their
// input / weight gradients; what changes is WHERE the fp32 -> bf16 hi/lo split happens: in the epilogue of the kernel that
// PRODUCES a tensor (LayerNorm, GEMM, attention, the post-Adam weight refresh), once, instead of in the K-loop of every
// GEMM that consumes it.  The K-loop then has no VALU split, no VGPR staging and no ds_write: global_load -> LDS directly.
//
// HL16 image of an fp32 matrix X[R][K] (K % 16 == 0), same bytes and row pitch (4 K) as X:  row r, 16-column block b:
//     bytes [64 b, 64 b + 32)      bf16 hi of X[r][16 b .. 16 b + 15]          hi = bf16(x)
//     bytes [64 b + 32, 64 b + 64) bf16 lo of the same 16 elements             lo = bf16(x - hi)     |x - hi - lo| <= 2^-18 |x|
//
// Kernel structure (measured in tools/lab/gemm_img.hip): 256 x 256 output tile, 8 waves (wave tile
// 128 x 64, 128 accumulator VGPRs), a RING OF FOUR 16-deep sub-stages in LDS (4 x 32 KB), and ROLE PING-PONG: the two
// waves of every SIMD (wave groups 0-3 / 4-7) run the same loop ONE PHASE APART (group 1 takes an extra barrier up
// front); a sub-step is two barrier-separated phases
//        R: 4 LDS-DMA instructions (sub-stage s+2), 12 fragment ds_reads (sub-stage s), counted wait vmcnt(4)
//        M: 24 MFMAs (v_mfma_f32_32x32x16_bf16), s_setprio 1
//   phase    2s      2s+1    2s+2     2s+3
//   group 0  R(s)    M(s)    R(s+1)   M(s+1)
//   group 1  M(s-1)  R(s)    M(s)     R(s+1)
// so on each SIMD one wave is always inside an MFMA block whose operands are already in registers while its partner
// fetches.  DMA(s+2) goes to ring slot (s+2) & 3, last read (as sub-stage s-2) three phases earlier; every wave waits for
// its own DMA(s+1) in R(s), and the barrier that ends phase 2s+1 publishes it before anyone reads it in phase 2s+2.
// The MFMA operands are swapped (the "B" image as the MFMA A operand): a lane then owns ONE output row and 4 consecutive
// columns per register group, one v_permlane32_swap per register makes that 8 consecutive columns -> two 16-byte stores
// (fp32) or one 16-byte hi + one 16-byte lo store (image output), no transposes.  All global accesses of the tile go through
// buffer descriptors sized to the real row count: rows beyond M read as zeros and their stores are dropped by the bounds
// check -- no tail code, no branches around MFMA blocks (a branch around an MFMA block makes this compiler spill ~250 VGPRs).
#include "ltrx_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) * lds_ptr_t;

namespace {

constexpr unsigned BUF_FLAGS = 0x00020000u;      // raw buffer, 32-bit data format (gfx9 word 3)

__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// byte offset of element column `col` (hi plane) inside an HL16 row; the lo plane is 32 bytes further
__device__ __forceinline__ int hl16_off(int col) { return (col >> 4) * 64 + (col & 15) * 2; }

__device__ __forceinline__ void split8(const float (&z)[8], bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (__bf16)z[e];
    l[e] = (__bf16)(z[e] - (float)h[e]);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// fp32 [rows][K] (leading dimension ldx) -> HL16 image (pitch ldi floats-worth of bytes); one thread per 8 columns
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ltrx_to_image_kernel(const float* __restrict__ X, int ldx, size_t rows, int K,
                                                            unsigned char* __restrict__ img, int ldi) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int k8 = K / 8;
  if (i >= rows * (size_t)k8) return;
  const size_t r = i / k8;
  const int k = (int)(i % k8) * 8;
  const f32x4 a = *reinterpret_cast<const f32x4*>(X + r * ldx + k), b = *reinterpret_cast<const f32x4*>(X + r * ldx + k + 4);
  const float z[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  bf16x8 h, l;
  split8(z, h, l);
  unsigned char* p = img + r * (size_t)ldi * 4 + hl16_off(k);
  *reinterpret_cast<bf16x8*>(p) = h;
  *reinterpret_cast<bf16x8*>(p + 32) = l;
}

// batch form for the weights (after every optimizer step): descriptor d = {src offset, dst offset (both in floats), rows, K};
// unit_start[d] = first 8-column unit of descriptor d in the flat unit numbering
__global__ void __launch_bounds__(256) ltrx_to_image_batch_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst,
                                                                  const long long* __restrict__ desc,
                                                                  const int* __restrict__ unit_start, int n, int total_units) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= total_units) return;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (unit_start[mid] <= u) lo = mid; else hi = mid - 1;
  }
  const long long so = desc[4 * lo + 0], dof = desc[4 * lo + 1];
  const int K = (int)desc[4 * lo + 3], k8 = K / 8;
  const int v = u - unit_start[lo];
  const int r = v / k8, k = (v % k8) * 8;
  const float* x = src + so + (size_t)r * K + k;
  const f32x4 a = *reinterpret_cast<const f32x4*>(x), b = *reinterpret_cast<const f32x4*>(x + 4);
  const float z[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  bf16x8 h, l;
  split8(z, h, l);
  unsigned char* p = dst + ((size_t)dof + (size_t)r * K) * 4 + hl16_off(k);
  *reinterpret_cast<bf16x8*>(p) = h;
  *reinterpret_cast<bf16x8*>(p + 32) = l;
}

// ------------------------------------------------------------------------------------------------------------------
// NT:  C[M,N] = epi( A[M,K] B[N,K]^T ),  A and B as HL16 images
// LDS sub-stage (ring slot): [operand a|b][256 rows][4 x 16 B]; position j of row r holds logical piece j ^ ((r >> 2) & 3),
// piece = plane * 2 + (k >> 3 & 1)  -- conflict-free for the 16-lane groups of ds_read_b128, and the XOR only involves lane
// bits of the DMA instruction that fills it (the DMA destination is lane-linear: the permutation goes on the SOURCE).
// ACT: 0 none, 1 ReLU, 2 multiply by (aux > 0) * inv_keep (aux = HL16 image of the saved post-activation tensor)
// ------------------------------------------------------------------------------------------------------------------
template <int ACT, bool OUT_IMG>
__global__ void __launch_bounds__(512) ltrx_gemm_nt_img_kernel(const unsigned char* __restrict__ A, int lda,
                                                               const unsigned char* __restrict__ B, int ldb, void* __restrict__ Cv,
                                                               int ldc, int M, int N, int K, const float* __restrict__ bias,
                                                               const unsigned char* __restrict__ aux, int ldaux, int tiles_n,
                                                               ltrx::DropSpec drop, const uint32_t* __restrict__ drop_step) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (id / tiles_n) * 256, n0 = (id % tiles_n) * 256;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = wave >> 2, wc = wave & 3;
  const int grp = wave >> 2;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * lda * 4), BUF_FLAGS);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)N * ldb * 4), BUF_FLAGS);
  // DMA: instruction p (0..1) of wave w moves rows 32 w + 16 p + (lane >> 2); LDS position = lane & 3
  const int jp = (lane & 3) ^ ((lane >> 4) & 3);                 // logical piece held at this lane's LDS position
  const int src_off = (jp >> 1) * 32 + (jp & 1) * 16;
  const int va = (lane >> 2) * lda * 4 + src_off, vb = (lane >> 2) * ldb * 4 + src_off;
  const int sa0 = (m0 + 32 * wave) * lda * 4, sb0 = (n0 + 32 * wave) * ldb * 4;
  auto dma4 = [&](int sub) {                              // the 4 DMA instructions of sub-stage `sub`
    unsigned char* base = smem_raw + (sub & 3) * 32768 + (32 * wave) * 64;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(base + p * 1024), 16, va, sa0 + 16 * p * lda * 4 + sub * 64, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(base + 16384 + p * 1024), 16, vb, sb0 + 16 * p * ldb * 4 + sub * 64, 0, 0);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
  int fo[2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) fo[pl] = l31 * 64 + ((((pl << 1) | half) ^ ((l31 >> 2) & 3)) << 4);
  const int fa = wr * 128 * 64, fb = 16384 + wc * 64 * 64;
  bf16x8 af[2][4], bfr[2][2];
  const int S = K / 16;
  dma4(0);
  if (S > 1) dma4(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one phase behind
  for (int sub = 0; sub < S; ++sub) {
    // ---- R(sub)
    if (sub + 2 < S) dma4(sub + 2);
    const unsigned char* sl = smem_raw + (sub & 3) * 32768;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) bfr[tt][jj] = *reinterpret_cast<const bf16x8*>(sl + fb + jj * 2048 + fo[tt]);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[tt][i] = *reinterpret_cast<const bf16x8*>(sl + fa + i * 2048 + fo[tt]);
    }
    if (sub + 2 < S) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- M(sub): D[n][m] += W[n][k] X[m][k]  (the B image as the MFMA "A" operand)
    __builtin_amdgcn_s_setprio(1);
#define LTRX_MMA(TA, TB)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)  \
      acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[TB][jj], af[TA][i], acc[i][jj], 0, 0, 0);
    LTRX_MMA(0, 1)
    LTRX_MMA(1, 0)
    LTRX_MMA(0, 0)
#undef LTRX_MMA
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();            // both groups execute the same number of barriers

  // ---- epilogue.  acc[i][jj][4g + e] = C[m0 + wr*128 + i*32 + l31][n0 + wc*64 + jj*32 + 8g + 4*half + e]
  ltrx::DropSpec dsp = drop;
  if (drop_step) dsp.seed ^= drop_step[0] * 0x9E3779B9u;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (unsigned)((size_t)M * ldc * 4), BUF_FLAGS);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)aux, 0, (unsigned)((size_t)M * ldaux * 4), BUF_FLAGS);
  f32x4 bv[2][2][2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int col = n0 + wc * 64 + jj * 32 + 16 * gp + 8 * half;
      if (bias) {
        bv[jj][gp][0] = *reinterpret_cast<const f32x4*>(bias + col);
        bv[jj][gp][1] = *reinterpret_cast<const f32x4*>(bias + col + 4);
      } else {
        bv[jj][gp][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        bv[jj][gp][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wr * 128 + i * 32 + l31;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int col = n0 + wc * 64 + jj * 32 + 16 * gp + 8 * half;
        float z[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // v_permlane32_swap(x, y): x' = [x.low half | y.low half], y' = [x.high half | y.high half]
          //   low lanes : x' = own group 2gp (cols +0..3),            y' = high lanes' group 2gp (cols +4..7)
          //   high lanes: x' = low lanes' group 2gp+1 (cols +8..11),  y' = own group 2gp+1 (cols +12..15)
          // (vector elements are copied to scalars first: __builtin_bit_cast applied directly to an ext_vector element
          //  subscript reads element 0 with this compiler)
          const float xe = acc[i][jj][8 * gp + e], ye = acc[i][jj][8 * gp + 4 + e];
          const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xe), __builtin_bit_cast(unsigned, ye), false, false);
          const unsigned s0 = sw[0], s1 = sw[1];
          z[e] = __builtin_bit_cast(float, s0);
          z[4 + e] = __builtin_bit_cast(float, s1);
        }
        const f32x4 b0 = bv[jj][gp][0], b1 = bv[jj][gp][1];
        z[0] += b0.x; z[1] += b0.y; z[2] += b0.z; z[3] += b0.w;
        z[4] += b1.x; z[5] += b1.y; z[6] += b1.z; z[7] += b1.w;
        if (ACT == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) z[e] = fmaxf(z[e], 0.f);
        }
        if (ACT == 2) {          // ReLU (+dropout) backward: the mask is carried by the hi plane of the saved activation
          const u32x4 ax = __builtin_amdgcn_raw_buffer_load_b128(rx, row * ldaux * 4 + hl16_off(col), 0, 0);
          const bf16x8 ah = __builtin_bit_cast(bf16x8, ax);
#pragma unroll
          for (int e = 0; e < 8; ++e) z[e] = ((float)ah[e] > 0.f) ? z[e] * drop.inv_keep : 0.f;
        } else if (drop.thresh != 0u) {
          const uint64_t e0 = (uint64_t)row * (uint64_t)N + (uint64_t)col;
#pragma unroll
          for (int e = 0; e < 8; ++e) z[e] *= ltrx::drop_keep_scale(dsp, e0 + e);
        }
        if (OUT_IMG) {
          bf16x8 h, l;
          split8(z, h, l);
          const int off = row * ldc * 4 + hl16_off(col);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rc, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, l), rc, off + 32, 0, 0);
        } else {
          const int off = (row * ldc + col) * 4;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4{z[0], z[1], z[2], z[3]})), rc, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4{z[4], z[5], z[6], z[7]})), rc, off + 16, 0, 0);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// TN (weight gradient):  C[NP,KP] = sum_m A[m][n'] B[m][k'],  A = dY, B = X as HL16 images [M][*]; split over m into slabs.
// Both operands are contraction-STRIDED; their tiles are DMA'd as they lie ([16 m][256 cols], one 1-KB row per DMA
// instruction) and the MFMA fragments are gathered with ds_read_b64_tr_b16 (each 16-lane group reads a [4 m][16 col]
// block and gets it column-major: lane = column, 4 consecutive m).  LDS row m (local, 0..15) stores its 16-byte chunk c at
// position c ^ ((m & 1) * 2 + ((m >> 1) & 1) * 8): the four rows of a transposed read then sit in four different 32-byte
// bank groups and the two 16-lane groups of a half-wave in different 64-byte halves (conflict-free).
// Bias gradient (column sums of dY) = dY^T * 1: two extra MFMAs per sub-step against an all-ones operand in the workgroups
// of the first tile column, split over the waves (wave (wr, wc) sums the 32 columns wr*128 + wc*32 ..).
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) ltrx_gemm_tn_img_kernel(const unsigned char* __restrict__ A, int lda,
                                                               const unsigned char* __restrict__ B, int ldb, float* __restrict__ slabs,
                                                               float* __restrict__ bias_slabs, int M, int NP, int KP, int tiles_k,
                                                               int m_per_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_tiles = gridDim.x;
  const int wg = xcd_remap(blockIdx.x + n_tiles * blockIdx.y, n_tiles * gridDim.y);
  const int tile = wg % n_tiles, split = wg / n_tiles;
  const int n0 = (tile / tiles_k) * 256, k0 = (tile % tiles_k) * 256;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = wave >> 2, wc = wave & 3;
  const int grp = wave >> 2;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31, i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int mbeg = split * m_per_split, mend = min(M, mbeg + m_per_split);
  const bool want_bias = bias_slabs != nullptr && (tile % tiles_k) == 0;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * lda * 4), BUF_FLAGS);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)M * ldb * 4), BUF_FLAGS);
  // DMA: wave w moves local rows m = 2 w and 2 w + 1 of both operands (one 1-KB row per instruction); LDS position = lane
  int vo[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int m = 2 * wave + e;
    vo[e] = (lane ^ ((m & 1) * 2 + ((m >> 1) & 1) * 8)) * 16;
  }
  const int sa0 = n0 * 4, sb0 = k0 * 4;
  auto dma4 = [&](int sub) {
    unsigned char* base = smem_raw + (sub & 3) * 32768 + (2 * wave) * 1024;
    const int mrow = mbeg + sub * 16 + 2 * wave;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(base + e * 1024), 16, vo[e], (mrow + e) * lda * 4 + sa0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(base + 16384 + e * 1024), 16, vo[e], (mrow + e) * ldb * 4 + sb0, 0, 0);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
  f32x16 accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  // transposed fragment reads: lane (i16, g16, half) reads local row m = 8 half + 4 q + (i16 >> 2), q = 0, 1, the 8 bytes
  // (i16 & 3) * 8 of the 32-byte plane segment of 16-column block blk (+ g16): chunk = 4 blk + 2 plane + ((i16 & 3) >> 1)
  const int mloc = 8 * half + (i16 >> 2);
  const int kx = ((mloc & 1) * 2) + (((mloc >> 1) & 1) * 8);                 // (4 q does not change m & 3)
  const int lane_base = mloc * 1024 + (i16 & 1) * 8;
  const int hb = (i16 & 3) >> 1;
  const int blk_a = wr * 8 + g16, blk_b = wc * 4 + g16;                       // + 2 i  /  + 2 jj
  bf16x8 af[2][4], bfr[2][2];
  const int S = (mend - mbeg) / 16;
  if (S > 0) {
    dma4(0);
    if (S > 1) dma4(1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();
  for (int sub = 0; sub < S; ++sub) {
    if (sub + 2 < S) dma4(sub + 2);
    const unsigned char* sl = smem_raw + (sub & 3) * 32768 + lane_base;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int off = 16384 + ((((blk_b + 2 * jj) * 4 + 2 * tt + hb) ^ kx) << 4);
        const bf16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(sl + off));
        const bf16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(sl + off + 4096));
        bfr[tt][jj] = bf16x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = ((((blk_a + 2 * i) * 4 + 2 * tt + hb) ^ kx) << 4);
        const bf16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(sl + off));
        const bf16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(sl + off + 4096));
        af[tt][i] = bf16x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      }
    }
    if (sub + 2 < S) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#define LTRX_MMA(TA, TB)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)  \
      acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[TB][jj], af[TA][i], acc[i][jj], 0, 0, 0);
    LTRX_MMA(0, 1)
    LTRX_MMA(1, 0)
    LTRX_MMA(0, 0)
#undef LTRX_MMA
    if (want_bias) {           // D[n'][*] += dY^T[n'][m] * 1   (block i = wc of this wave's 128 rows; lo first)
      bf16x8 fl, fh;
      fl = wc == 0 ? af[1][0] : wc == 1 ? af[1][1] : wc == 2 ? af[1][2] : af[1][3];
      fh = wc == 0 ? af[0][0] : wc == 1 ? af[0][1] : wc == 2 ? af[0][2] : af[0][3];
      accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, ones, accb, 0, 0, 0);
      accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, ones, accb, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  // slab[n'][k']: lane owns row n' = n0 + wr*128 + i*32 + l31, register group g holds k' = k0 + wc*64 + jj*32 + 8g + 4 half ..+3
  float* slab = slabs + (size_t)split * NP * KP;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* rowp = slab + (size_t)(n0 + wr * 128 + i * 32 + l31) * KP + k0 + wc * 64 + 4 * half;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(rowp + jj * 32 + 8 * g) =
            f32x4{acc[i][jj][4 * g + 0], acc[i][jj][4 * g + 1], acc[i][jj][4 * g + 2], acc[i][jj][4 * g + 3]};
  }
  if (want_bias && l31 == 0) {        // accb[r] = sum over m of dY[m][n0 + wr*128 + wc*32 + rowmap(r, half)] (every column)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bias_slabs[(size_t)split * NP + n0 + wr * 128 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = accb[r];
  }
}

// C[i] = sum_s slabs[s][i] in a fixed order; the second job (bias slabs) rides in the same launch
__global__ void __launch_bounds__(256) ltrx_gemm_img_slab_reduce_kernel(const float* __restrict__ slabs, int splits, size_t n,
                                                                        float* __restrict__ C, int main_blocks,
                                                                        const float* __restrict__ slabs2, size_t n2,
                                                                        float* __restrict__ C2) {
  const bool second = (int)blockIdx.x >= main_blocks;
  const float* s = second ? slabs2 : slabs;
  const size_t cnt = second ? n2 : n;
  float* out = second ? C2 : C;
  const size_t b = second ? blockIdx.x - main_blocks : blockIdx.x;
  const size_t stride = (size_t)(second ? gridDim.x - main_blocks : main_blocks) * blockDim.x;
  for (size_t i = b * blockDim.x + threadIdx.x; i < cnt; i += stride) {
    float a = 0.f;
    for (int k = 0; k < splits; ++k) a += s[(size_t)k * cnt + i];
    out[i] = a;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
extern "C" int ltrx_to_image(const float* X, int ldx, int rows, int K, void* image, int ld_image, ltrx_stream_t stream) {
  if (!X || !image || rows <= 0 || K <= 0) return LTRX_EINVAL;
  if ((K & 15) || (ldx & 3) || ldx < K || ld_image < K || ((uintptr_t)X & 15) || ((uintptr_t)image & 15)) return LTRX_EUNSUPPORTED;
  const size_t units = (size_t)rows * (K / 8);
  hipLaunchKernelGGL(ltrx_to_image_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, ldx,
                     (size_t)rows, K, (unsigned char*)image, ld_image);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

extern "C" int ltrx_to_image_batch(const float* src, void* dst, const long long* desc, const int* unit_start, int n,
                                   int total_units, ltrx_stream_t stream) {
  if (!src || !dst || !desc || !unit_start || n <= 0 || total_units <= 0) return LTRX_EINVAL;
  hipLaunchKernelGGL(ltrx_to_image_batch_kernel, dim3((unsigned)((total_units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (unsigned char*)dst, desc, unit_start, n, total_units);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

static bool fits_u32(size_t rows, int ld) { return rows * (size_t)ld * 4 < 0xFFFFFFFFull; }

extern "C" int ltrx_gemm_nt_img(const void* A_image, int lda, const void* B_image, int ldb, void* C, int ldc, int c_is_image,
                                int M, int N, int K, const float* bias, int act, const void* aux_image, int ldaux, float drop_p,
                                uint32_t drop_seed, const uint32_t* drop_step, ltrx_stream_t stream) {
  if (!A_image || !B_image || !C || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return LTRX_EINVAL;
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return LTRX_EINVAL;
  if (act == 2 && (!aux_image || ldaux < N)) return LTRX_EINVAL;
  if ((N % 256) || (K % 16) || lda < K || ldb < K || ldc < N || (lda & 3) || (ldb & 3) || (ldc & 3) || (ldaux & 3)) return LTRX_EUNSUPPORTED;
  if (((uintptr_t)A_image | (uintptr_t)B_image | (uintptr_t)C | (uintptr_t)aux_image | (uintptr_t)bias) & 15) return LTRX_EUNSUPPORTED;
  if (!fits_u32(M, lda) || !fits_u32(N, ldb) || !fits_u32(M, ldc) || (aux_image && !fits_u32(M, ldaux))) return LTRX_EUNSUPPORTED;
  const ltrx::DropSpec drop = ltrx_make_drop(drop_p, drop_seed);
  hipStream_t s = (hipStream_t)stream;
  const int tiles_n = N / 256;
  const dim3 grid(((M + 255) / 256) * tiles_n);
  const unsigned char* a = (const unsigned char*)A_image;
  const unsigned char* b = (const unsigned char*)B_image;
  const unsigned char* x = (const unsigned char*)aux_image;
#define LTRX_NT_IMG(ACT_, IMG_)                                                                                            \
  do {                                                                                                                     \
    static bool attr = false;                                                                                              \
    if (!attr) {                                                                                                           \
      if (hipFuncSetAttribute((const void*)ltrx_gemm_nt_img_kernel<ACT_, IMG_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              131072) != hipSuccess)                                                                       \
        return LTRX_EHIP;                                                                                                  \
      attr = true;                                                                                                         \
    }                                                                                                                      \
    hipLaunchKernelGGL((ltrx_gemm_nt_img_kernel<ACT_, IMG_>), grid, dim3(512), 131072, s, a, lda, b, ldb, C, ldc, M, N, K, bias, x, \
                       ldaux, tiles_n, drop, drop_step);                                                                   \
  } while (0)
  if (c_is_image) {
    if (act == 0) LTRX_NT_IMG(0, true); else if (act == 1) LTRX_NT_IMG(1, true); else LTRX_NT_IMG(2, true);
  } else {
    if (act == 0) LTRX_NT_IMG(0, false); else if (act == 1) LTRX_NT_IMG(1, false); else LTRX_NT_IMG(2, false);
  }
#undef LTRX_NT_IMG
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// one workgroup per CU and ONE round: never more than 256 workgroups; at least 8 sub-steps (128 rows) per split
static void tn_img_plan(int M, int NP, int KP, int* splits, int* mps) {
  const int tiles = (NP / 256) * (KP / 256);
  int sp = 256 / tiles;
  if (sp > M / 128) sp = M / 128;
  if (sp < 1) sp = 1;
  const int m = ((M + sp - 1) / sp + 15) / 16 * 16;
  *mps = m;
  *splits = (M + m - 1) / m;
}

// (workspace: ltrx_gemm_tn_workspace_bytes(M, NP, KP) of ltrx_gemm.hip is an upper bound for this plan too)
extern "C" int ltrx_gemm_tn_img(const void* A_image, int lda, const void* B_image, int ldb, float* C, float* bias_out, int M,
                                int NP, int KP, void* ws, ltrx_stream_t stream) {
  if (!A_image || !B_image || !C || !ws || M <= 0 || NP <= 0 || KP <= 0) return LTRX_EINVAL;
  if ((NP % 256) || (KP % 256) || (M % 16) || lda < NP || ldb < KP || (lda & 3) || (ldb & 3)) return LTRX_EUNSUPPORTED;
  if (((uintptr_t)A_image | (uintptr_t)B_image | (uintptr_t)C | (uintptr_t)ws) & 15) return LTRX_EUNSUPPORTED;
  if (!fits_u32(M, lda) || !fits_u32(M, ldb) || (NP / 256) * (KP / 256) > 256) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int splits, mps;
  tn_img_plan(M, NP, KP, &splits, &mps);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)ltrx_gemm_tn_img_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess)
      return LTRX_EHIP;
    attr = true;
  }
  float* bslabs = bias_out ? (float*)ws + (size_t)splits * NP * KP : nullptr;
  hipLaunchKernelGGL(ltrx_gemm_tn_img_kernel, dim3((NP / 256) * (KP / 256), splits), dim3(512), 131072, s, (const unsigned char*)A_image,
                     lda, (const unsigned char*)B_image, ldb, (float*)ws, bslabs, M, NP, KP, KP / 256, mps);
  LTRX_LAUNCH_CHECK();
  const size_t n = (size_t)NP * KP;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const size_t bblocks = bias_out ? ((size_t)NP + 255) / 256 : 0;
  hipLaunchKernelGGL(ltrx_gemm_img_slab_reduce_kernel, dim3((unsigned)(blocks + bblocks)), dim3(256), 0, s, (const float*)ws, splits, n, C,
                     (int)blocks, (const float*)bslabs, (size_t)NP, bias_out);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
