// Fused masked self-attention over a slate, forward and backward, on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
// Reference: allrank/models/transformer.py:137-156 (attention) as called from MultiHeadedAttention.forward
// (:178-203): scores = q k^T / sqrt(d_k); masked_fill(key padded, -inf); softmax over keys; p_attn @ v.
// The [B,h,L,L] score / probability tensors of the reference are never materialised (flash-style online
// softmax); only the row log-sum-exp [B,h,L] is kept for the backward, which recomputes P tile by tile.
//
// ---- MFMA tiling (wave64, one 32x32 output tile per wave-instruction chain) -------------------------------------
// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32]; lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; D register r of lane l is D[row(r, l>>5)][l&31], row(r,h) = (r&3) + 8*(r>>2) + 4*h.
// Exact fp32 products and accumulation (bit-equal to an fmaf chain) -> strict-parity arithmetic.
//
// Both contractions are arranged so that the wave's FIXED operand index (its 32 queries, or its 32 keys in the
// dK/dV kernel) is the lane index l&31, and the STREAMED operand (a 32-row tile staged in LDS) supplies rows:
//   "rows x fixed" product  (QK^T, dO V^T):   acc[r] = sum_c TILE[row(r,h)][c] * FIXED[l&31][c]
//        A = TILE[l&31][c], B = FIXED[l&31][c] with c = h*DKP/2 + t at step t (the k index of the MFMA is free to
//        be permuted: each half-wave walks its own half of the head dimension -> every lane reads 128 contiguous
//        bytes of its LDS row with ds_read_b128 and keeps its FIXED fragment in DKP/2 registers).
//   "cols x fixed" product  (P V, dS K, P^T dO, dS^T Q):  out[ct][r'] (+)= sum_row TILE[row][32 ct + (l&31)] * P[row][l&31]
//        the probabilities produced by the first product are ALREADY in B-operand position (register r of half h is
//        row(r,h)), so P never moves between lanes: A = TILE[row(r,h)][32 ct + (l&31)], B = p[r].
// With this "swapped" orientation softmax statistics of a query are lane-local: a max/sum over 16 registers plus
// one exchange with lane^32.
//
// ---- kernels ----------------------------------------------------------------------------------------------------
//   fwd   : workgroup = 4 waves = 128 queries of one (slate, head); loops over 32-key tiles (K and V staged in LDS).
//   bwd dq: same decomposition; per key tile: S, dP = dO V^T, dS = P (dP - delta) / sqrt(dk), dQ += dS K.
//   bwd dkdv: workgroup = 4 waves = 128 keys; loops over 32-query tiles (Q and dO staged in LDS):
//             dV += P^T dO, dK += dS^T Q.  No atomics anywhere: every output element has one owner (deterministic).
// Arithmetic intensity: 4 L d_k flop per (query, head) against 4 d_k-float rows read once per workgroup -> compute
// (fp32 MFMA) bound; K/V re-reads by the 2 workgroups of a (slate, head) are served by L2.
#include "ltrx_device.h"

using namespace ltrx;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int DKP>
struct Tile {
  static constexpr int LDT = DKP + 4;          // row stride in floats (16-B aligned rows, conflict-free b128 reads)
  static constexpr int FLOATS = 32 * LDT;
};

// Stage a [32][dk] slab (rows row0.., global row stride `rs` floats) into an LDS tile, zero-filling rows >= nrows
// and columns >= dk.  All 256 threads participate; 16-B loads (dk % 4 == 0 is required by the host wrapper).
template <int DKP>
__device__ __forceinline__ void stage_tile(float* tile, const float* __restrict__ base, int row0, int nrows, int dk,
                                           size_t rs) {
  constexpr int C4 = DKP / 4;
  for (int idx = threadIdx.x; idx < 32 * C4; idx += blockDim.x) {
    const int r = idx / C4, c = (idx % C4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < nrows && c < dk) v = *reinterpret_cast<const float4*>(base + (size_t)(row0 + r) * rs + c);
    *reinterpret_cast<float4*>(tile + r * Tile<DKP>::LDT + c) = v;
  }
}

// Split staging for software prefetch: tile_gload issues the global loads of a [32][dk] slab into registers (DKP/32
// float4 per thread), tile_sstore writes them to the LDS tile later -- the loads of tile t+1 fly behind the MFMAs of
// tile t instead of being waited for right after their issue.
template <int DKP>
struct TileRegs {
  float4 v[DKP / 32];
};
template <int DKP>
__device__ __forceinline__ void tile_gload(TileRegs<DKP>& r, const float* __restrict__ base, int row0, int nrows, int dk,
                                           size_t rs) {
  constexpr int C4 = DKP / 4;
#pragma unroll
  for (int p = 0; p < DKP / 32; ++p) {
    const int idx = threadIdx.x + 256 * p;
    const int row = idx / C4, c = (idx % C4) * 4;
    r.v[p] = (row0 + row < nrows && c < dk) ? *reinterpret_cast<const float4*>(base + (size_t)(row0 + row) * rs + c)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int DKP>
__device__ __forceinline__ void tile_sstore(float* tile, const TileRegs<DKP>& r) {
  constexpr int C4 = DKP / 4;
#pragma unroll
  for (int p = 0; p < DKP / 32; ++p) {
    const int idx = threadIdx.x + 256 * p;
    const int row = idx / C4, c = (idx % C4) * 4;
    *reinterpret_cast<float4*>(tile + row * Tile<DKP>::LDT + c) = r.v[p];
  }
}

// FIXED fragment of the wave: lane keeps FIXED[row0 + (l&31)][half*DKP/2 + t], t = 0..DKP/2-1 (zero beyond nrows/dk).
template <int DKP>
__device__ __forceinline__ void load_fixed(float (&frag)[DKP / 2], const float* __restrict__ base, int row0, int nrows,
                                           int dk, size_t rs) {
  const int row = row0 + (threadIdx.x & 31);
  const int c0 = ((threadIdx.x & 63) >> 5) * (DKP / 2);
#pragma unroll
  for (int t4 = 0; t4 < DKP / 8; ++t4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = c0 + 4 * t4;
    if (row < nrows && c < dk) v = *reinterpret_cast<const float4*>(base + (size_t)row * rs + c);
    frag[4 * t4 + 0] = v.x;
    frag[4 * t4 + 1] = v.y;
    frag[4 * t4 + 2] = v.z;
    frag[4 * t4 + 3] = v.w;
  }
}

// acc[r] = sum_c TILE[row(r,half)][c] * FIXED[l&31][c]
template <int DKP>
__device__ __forceinline__ f32x16 rows_x_fixed(const float* tile, const float (&frag)[DKP / 2]) {
  const int lane = threadIdx.x & 63;
  const float* rowp = tile + (lane & 31) * Tile<DKP>::LDT + (lane >> 5) * (DKP / 2);
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t4 = 0; t4 < DKP / 8; ++t4) {
    const float4 a = *reinterpret_cast<const float4*>(rowp + 4 * t4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, frag[4 * t4 + 0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, frag[4 * t4 + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, frag[4 * t4 + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, frag[4 * t4 + 3], acc, 0, 0, 0);
  }
  return acc;
}

// same product with the wave's FIXED operand read from an LDS image ([32 rows of the wave][LDT], same layout as a tile)
// instead of DKP/2 registers: frees 32 VGPRs (the dQ kernel drops from 194 to <= 168 and runs 3 waves per SIMD)
template <int DKP>
__device__ __forceinline__ f32x16 rows_x_fixed_lds(const float* tile, const float* fixed_img) {
  const int lane = threadIdx.x & 63;
  const float* rowp = tile + (lane & 31) * Tile<DKP>::LDT + (lane >> 5) * (DKP / 2);
  const float* fixp = fixed_img + (lane & 31) * Tile<DKP>::LDT + (lane >> 5) * (DKP / 2);
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t4 = 0; t4 < DKP / 8; ++t4) {
    const float4 a = *reinterpret_cast<const float4*>(rowp + 4 * t4);
    const float4 f = *reinterpret_cast<const float4*>(fixp + 4 * t4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, f.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, f.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, f.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, f.w, acc, 0, 0, 0);
  }
  return acc;
}

// out[ct][r'] += sum_row TILE[row][32 ct + (l&31)] * p[row]   (p[r] belongs to row(r,half))
template <int DKP>
__device__ __forceinline__ void cols_x_p(const float* tile, const f32x16& p, f32x16 (&out)[DKP / 32]) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5;
  const float* colp = tile + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* rp = colp + rowmap(r, half) * Tile<DKP>::LDT;
#pragma unroll
    for (int ct = 0; ct < DKP / 32; ++ct) out[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(rp[32 * ct], p[r], out[ct], 0, 0, 0);
  }
}

// Store OUT^T accumulators: lane owns output row (row0 + l&31); register r of out[ct] is column 32 ct + row(r,half).
// Registers 4g..4g+3 are 4 consecutive columns -> one 16-B store each.
template <int DKP>
__device__ __forceinline__ void store_rows(float* __restrict__ base, int row0, int nrows, int dk, size_t rs,
                                           const f32x16 (&out)[DKP / 32], float scale) {
  const int lane = threadIdx.x & 63;
  const int row = row0 + (lane & 31);
  if (row >= nrows) return;
  float* rp = base + (size_t)row * rs;
#pragma unroll
  for (int ct = 0; ct < DKP / 32; ++ct) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * ct + 8 * g + 4 * (lane >> 5);
      if (c < dk)
        *reinterpret_cast<float4*>(rp + c) = make_float4(out[ct][4 * g + 0] * scale, out[ct][4 * g + 1] * scale,
                                                         out[ct][4 * g + 2] * scale, out[ct][4 * g + 3] * scale);
    }
  }
}

// Dropout on the attention probabilities (transformer.py:154-155): counter-based, two-level -- a fully mixed 32-bit seed
// per (slate*head, query) row and a short 2-multiply mix per key, so the per-element cost is ~8 VALU operations with 32-bit
// arithmetic only.  The forward and both backward kernels regenerate the same mask from (seed, row, key).
typedef DropSpec DropCfg;
__device__ __forceinline__ uint32_t drop_row_seed(const DropCfg& d, uint32_t bh, int L, int qrow) {
  uint32_t x = d.seed ^ ((bh * (uint32_t)L + (uint32_t)qrow) * 0x9E3779B9u);
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float drop_scale_rk(const DropCfg& d, uint32_t row_seed, int key) {
  uint32_t x = (row_seed ^ (uint32_t)key) * 0x9E3779B1u;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  return ((x >> 8) >= d.thresh) ? d.inv_keep : 0.f;
}

// Softmax arithmetic runs in the log2 domain: one v_exp_f32 per probability (exp2 of scores pre-multiplied by
// log2(e)/sqrt(d_k)) instead of the 14-instruction expf expansion, and key padding enters as an additive -inf bias, so
// masked and out-of-range entries fall out of exp2(-inf) = 0 without compare/select pairs.  The backward kernels recompute
// P = exp2(s * log2e/sqrt(d_k) - lse * log2e) from the stored natural-log LSE (lse_out keeps its meaning).
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int DKP>
__device__ __forceinline__ void zero_acc(f32x16 (&o)[DKP / 32]) {
#pragma unroll
  for (int ct = 0; ct < DKP / 32; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <int DKP, bool DROP>
__global__ void __launch_bounds__(256, 2) ltrx_mha_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v,
                                                              const uint8_t* __restrict__ kpm, int L, int h, int dk,
                                                              int rs, float* __restrict__ o, int ors,
                                                              float* __restrict__ lse, float scale, DropCfg drop,
                                                              const uint32_t* __restrict__ drop_step,
                                                              const int* __restrict__ cu, const int* __restrict__ order) {
  if (DROP && drop_step) drop.seed ^= drop_step[0] * 0x9E3779B9u;
  __shared__ __attribute__((aligned(16))) float ktile[Tile<DKP>::FLOATS];
  __shared__ __attribute__((aligned(16))) float vtile[Tile<DKP>::FLOATS];
  __shared__ float kmask[32];
  // grid = (slate*head, 128-row block): the (slate, head) index is the FAST dimension, so the workgroups of row block 0 have
  // consecutive linear ids and spread over all 8 XCDs (ids are dealt to XCDs round-robin).  With the row block as the fast
  // dimension a batch of short slates (one row block each, the second one exiting at once) put every live workgroup on an
  // even id, i.e. on 4 of the 8 XCDs: measured 162 vs 302 us for 1/4 of the tile work.
  const int head = blockIdx.x % h;
  const int b = order ? order[blockIdx.x / h] : (int)(blockIdx.x / h);   // launch order (longest slates first) -> slate
  const int bh = b * h + head;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int q0 = blockIdx.y * 128 + wave * 32;
  // variable-length (compacted) batches: cu[b] = first row of slate b, cu[b+1] - cu[b] its item count; L stays the stride of
  // the per-query statistics (lse / delta) and of the dropout row hash
  const int Lmax = L;
  const size_t slate = cu ? (size_t)cu[b] : (size_t)b * L;
  if (cu) L = cu[b + 1] - cu[b];
  if ((int)(blockIdx.y * 128) >= L) return;          // whole workgroup beyond this slate (uniform: no barrier is skipped)
  const float* qb = q + slate * rs + (size_t)head * dk;
  const float* kb = k + slate * rs + (size_t)head * dk;
  const float* vb = v + slate * rs + (size_t)head * dk;

  float qfrag[DKP / 2];
  load_fixed<DKP>(qfrag, qb, q0, L, dk, rs);
  f32x16 oacc[DKP / 32];
  zero_acc<DKP>(oacc);
  float m = -INFINITY, l = 0.f;        // running max (log2 domain) and normaliser
  const uint32_t drow = DROP ? drop_row_seed(drop, bh, Lmax, q0 + (lane & 31)) : 0u;
  const float sl2 = scale * kLog2e;

  const int nkt = (L + 31) / 32;
  TileRegs<DKP> kr, vr;
  tile_gload<DKP>(kr, kb, 0, L, dk, rs);
  tile_gload<DKP>(vr, vb, 0, L, dk, rs);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();   // previous tile fully consumed
    tile_sstore<DKP>(ktile, kr);
    tile_sstore<DKP>(vtile, vr);
    if (threadIdx.x < 32) {
      const int key = kt * 32 + threadIdx.x;
      kmask[threadIdx.x] = (key >= L || (kpm && kpm[slate + key])) ? -INFINITY : 0.f;       // additive bias
    }
    __syncthreads();
    if (kt + 1 < nkt) {   // prefetch: in flight during this tile's MFMAs
      tile_gload<DKP>(kr, kb, (kt + 1) * 32, L, dk, rs);
      tile_gload<DKP>(vr, vb, (kt + 1) * 32, L, dk, rs);
    }
    f32x16 s = rows_x_fixed<DKP>(ktile, qfrag);
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = s[r] * sl2 + kmask[rowmap(r, half)];          // log2 domain
      mt = fmaxf(mt, s[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float mref = (mn == -INFINITY) ? 0.f : mn;             // all keys so far padded: keep the differences finite
    const float alpha = fast_exp2(m - mref);                      // m == -inf -> 0
    float ps = 0.f;
    f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = fast_exp2(s[r] - mref);
      ps += p[r];
    }
    l = l * alpha + ps;                      // the softmax normaliser counts every key (dropout comes after softmax)
    if (DROP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r] *= drop_scale_rk(drop, drow, kt * 32 + rowmap(r, half));
    }
#pragma unroll
    for (int ct = 0; ct < DKP / 32; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[ct][r] *= alpha;
    cols_x_p<DKP>(vtile, p, oacc);
    m = mn;
  }
  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = (lt > 0.f) ? 1.0f / lt : 0.f;   // a row whose keys are all padded: 0 (the reference yields NaN)
  store_rows<DKP>(o + slate * ors + (size_t)head * dk, q0, L, dk, ors, oacc, inv);
  const int qrow = q0 + (lane & 31);
  if (half == 0 && qrow < L) lse[((size_t)b * h + head) * Lmax + qrow] = (lt > 0.f) ? (m + log2f(lt)) * kLn2 : 0.f;   // natural log
}

// ------------------------------------------------------------------------------------------------------------------
// backward: dQ   (wave owns 32 queries, streams key tiles)
// ------------------------------------------------------------------------------------------------------------------
template <int DKP, bool DROP>
__global__ void __launch_bounds__(256, (DKP <= 64) ? 3 : (DKP <= 96 ? 2 : 1)) ltrx_mha_bwd_dq_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ kpm, const float* __restrict__ o, const float* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ delta, int L, int h, int dk, int rs, int ors,
    float* __restrict__ dq, int drs, float scale, DropCfg drop, const uint32_t* __restrict__ drop_step,
    const int* __restrict__ cu, const int* __restrict__ order) {
  if (DROP && drop_step) drop.seed ^= drop_step[0] * 0x9E3779B9u;
  __shared__ __attribute__((aligned(16))) float ktile[Tile<DKP>::FLOATS];
  __shared__ __attribute__((aligned(16))) float vtile[Tile<DKP>::FLOATS];
  __shared__ float kmask[32];
  // dO rows of the workgroup's 128 queries (the fixed operand of dP = dO V^T), one 32-row image per wave
  __shared__ __attribute__((aligned(16))) float doimg[4 * Tile<DKP>::FLOATS];
  // grid = (slate*head, 128-row block): the (slate, head) index is the FAST dimension, so the workgroups of row block 0 have
  // consecutive linear ids and spread over all 8 XCDs (ids are dealt to XCDs round-robin).  With the row block as the fast
  // dimension a batch of short slates (one row block each, the second one exiting at once) put every live workgroup on an
  // even id, i.e. on 4 of the 8 XCDs: measured 162 vs 302 us for 1/4 of the tile work.
  const int head = blockIdx.x % h;
  const int b = order ? order[blockIdx.x / h] : (int)(blockIdx.x / h);   // launch order (longest slates first) -> slate
  const int bh = b * h + head;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int q0 = blockIdx.y * 128 + wave * 32;
  // variable-length (compacted) batches: cu[b] = first row of slate b, cu[b+1] - cu[b] its item count; L stays the stride of
  // the per-query statistics (lse / delta) and of the dropout row hash
  const int Lmax = L;
  const size_t slate = cu ? (size_t)cu[b] : (size_t)b * L;
  if (cu) L = cu[b + 1] - cu[b];
  if ((int)(blockIdx.y * 128) >= L) return;          // whole workgroup beyond this slate (uniform: no barrier is skipped)
  const float* kb = k + slate * rs + (size_t)head * dk;
  const float* vb = v + slate * rs + (size_t)head * dk;
  float qfrag[DKP / 2];
  load_fixed<DKP>(qfrag, q + slate * rs + (size_t)head * dk, q0, L, dk, rs);
  float* myimg = doimg + wave * Tile<DKP>::FLOATS;
  const int qrow = q0 + (lane & 31);
  const size_t stat = ((size_t)b * h + head) * Lmax + qrow;
  const float lse_q = (qrow < L) ? lse[stat] * kLog2e : 0.f;          // log2 domain
  const uint32_t drow = DROP ? drop_row_seed(drop, bh, Lmax, qrow) : 0u;
  const float sl2 = scale * kLog2e;
  // delta_q = <dO_q, O_q> (rowsum(dP * P)); each half-wave holds half of the head dimension.  Published for the
  // dK/dV kernel, which is launched after this one on the same stream.
  float del_q = 0.f;
  {
    float ofrag[DKP / 2], dofrag[DKP / 2];
    load_fixed<DKP>(ofrag, o + slate * ors + (size_t)head * dk, q0, L, dk, ors);
    load_fixed<DKP>(dofrag, dout + slate * ors + (size_t)head * dk, q0, L, dk, ors);
#pragma unroll
    for (int t = 0; t < DKP / 2; ++t) del_q += dofrag[t] * ofrag[t];
    float* mine = myimg + (lane & 31) * Tile<DKP>::LDT + half * (DKP / 2);       // the lane's own half row (wave-private image)
#pragma unroll
    for (int t4 = 0; t4 < DKP / 8; ++t4)
      *reinterpret_cast<float4*>(mine + 4 * t4) = make_float4(dofrag[4 * t4], dofrag[4 * t4 + 1], dofrag[4 * t4 + 2], dofrag[4 * t4 + 3]);
    del_q += __shfl_xor(del_q, 32, 64);
    if (half == 0 && qrow < L) delta[stat] = del_q;
  }
  f32x16 dqacc[DKP / 32];
  zero_acc<DKP>(dqacc);
  const int nkt = (L + 31) / 32;
  TileRegs<DKP> kr, vr;
  tile_gload<DKP>(kr, kb, 0, L, dk, rs);
  tile_gload<DKP>(vr, vb, 0, L, dk, rs);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    tile_sstore<DKP>(ktile, kr);
    tile_sstore<DKP>(vtile, vr);
    if (threadIdx.x < 32) {
      const int key = kt * 32 + threadIdx.x;
      kmask[threadIdx.x] = (key >= L || (kpm && kpm[slate + key])) ? -INFINITY : 0.f;       // additive bias
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_gload<DKP>(kr, kb, (kt + 1) * 32, L, dk, rs);
      tile_gload<DKP>(vr, vb, (kt + 1) * 32, L, dk, rs);
    }
    const f32x16 s = rows_x_fixed<DKP>(ktile, qfrag);
    const f32x16 dp = rows_x_fixed_lds<DKP>(vtile, myimg);
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(s[r] * sl2 + kmask[rowmap(r, half)] - lse_q);
      const float dm = DROP ? drop_scale_rk(drop, drow, kt * 32 + rowmap(r, half)) : 1.0f;
      ds[r] = p * (dp[r] * dm - del_q) * scale;
    }
    cols_x_p<DKP>(ktile, ds, dqacc);
  }
  store_rows<DKP>(dq + slate * drs + (size_t)head * dk, q0, L, dk, drs, dqacc, 1.0f);
}

// ------------------------------------------------------------------------------------------------------------------
// backward: dK, dV   (wave owns 32 keys, streams query tiles)
// ------------------------------------------------------------------------------------------------------------------
template <int DKP, bool DROP>
__global__ void __launch_bounds__(256, (DKP <= 64) ? 2 : 1) ltrx_mha_bwd_dkdv_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ kpm, const float* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, int L, int h, int dk, int rs, int ors, float* __restrict__ dkout,
    float* __restrict__ dvout, int drs, float scale, DropCfg drop, const uint32_t* __restrict__ drop_step,
    const int* __restrict__ cu, const int* __restrict__ order) {
  if (DROP && drop_step) drop.seed ^= drop_step[0] * 0x9E3779B9u;
  __shared__ __attribute__((aligned(16))) float qtile[Tile<DKP>::FLOATS];
  __shared__ __attribute__((aligned(16))) float dotile[Tile<DKP>::FLOATS];
  __shared__ float lse_t[32];
  __shared__ float del_t[32];
  __shared__ uint32_t drow_t[32];
  // grid = (slate*head, 128-row block): the (slate, head) index is the FAST dimension, so the workgroups of row block 0 have
  // consecutive linear ids and spread over all 8 XCDs (ids are dealt to XCDs round-robin).  With the row block as the fast
  // dimension a batch of short slates (one row block each, the second one exiting at once) put every live workgroup on an
  // even id, i.e. on 4 of the 8 XCDs: measured 162 vs 302 us for 1/4 of the tile work.
  const int head = blockIdx.x % h;
  const int b = order ? order[blockIdx.x / h] : (int)(blockIdx.x / h);   // launch order (longest slates first) -> slate
  const int bh = b * h + head;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.y * 128 + wave * 32;
  // variable-length (compacted) batches: cu[b] = first row of slate b, cu[b+1] - cu[b] its item count; L stays the stride of
  // the per-query statistics (lse / delta) and of the dropout row hash
  const int Lmax = L;
  const size_t slate = cu ? (size_t)cu[b] : (size_t)b * L;
  if (cu) L = cu[b + 1] - cu[b];
  if ((int)(blockIdx.y * 128) >= L) return;          // whole workgroup beyond this slate (uniform: no barrier is skipped)
  const float* qb = q + slate * rs + (size_t)head * dk;
  const float* dob = dout + slate * ors + (size_t)head * dk;
  float kfrag[DKP / 2], vfrag[DKP / 2];
  load_fixed<DKP>(kfrag, k + slate * rs + (size_t)head * dk, k0, L, dk, rs);
  load_fixed<DKP>(vfrag, v + slate * rs + (size_t)head * dk, k0, L, dk, rs);
  const int key = k0 + (lane & 31);
  const bool key_masked = (key >= L) || (kpm && kpm[slate + (key < L ? key : 0)] != 0);
  const float kbias = key_masked ? -INFINITY : 0.f;
  const float sl2 = scale * kLog2e;
  f32x16 dkacc[DKP / 32], dvacc[DKP / 32];
  zero_acc<DKP>(dkacc);
  zero_acc<DKP>(dvacc);
  const size_t statb = ((size_t)b * h + head) * Lmax;
  const int nqt = (L + 31) / 32;
  TileRegs<DKP> qr, dor;
  tile_gload<DKP>(qr, qb, 0, L, dk, rs);
  tile_gload<DKP>(dor, dob, 0, L, dk, ors);
  for (int qt = 0; qt < nqt; ++qt) {
    __syncthreads();
    tile_sstore<DKP>(qtile, qr);
    tile_sstore<DKP>(dotile, dor);
    if (threadIdx.x < 32) {
      const int qrow = qt * 32 + threadIdx.x;
      lse_t[threadIdx.x] = (qrow < L) ? lse[statb + qrow] * kLog2e : INFINITY;   // +inf -> P = exp2(-inf) = 0 for rows >= L
      del_t[threadIdx.x] = (qrow < L) ? delta[statb + qrow] : 0.f;
      if (DROP) drow_t[threadIdx.x] = drop_row_seed(drop, bh, Lmax, qrow);
    }
    __syncthreads();
    if (qt + 1 < nqt) {
      tile_gload<DKP>(qr, qb, (qt + 1) * 32, L, dk, rs);
      tile_gload<DKP>(dor, dob, (qt + 1) * 32, L, dk, ors);
    }
    const f32x16 s = rows_x_fixed<DKP>(qtile, kfrag);     // S[q = row(r,half)][key = l&31]
    f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = fast_exp2(s[r] * sl2 + kbias - lse_t[rowmap(r, half)]);
    f32x16 dm;                                             // dropout keep-scale of (q, key), 1 when dropout is off
#pragma unroll
    for (int r = 0; r < 16; ++r)
      dm[r] = DROP ? drop_scale_rk(drop, drow_t[rowmap(r, half)], key) : 1.0f;
    {
      f32x16 pd;
#pragma unroll
      for (int r = 0; r < 16; ++r) pd[r] = p[r] * dm[r];
      cols_x_p<DKP>(dotile, pd, dvacc);                    // dV^T[c][key] += sum_q dO[q][c] (P*M)[q][key]
    }
    const f32x16 dp = rows_x_fixed<DKP>(dotile, vfrag);    // dP[q][key] (before the dropout mask)
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = p[r] * (dp[r] * dm[r] - del_t[rowmap(r, half)]) * scale;
    cols_x_p<DKP>(qtile, ds, dkacc);                       // dK^T[c][key] += sum_q Q[q][c] dS[q][key]
  }
  store_rows<DKP>(dkout + slate * drs + (size_t)head * dk, k0, L, dk, drs, dkacc, 1.0f);
  store_rows<DKP>(dvout + slate * drs + (size_t)head * dk, k0, L, dk, drs, dvacc, 1.0f);
}

// ------------------------------------------------------------------------------------------------------------------
// MFMA layout self-test (one wave): D[32x32] = A[32x2] * B[2x32] through the layout assumptions used above.
// ------------------------------------------------------------------------------------------------------------------
__global__ void ltrx_selftest_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ D) {
  const int lane = threadIdx.x & 63, half = lane >> 5;
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(lane & 31) * 2 + half], Bm[half * 32 + (lane & 31)], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[rowmap(r, half) * 32 + (lane & 31)] = acc[r];
}

extern "C" int ltrx_selftest_mfma32x32x2(const float* A, const float* Bm, float* D, ltrx_stream_t stream) {
  if (!A || !Bm || !D) return LTRX_EINVAL;
  hipLaunchKernelGGL(ltrx_selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, D);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// host wrappers
// ------------------------------------------------------------------------------------------------------------------
// arithmetic of the attention contractions:
// (`mode` is an argument of every ltrx_mha_fwd / ltrx_mha_bwd call -- the library keeps no attention mode of its own)
//   mode 1: split-bf16 on the bf16 MFMA with the whole slate resident in LDS (ltrx_mha_res.hip) wherever the
//           shape fits (slate length <= 256, 32 < d_k <= 64); fp32-class (three bf16 products per fp32 product, like the dense
//           projections).  Other shapes run the exact kernels of this file.
//   mode 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-exact fp32 products) for every shape -- the strict reference.
//   mode 2: the kernels of mode 1 with ONE bf16 product per contraction (plain bf16 operands, fp32 accumulate and softmax):
//           the throughput mode, about 2^-9 relative error per product -- NOT the parity arithmetic.
bool ltrx_mha_res_fits(int L, int dk);
int ltrx_mha_fwd_res_launch(const float* q, const float* k, const float* v, const uint8_t* kpm, int B, int L, int h, int dk, int rs,
                            float* o, int ors, float* lse, float p_drop, uint32_t seed, const uint32_t* seed_step, const int* cu,
                            const int* order, bool plain, hipStream_t s);
int ltrx_mha_bwd_res_launch(const float* q, const float* k, const float* v, const uint8_t* kpm, const float* o, const float* dout,
                            const float* lse, int B, int L, int h, int dk, int rs, int ors, float* dq, float* dkk, float* dv, int drs,
                            void* ws, float p_drop, uint32_t seed, const uint32_t* seed_step, const int* cu, const int* order,
                            bool plain, hipStream_t s);
bool ltrx_mha_res_bwd_fits(int L, int dk);
size_t ltrx_mha_res_bwd_ws_bytes(int B, int L, int h);

static int mha_check(int B, int L, int h, int dk, int rs, int ors) {
  if (B <= 0 || L <= 0 || h <= 0 || dk <= 0) return LTRX_EINVAL;
  if (dk % 4 != 0 || dk > 128 || rs % 4 != 0 || ors % 4 != 0 || rs < h * dk || ors < h * dk) return LTRX_EUNSUPPORTED;
  if ((long long)B * h > 0x7fffffffLL || (L + 127) / 128 > 65535) return LTRX_EUNSUPPORTED;
  return LTRX_OK;
}

#define LTRX_DKP_DISPATCH(dk, CALL) \
  do {                              \
    if ((dk) <= 32) { CALL(32); }   \
    else if ((dk) <= 64) { CALL(64); } \
    else if ((dk) <= 96) { CALL(96); } \
    else { CALL(128); }             \
  } while (0)

static DropCfg make_drop(float p_drop, uint32_t seed) { return ltrx_make_drop(p_drop, seed); }

extern "C" int ltrx_mha_fwd(const float* q, const float* k, const float* v, const uint8_t* key_pad_mask, int B, int L,
                            int h, int d_k, int row_stride, float* o, int o_row_stride, float* lse_out, float p_drop,
                            uint32_t seed, const uint32_t* seed_step, const int32_t* cu_seqlens, const int32_t* slate_order,
                            int mode, ltrx_stream_t stream) {
  if (!q || !k || !v || (!key_pad_mask && !cu_seqlens) || !o || !lse_out || !(p_drop >= 0.f) || p_drop >= 1.f) return LTRX_EINVAL;
  if (mode < 0 || mode > 2) return LTRX_EINVAL;
  int rc = mha_check(B, L, h, d_k, row_stride, o_row_stride);
  if (rc != LTRX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (mode != 0 && ltrx_mha_res_fits(L, d_k))
    return ltrx_mha_fwd_res_launch(q, k, v, key_pad_mask, B, L, h, d_k, row_stride, o, o_row_stride, lse_out, p_drop, seed, seed_step,
                                   cu_seqlens, slate_order, mode == 2, s);
  const DropCfg drop = make_drop(p_drop, seed);
  const dim3 grid(B * h, (L + 127) / 128);
  const float scale = 1.0f / sqrtf((float)d_k);
#define CALL(DKP)                                                                                                 \
  if (drop.thresh != 0u)                                                                                          \
    hipLaunchKernelGGL((ltrx_mha_fwd_kernel<DKP, true>), grid, dim3(256), 0, s, q, k, v, key_pad_mask, L, h, d_k, row_stride, \
                       o, o_row_stride, lse_out, scale, drop, seed_step, cu_seqlens, slate_order);                              \
  else                                                                                                            \
    hipLaunchKernelGGL((ltrx_mha_fwd_kernel<DKP, false>), grid, dim3(256), 0, s, q, k, v, key_pad_mask, L, h, d_k, row_stride, \
                       o, o_row_stride, lse_out, scale, drop, seed_step, cu_seqlens, slate_order)
  LTRX_DKP_DISPATCH(d_k, CALL);
#undef CALL
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// The split-bf16 backward hands dS[B, h, LK, LK] (LK = L rounded up to 64) from its dK/dV kernel to its dQ kernel through the
// workspace: 134 MB at 64 x 8 x 240, 537 MB at 16 x 8 x 1024 -- and 8.6 GB at 64 x 8 x 2048.  Above LTRX_MHA_DS_BUDGET_BYTES the call
// uses the exact-fp32 kernels instead (workspace B L h floats, no O(L^2) memory) -- same contract, fp32 MFMA rate.
static bool res_bwd_selected(int B, int L, int h, int d_k) {
  return ltrx_mha_res_bwd_fits(L, d_k) && ltrx_mha_res_bwd_ws_bytes(B, L, h) <= (size_t)LTRX_MHA_DS_BUDGET_BYTES;
}

extern "C" size_t ltrx_mha_bwd_workspace_bytes(int B, int L, int h, int d_k, int mode) {
  if (B <= 0 || L <= 0 || h <= 0 || d_k <= 0) return 0;
  if (mode != 0 && res_bwd_selected(B, L, h, d_k)) return ltrx_mha_res_bwd_ws_bytes(B, L, h);     // the dS exchange
  return (size_t)B * L * h * sizeof(float);
}

extern "C" int ltrx_mha_bwd(const float* q, const float* k, const float* v, const uint8_t* key_pad_mask, const float* o,
                            const float* dout, const float* lse, int B, int L, int h, int d_k, int row_stride,
                            int o_row_stride, float* dq, float* dk, float* dv, int d_row_stride, float p_drop,
                            uint32_t seed, const uint32_t* seed_step, const int32_t* cu_seqlens, const int32_t* slate_order,
                            int mode, void* ws, ltrx_stream_t stream) {
  if (!q || !k || !v || (!key_pad_mask && !cu_seqlens) || !o || !dout || !lse || !dq || !dk || !dv || !ws) return LTRX_EINVAL;
  if (mode < 0 || mode > 2) return LTRX_EINVAL;
  if (!(p_drop >= 0.f) || p_drop >= 1.f) return LTRX_EINVAL;
  int rc = mha_check(B, L, h, d_k, row_stride, o_row_stride);
  if (rc != LTRX_OK) return rc;
  if (d_row_stride % 4 != 0 || d_row_stride < h * d_k) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* delta = (float*)ws;
  if (mode != 0 && res_bwd_selected(B, L, h, d_k))
    return ltrx_mha_bwd_res_launch(q, k, v, key_pad_mask, o, dout, lse, B, L, h, d_k, row_stride, o_row_stride, dq, dk, dv, d_row_stride,
                                   ws, p_drop, seed, seed_step, cu_seqlens, slate_order, mode == 2, s);
  const DropCfg drop = make_drop(p_drop, seed);
  const dim3 grid(B * h, (L + 127) / 128);
  const float scale = 1.0f / sqrtf((float)d_k);
#define CALLQ(DKP)                                                                                                   \
  if (drop.thresh != 0u)                                                                                             \
    hipLaunchKernelGGL((ltrx_mha_bwd_dq_kernel<DKP, true>), grid, dim3(256), 0, s, q, k, v, key_pad_mask, o, dout, lse, delta, \
                       L, h, d_k, row_stride, o_row_stride, dq, d_row_stride, scale, drop, seed_step, cu_seqlens, slate_order);    \
  else                                                                                                               \
    hipLaunchKernelGGL((ltrx_mha_bwd_dq_kernel<DKP, false>), grid, dim3(256), 0, s, q, k, v, key_pad_mask, o, dout, lse, delta, \
                       L, h, d_k, row_stride, o_row_stride, dq, d_row_stride, scale, drop, seed_step, cu_seqlens, slate_order)
  LTRX_DKP_DISPATCH(d_k, CALLQ);
#undef CALLQ
  LTRX_LAUNCH_CHECK();
#define CALLK(DKP)                                                                                                     \
  if (drop.thresh != 0u)                                                                                               \
    hipLaunchKernelGGL((ltrx_mha_bwd_dkdv_kernel<DKP, true>), grid, dim3(256), 0, s, q, k, v, key_pad_mask, dout, lse, delta, L, \
                       h, d_k, row_stride, o_row_stride, dk, dv, d_row_stride, scale, drop, seed_step, cu_seqlens, slate_order);      \
  else                                                                                                                 \
    hipLaunchKernelGGL((ltrx_mha_bwd_dkdv_kernel<DKP, false>), grid, dim3(256), 0, s, q, k, v, key_pad_mask, dout, lse, delta, L, \
                       h, d_k, row_stride, o_row_stride, dk, dv, d_row_stride, scale, drop, seed_step, cu_seqlens, slate_order)
  LTRX_DKP_DISPATCH(d_k, CALLK);
#undef CALLK
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
