// Fused masked self-attention, forward and backward, on the bf16 matrix cores with fp32-class accuracy ("split-bf16").
// Same math, decomposition and lane-local softmax as ltrx_mha.hip (exact-fp32 MFMA, kept as the strict mode); here
// every contraction runs as three v_mfma_f32_32x32x16_bf16 products  X Y ~= Xhi Yhi + Xhi Ylo + Xlo Yhi  (x = hi + lo,
// both bf16, fp32 accumulation) -- 5.3x fewer matrix-pipe cycles than the fp32 MFMA at an error of <= 3 * 2^-18 per
// product (measured: |dO|, |dQ|, |dK|, |dV| errors ~1e-6 relative, like the fp32 kernels).
// Reference: allrank/models/transformer.py:137-156 (attention), :178-203 (MultiHeadedAttention.forward).
//
// MFMA operand geometry (v_mfma_f32_32x32x16_bf16, lane l, l31 = l & 31, half = l >> 5):
//   A[i = l31][k = 8 half + e], B[k = 8 half + e][j = l31], e = 0..7 (one 16-byte register quad each);
//   D register r = D[row(r, half)][l31], row(r, h) = (r & 3) + 8 (r >> 2) + 4 h.
// Only the PAIRING of A and B elements matters, so the contraction index may be permuted freely:
//   "rows x fixed"  (S^T = K Q^T, dP^T = V dO^T, S = Q K^T, dP = dO V^T): the streamed 32-row tile lives in LDS as
//       [row][c] bf16 (c contiguous, 16-byte chunks XOR-swizzled by row) -> one ds_read_b128 per (term, 16-deep k-step);
//       the wave's fixed operand sits in registers as pre-split bf16x8 fragments.
//   "cols x P"  (O^T += V^T P^T, dQ^T += K^T dS^T, dV^T += dO^T P, dK^T += Q^T dS): P comes out of the first product in
//       D layout; registers 8u..8u+7 of a lane are tile rows {16u + 4 half + (0..3), 16u + 8 + 4 half + (0..3)}.  The
//       streamed operand is therefore staged TRANSPOSED, [c][pos] with pos = 16 b4 + 8 b2 + 4 b3 + (b1 b0) of the row
//       index, so that those 8 rows are one contiguous 16-byte chunk: P never moves between lanes and the A fragment
//       is again a single ds_read_b128.  The transposition happens on the way into LDS: a lane owns one column c, loads
//       4 consecutive rows (coalesced across lanes) and writes one 8-byte piece.
#include "ltrx_device.h"

using namespace ltrx;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ int perm32(int row) {   // position of tile row `row` inside a transposed image row
  return 16 * (row >> 4) + 8 * ((row >> 2) & 1) + 4 * ((row >> 3) & 1) + (row & 3);
}

// ---- LDS images ------------------------------------------------------------------------------------------------------
// K-style: [32 rows][DKP] bf16; chunk c8 of row r at c8 ^ ((r >> SH) & (CH-1)), CH = DKP/8 chunks, SH = log2(16/CH).
template <int DKP>
__device__ __forceinline__ int koff(int row, int c) {
  constexpr int CH = DKP / 8;
  constexpr int SH = (CH == 4) ? 2 : ((CH == 8) ? 1 : 0);
  return row * DKP + ((((c >> 3) ^ ((row >> SH) & (CH - 1)))) << 3) + (c & 7);
}
// T-style: [DKP rows (= columns c of the tile)][32 positions] bf16; 4 chunks per row, swizzled by (crow >> 2) & 3.
__device__ __forceinline__ int toff(int crow, int pos) { return crow * 32 + ((((pos >> 3) ^ ((crow >> 2) & 3))) << 3) + (pos & 7); }

template <int DKP>
struct KImg {
  __bf16 hi[32 * DKP];
  __bf16 lo[32 * DKP];
};
template <int DKP>
struct TImg {
  __bf16 hi[DKP * 32];
  __bf16 lo[DKP * 32];
};

__device__ __forceinline__ void split4(const float x0, const float x1, const float x2, const float x3, bf16x4& hi, bf16x4& lo) {
  const float x[4] = {x0, x1, x2, x3};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}

// stage rows row0..row0+31 (global row stride rs floats, dk valid columns) as a K-style image
template <int DKP>
__device__ __forceinline__ void stage_k(KImg<DKP>& img, const float* __restrict__ base, int row0, int nrows, int dk, size_t rs) {
  constexpr int C4 = DKP / 4;
  for (int idx = threadIdx.x; idx < 32 * C4; idx += blockDim.x) {
    const int r = idx / C4, c = (idx % C4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < nrows && c < dk) v = *reinterpret_cast<const float4*>(base + (size_t)(row0 + r) * rs + c);
    bf16x4 h, l;
    split4(v.x, v.y, v.z, v.w, h, l);
    const int o = koff<DKP>(r, c);
    *reinterpret_cast<bf16x4*>(&img.hi[o]) = h;
    *reinterpret_cast<bf16x4*>(&img.lo[o]) = l;
  }
}

// stage the same 32 rows TRANSPOSED: image row = column c, position = perm32(tile row)
template <int DKP>
__device__ __forceinline__ void stage_t(TImg<DKP>& img, const float* __restrict__ base, int row0, int nrows, int dk, size_t rs) {
  const int c = threadIdx.x % DKP;
  for (int g = threadIdx.x / DKP; g < 8; g += blockDim.x / DKP) {     // 8 groups of 4 consecutive tile rows
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = row0 + 4 * g + e;
      x[e] = (r < nrows && c < dk) ? base[(size_t)r * rs + c] : 0.f;
    }
    bf16x4 h, l;
    split4(x[0], x[1], x[2], x[3], h, l);
    const int o = toff(c, perm32(4 * g));
    *reinterpret_cast<bf16x4*>(&img.hi[o]) = h;
    *reinterpret_cast<bf16x4*>(&img.lo[o]) = l;
  }
}

// the wave's fixed operand: FIXED[row0 + l31][16 ks + 8 half + (0..7)], pre-split
template <int DKP>
__device__ __forceinline__ void load_fixed(bf16x8 (&fh)[DKP / 16], bf16x8 (&fl)[DKP / 16], const float* __restrict__ base,
                                           int row0, int nrows, int dk, size_t rs) {
  const int row = row0 + (threadIdx.x & 31);
  const int half = (threadIdx.x & 63) >> 5;
#pragma unroll
  for (int ks = 0; ks < DKP / 16; ++ks) {
    const int c = 16 * ks + 8 * half;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (row < nrows && c < dk) v0 = *reinterpret_cast<const float4*>(base + (size_t)row * rs + c);
    if (row < nrows && c + 4 < dk) v1 = *reinterpret_cast<const float4*>(base + (size_t)row * rs + c + 4);
    bf16x4 h0, l0, h1, l1;
    split4(v0.x, v0.y, v0.z, v0.w, h0, l0);
    split4(v1.x, v1.y, v1.z, v1.w, h1, l1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      fh[ks][e] = h0[e];
      fh[ks][4 + e] = h1[e];
      fl[ks][e] = l0[e];
      fl[ks][4 + e] = l1[e];
    }
  }
}

#define LTRX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// acc[r] = sum_c IMG[row(r,half)][c] * FIXED[l31][c]     (two accumulators to break the dependent chain)
template <int DKP>
__device__ __forceinline__ f32x16 rows_x_fixed(const KImg<DKP>& img, const bf16x8 (&fh)[DKP / 16], const bf16x8 (&fl)[DKP / 16]) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  f32x16 a0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 a1 = a0;
#pragma unroll
  for (int ks = 0; ks < DKP / 16; ++ks) {
    const int o = koff<DKP>(l31, 16 * ks + 8 * half);
    const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&img.hi[o]);
    const bf16x8 xl = *reinterpret_cast<const bf16x8*>(&img.lo[o]);
    if (ks & 1) {
      a1 = LTRX_MFMA(xl, fh[ks], a1);
      a1 = LTRX_MFMA(xh, fl[ks], a1);
      a1 = LTRX_MFMA(xh, fh[ks], a1);
    } else {
      a0 = LTRX_MFMA(xl, fh[ks], a0);
      a0 = LTRX_MFMA(xh, fl[ks], a0);
      a0 = LTRX_MFMA(xh, fh[ks], a0);
    }
  }
  return a0 + a1;
}

// out[ct][r'] += sum_row TIMG[32 ct + l31][row] * p[row]   (p in D layout: register r <-> tile row rowmap(r, half))
template <int DKP>
__device__ __forceinline__ void cols_x_p(const TImg<DKP>& img, const f32x16& p, f32x16 (&out)[DKP / 32]) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  bf16x8 ph[2], pl[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = p[8 * u + e];
      const __bf16 h = (__bf16)x;
      ph[u][e] = h;
      pl[u][e] = (__bf16)(x - (float)h);
    }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ct = 0; ct < DKP / 32; ++ct) {
      const int o = toff(32 * ct + l31, 16 * u + 8 * half);
      const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&img.hi[o]);
      const bf16x8 xl = *reinterpret_cast<const bf16x8*>(&img.lo[o]);
      out[ct] = LTRX_MFMA(xl, ph[u], out[ct]);
      out[ct] = LTRX_MFMA(xh, pl[u], out[ct]);
      out[ct] = LTRX_MFMA(xh, ph[u], out[ct]);
    }
}

template <int DKP>
__device__ __forceinline__ void store_rows(float* __restrict__ base, int row0, int nrows, int dk, size_t rs,
                                           const f32x16 (&out)[DKP / 32], float scale) {
  const int lane = threadIdx.x & 63;
  const int row = row0 + (lane & 31);
  if (row >= nrows) return;
  float* rp = base + (size_t)row * rs;
#pragma unroll
  for (int ct = 0; ct < DKP / 32; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * ct + 8 * g + 4 * (lane >> 5);
      if (c < dk)
        *reinterpret_cast<float4*>(rp + c) = make_float4(out[ct][4 * g + 0] * scale, out[ct][4 * g + 1] * scale,
                                                         out[ct][4 * g + 2] * scale, out[ct][4 * g + 3] * scale);
    }
}

template <int DKP>
__device__ __forceinline__ void zero_acc(f32x16 (&o)[DKP / 32]) {
#pragma unroll
  for (int ct = 0; ct < DKP / 32; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// forward: workgroup = 4 waves = 128 queries of one (slate, head); 32-key tiles
// ------------------------------------------------------------------------------------------------------------------
template <int DKP>
__global__ void __launch_bounds__(256) ltrx_mha_fwd_bf16_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const uint8_t* __restrict__ kpm,
                                                                int L, int h, int dk, int rs, float* __restrict__ o, int ors,
                                                                float* __restrict__ lse, float scale) {
  __shared__ __attribute__((aligned(16))) KImg<DKP> kimg;
  __shared__ __attribute__((aligned(16))) TImg<DKP> vimg;
  __shared__ float kmask[32];
  const int b = blockIdx.y / h, head = blockIdx.y % h;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const size_t slate = (size_t)b * L;
  const float* kb = k + slate * rs + (size_t)head * dk;
  const float* vb = v + slate * rs + (size_t)head * dk;
  bf16x8 qh[DKP / 16], ql[DKP / 16];
  load_fixed<DKP>(qh, ql, q + slate * rs + (size_t)head * dk, q0, L, dk, rs);
  f32x16 oacc[DKP / 32];
  zero_acc<DKP>(oacc);
  float m = -INFINITY, l = 0.f;
  const int nkt = (L + 31) / 32;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    stage_k<DKP>(kimg, kb, kt * 32, L, dk, rs);
    stage_t<DKP>(vimg, vb, kt * 32, L, dk, rs);
    if (threadIdx.x < 32) {
      const int key = kt * 32 + threadIdx.x;
      kmask[threadIdx.x] = (key >= L || kpm[slate + key]) ? 1.f : 0.f;
    }
    __syncthreads();
    f32x16 s = rows_x_fixed<DKP>(kimg, qh, ql);
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = (kmask[rowmap(r, half)] != 0.f) ? -INFINITY : s[r] * scale;
      mt = fmaxf(mt, s[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float alpha = (m == -INFINITY) ? 0.f : expf(m - mn);
    float ps = 0.f;
    f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = (s[r] == -INFINITY) ? 0.f : expf(s[r] - mn);
      ps += p[r];
    }
    l = l * alpha + ps;
#pragma unroll
    for (int ct = 0; ct < DKP / 32; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[ct][r] *= alpha;
    cols_x_p<DKP>(vimg, p, oacc);
    m = mn;
  }
  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = (lt > 0.f) ? 1.0f / lt : 0.f;
  store_rows<DKP>(o + slate * ors + (size_t)head * dk, q0, L, dk, ors, oacc, inv);
  const int qrow = q0 + (lane & 31);
  if (half == 0 && qrow < L) lse[((size_t)b * h + head) * L + qrow] = (lt > 0.f) ? m + logf(lt) : 0.f;
}

// ------------------------------------------------------------------------------------------------------------------
// backward dQ (+ delta): wave owns 32 queries, streams key tiles.  K is staged in both layouts, V K-style.
// ------------------------------------------------------------------------------------------------------------------
template <int DKP>
__global__ void __launch_bounds__(256) ltrx_mha_bwd_dq_bf16_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const uint8_t* __restrict__ kpm,
    const float* __restrict__ o, const float* __restrict__ dout, const float* __restrict__ lse, float* __restrict__ delta,
    int L, int h, int dk, int rs, int ors, float* __restrict__ dq, int drs, float scale) {
  __shared__ __attribute__((aligned(16))) KImg<DKP> kimg;
  __shared__ __attribute__((aligned(16))) KImg<DKP> vimg;
  __shared__ __attribute__((aligned(16))) TImg<DKP> ktimg;
  __shared__ float kmask[32];
  const int b = blockIdx.y / h, head = blockIdx.y % h;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const size_t slate = (size_t)b * L;
  const float* kb = k + slate * rs + (size_t)head * dk;
  const float* vb = v + slate * rs + (size_t)head * dk;
  const int qrow = q0 + (lane & 31);
  const size_t stat = ((size_t)b * h + head) * L + qrow;
  const float lse_q = (qrow < L) ? lse[stat] : 0.f;
  // delta_q = <dO_q, O_q> in full fp32 (each half-wave covers half of the head dimension)
  float del_q = 0.f;
  if (qrow < L) {
    const float* op = o + (slate + qrow) * ors + (size_t)head * dk;
    const float* dp = dout + (slate + qrow) * ors + (size_t)head * dk;
    const int c0 = half * (dk / 2 / 4 * 4), c1 = half ? dk : (dk / 2 / 4 * 4);
    for (int c = c0; c < c1; c += 4) {
      const float4 a = *reinterpret_cast<const float4*>(op + c);
      const float4 g = *reinterpret_cast<const float4*>(dp + c);
      del_q += a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
    }
  }
  del_q += __shfl_xor(del_q, 32, 64);
  if (half == 0 && qrow < L) delta[stat] = del_q;
  bf16x8 qh[DKP / 16], ql[DKP / 16], doh[DKP / 16], dol[DKP / 16];
  load_fixed<DKP>(qh, ql, q + slate * rs + (size_t)head * dk, q0, L, dk, rs);
  load_fixed<DKP>(doh, dol, dout + slate * ors + (size_t)head * dk, q0, L, dk, ors);
  f32x16 dqacc[DKP / 32];
  zero_acc<DKP>(dqacc);
  const int nkt = (L + 31) / 32;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    stage_k<DKP>(kimg, kb, kt * 32, L, dk, rs);
    stage_k<DKP>(vimg, vb, kt * 32, L, dk, rs);
    stage_t<DKP>(ktimg, kb, kt * 32, L, dk, rs);
    if (threadIdx.x < 32) {
      const int key = kt * 32 + threadIdx.x;
      kmask[threadIdx.x] = (key >= L || kpm[slate + key]) ? 1.f : 0.f;
    }
    __syncthreads();
    const f32x16 s = rows_x_fixed<DKP>(kimg, qh, ql);
    const f32x16 dp = rows_x_fixed<DKP>(vimg, doh, dol);
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (kmask[rowmap(r, half)] != 0.f) ? 0.f : expf(s[r] * scale - lse_q);
      ds[r] = p * (dp[r] - del_q) * scale;
    }
    cols_x_p<DKP>(ktimg, ds, dqacc);
  }
  store_rows<DKP>(dq + slate * drs + (size_t)head * dk, q0, L, dk, drs, dqacc, 1.0f);
}

// ------------------------------------------------------------------------------------------------------------------
// backward dK, dV: wave owns 32 keys, streams query tiles.  Q and dO are staged in both layouts.
// ------------------------------------------------------------------------------------------------------------------
template <int DKP>
__global__ void __launch_bounds__(256) ltrx_mha_bwd_dkdv_bf16_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const uint8_t* __restrict__ kpm,
    const float* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta, int L, int h, int dk,
    int rs, int ors, float* __restrict__ dkout, float* __restrict__ dvout, int drs, float scale) {
  __shared__ __attribute__((aligned(16))) KImg<DKP> qimg;
  __shared__ __attribute__((aligned(16))) KImg<DKP> doimg;
  __shared__ __attribute__((aligned(16))) TImg<DKP> qtimg;
  __shared__ __attribute__((aligned(16))) TImg<DKP> dotimg;
  __shared__ float lse_t[32];
  __shared__ float del_t[32];
  const int b = blockIdx.y / h, head = blockIdx.y % h;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const size_t slate = (size_t)b * L;
  const float* qb = q + slate * rs + (size_t)head * dk;
  const float* dob = dout + slate * ors + (size_t)head * dk;
  bf16x8 kh[DKP / 16], kl[DKP / 16], vh[DKP / 16], vl[DKP / 16];
  load_fixed<DKP>(kh, kl, k + slate * rs + (size_t)head * dk, k0, L, dk, rs);
  load_fixed<DKP>(vh, vl, v + slate * rs + (size_t)head * dk, k0, L, dk, rs);
  const int key = k0 + (lane & 31);
  const bool key_masked = (key >= L) || (kpm[slate + (key < L ? key : 0)] != 0);
  f32x16 dkacc[DKP / 32], dvacc[DKP / 32];
  zero_acc<DKP>(dkacc);
  zero_acc<DKP>(dvacc);
  const size_t statb = ((size_t)b * h + head) * L;
  const int nqt = (L + 31) / 32;
  for (int qt = 0; qt < nqt; ++qt) {
    __syncthreads();
    stage_k<DKP>(qimg, qb, qt * 32, L, dk, rs);
    stage_k<DKP>(doimg, dob, qt * 32, L, dk, ors);
    stage_t<DKP>(qtimg, qb, qt * 32, L, dk, rs);
    stage_t<DKP>(dotimg, dob, qt * 32, L, dk, ors);
    if (threadIdx.x < 32) {
      const int qrow = qt * 32 + threadIdx.x;
      lse_t[threadIdx.x] = (qrow < L) ? lse[statb + qrow] : INFINITY;
      del_t[threadIdx.x] = (qrow < L) ? delta[statb + qrow] : 0.f;
    }
    __syncthreads();
    const f32x16 s = rows_x_fixed<DKP>(qimg, kh, kl);          // S[q = row(r,half)][key = l31]
    f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = key_masked ? 0.f : expf(s[r] * scale - lse_t[rowmap(r, half)]);
    cols_x_p<DKP>(dotimg, p, dvacc);                            // dV^T[c][key] += sum_q dO[q][c] P[q][key]
    const f32x16 dp = rows_x_fixed<DKP>(doimg, vh, vl);         // dP[q][key]
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = p[r] * (dp[r] - del_t[rowmap(r, half)]) * scale;
    cols_x_p<DKP>(qtimg, ds, dkacc);                            // dK^T[c][key] += sum_q Q[q][c] dS[q][key]
  }
  store_rows<DKP>(dkout + slate * drs + (size_t)head * dk, k0, L, dk, drs, dkacc, 1.0f);
  store_rows<DKP>(dvout + slate * drs + (size_t)head * dk, k0, L, dk, drs, dvacc, 1.0f);
}

// ------------------------------------------------------------------------------------------------------------------
// host launchers (called from ltrx_mha.hip's C entry points when the split-bf16 mode is selected)
// ------------------------------------------------------------------------------------------------------------------
#define LTRX_BF16_DKP_DISPATCH(dk, CALL) \
  do {                                   \
    if ((dk) <= 32) { CALL(32); }        \
    else if ((dk) <= 64) { CALL(64); }   \
    else { CALL(128); }                  \
  } while (0)

int ltrx_mha_fwd_bf16_launch(const float* q, const float* k, const float* v, const uint8_t* kpm, int B, int L, int h, int dk,
                             int rs, float* o, int ors, float* lse, hipStream_t s) {
  const dim3 grid((L + 127) / 128, B * h);
  const float scale = 1.0f / sqrtf((float)dk);
#define CALL(DKP) hipLaunchKernelGGL(ltrx_mha_fwd_bf16_kernel<DKP>, grid, dim3(256), 0, s, q, k, v, kpm, L, h, dk, rs, o, ors, lse, scale)
  LTRX_BF16_DKP_DISPATCH(dk, CALL);
#undef CALL
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

int ltrx_mha_bwd_bf16_launch(const float* q, const float* k, const float* v, const uint8_t* kpm, const float* o,
                             const float* dout, const float* lse, int B, int L, int h, int dk, int rs, int ors, float* dq,
                             float* dkk, float* dv, int drs, float* delta, hipStream_t s) {
  const dim3 grid((L + 127) / 128, B * h);
  const float scale = 1.0f / sqrtf((float)dk);
#define CALLQ(DKP) hipLaunchKernelGGL(ltrx_mha_bwd_dq_bf16_kernel<DKP>, grid, dim3(256), 0, s, q, k, v, kpm, o, dout, lse, delta, L, h, dk, rs, ors, dq, drs, scale)
  LTRX_BF16_DKP_DISPATCH(dk, CALLQ);
#undef CALLQ
  LTRX_LAUNCH_CHECK();
#define CALLK(DKP) hipLaunchKernelGGL(ltrx_mha_bwd_dkdv_bf16_kernel<DKP>, grid, dim3(256), 0, s, q, k, v, kpm, dout, lse, delta, L, h, dk, rs, ors, dkk, dv, drs, scale)
  LTRX_BF16_DKP_DISPATCH(dk, CALLK);
#undef CALLK
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
