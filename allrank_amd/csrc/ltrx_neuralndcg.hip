// NeuralNDCG / NeuralNDCG-transposed (deterministic NeuralSort + Sinkhorn scaling), forward + backward.
// Reference: allrank/models/losses/neuralNDCG.py:10-136, allrank/models/losses/loss_utils.py:8-67.
//
// Math per slate (n = #valid items; everything below lives on the valid x valid block -- the padded block of
// the reference's L x L matrices decouples: it is 1/n_pad after the first normalisation, has zero adjoint, and
// the cross blocks are masked to 0 before and after Sinkhorn, loss_utils.py:17-18,28-29):
//   Bsum_j      = sum_k |s_j - s_k|                                              (loss_utils.py:49-52)
//   scaling_i   = n + 1 - 2(i+1) for row index i < n else 0                      (:54-57; i = ORIGINAL index of the row)
//   P0[i][j]    = softmax_j( (scaling_i * s_j - Bsum_j) / tau )                   (:59-66)
//   Sinkhorn    : repeat { P /= max(colsum, 1e-10); P /= max(rowsum, 1e-10) } up to max_iter times, stopping when
//                 the largest |rowsum-1|, |colsum-1| over the WHOLE BATCH drops below tol  (:20-26)
//   value_b     = sum_{i<k} disc_i sum_j P[i][j] g_j / (idcg_b + 1e-10),  0 if idcg_b == 0      (neuralNDCG.py:51-63)
//   loss        = - sum_b value_b / #{b : idcg_b != 0}                                         (:66-70)
// Backward: Sinkhorn is a chain of diagonal scalings, hence exactly reversible.  Only the clamped normaliser
// vectors of each step are kept (2 * iters * n floats per slate instead of the reference's 100 L x L autograd
// tensors); walking back, the state is recovered by multiplying the normaliser back in while the adjoint is
// propagated:   Y = X / c  =>  Xbar = (Ybar - [c > 1e-10] * sum(Ybar * Y)) / c,  X = Y * c.
//
// Batch-global early exit: the forward kernel always runs max_iter steps and records each slate's residual per
// step; a one-block kernel picks T* = first step at which every slate is below tol (else max_iter); the backward
// kernel first rewinds the state from max_iter to T* with the stored normalisers, then reads the loss out and
// runs the backward from T*.  At training shapes (L >= 40) fp32 never reaches 1e-6, so T* = max_iter and the
// rewind is empty.
//
// Two kernel families, same math and workspace contract.  GENERAL (any L <= 2048): one workgroup per slate; the n x n state S and
// adjoint A live in a global workspace (L2 / Infinity Cache resident: 2 * 230 KB per slate at L = 240); column phases: one thread
// per column, lanes read consecutive addresses of a row (coalesced); row phases: one wave per row, lane-strided, wave-shuffle sums.
// BLOCK-RESIDENT (L <= 240, the default there): state and adjoint-product in registers + LDS as a 2-D block decomposition -- see
// the section further down.
#include "ltrx_device.h"

// No FMA contraction in this file: the row max of P_max/tau and the exponent argument must be the SAME rounded
// number (the reference materialises P_max/tau, loss_utils.py:65-66); a contracted fma((..), 1/tau, -max) differs
// from the rounded product by one ulp of ~1e12 when tau is small and the scores are huge -> exp(+6e4) = inf.
#pragma clang fp contract(off)

using namespace ltrx;

namespace {
constexpr float kSinkEps = 1e-10f;   // DEFAULT_EPS in loss_utils.py:21-22

struct NeuralWs {
  float* per;      // [B]  value_b
  float* res;      // [B][max_iter]
  int* titer;      // [1]
  float* cn;       // [B][max_iter][L]
  float* rn;       // [B][max_iter][L]
  float* S;        // [B][L*L]
  float* A;        // [B][L*L]
};

__host__ __device__ inline size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

inline NeuralWs carve(void* ws, int B, int L, int max_iter) {
  NeuralWs w;
  char* p = (char*)ws;
  w.per = (float*)p;   p += align64((size_t)B * 4);
  w.res = (float*)p;   p += align64((size_t)B * max_iter * 4);
  w.titer = (int*)p;   p += 64;
  w.cn = (float*)p;    p += align64((size_t)B * (max_iter + 1) * L * 4);   // (+1: the block path also keeps the sums after the last step)
  w.rn = (float*)p;    p += align64((size_t)B * max_iter * L * 4);
  w.S = (float*)p;     p += align64((size_t)B * L * L * 4);
  w.A = (float*)p;
  return w;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// step 1: ideal DCG@k per slate (metrics.dcg(y_true, y_true, ats=[k]) as called at neuralNDCG.py:56/:120)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ltrx_neural_idcg_kernel(const float* __restrict__ y_true, int L, float pad,
                                                               int k, int powered, float* __restrict__ idcg_out,
                                                               float* __restrict__ nz_ws) {
  extern __shared__ float lds[];
  float* ys = lds;
  __shared__ float red[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  const float* yp = y_true + (size_t)b * L;
  for (int i = threadIdx.x; i < L; i += blockDim.x) ys[i] = yp[i];
  __syncthreads();
  const int kk = (k <= 0 || k > L) ? L : k;
  float acc = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float yi = ys[i];
    if (yi == pad) continue;
    int ry = 0;
    for (int j = 0; j < L; ++j) {
      const float yj = ys[j];
      ry += (yj != pad) && ((yj > yi) || (yj == yi && j < i));
    }
    if (ry < kk) acc += (powered ? exp2f(yi) - 1.0f : yi) / log2f((float)ry + 2.0f);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) {
    idcg_out[b] = acc;
    nz_ws[b] = (acc != 0.f) ? 1.0f : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// shared prologue of the forward/backward kernels: compaction of the valid items into LDS
// ---------------------------------------------------------------------------------------------------------
struct SlateLds {
  float* sc;    // [L] compacted scores
  float* gc;    // [L] compacted gains
  float* bs;    // [L] Bsum
  float* scal;  // [L] scaling of row r (by ORIGINAL index of that row)
  float* dk;    // [L] discount of row r (0 beyond k)
  float* q;     // [L] scratch
  int* vidx;    // [L] original index of compacted item
};

__device__ __forceinline__ SlateLds carve_lds(float* lds, int L) {
  SlateLds s;
  s.sc = lds;
  s.gc = lds + L;
  s.bs = lds + 2 * L;
  s.scal = lds + 3 * L;
  s.dk = lds + 4 * L;
  s.q = lds + 5 * L;
  s.vidx = (int*)(lds + 6 * L);
  return s;
}
#define LTRX_NEURAL_LDS_FLOATS(L) (7 * (size_t)(L))

// returns n (number of valid items); fills the LDS tables.  gain_mode: 0 = y, 1 = 2^y - 1.
__device__ __forceinline__ int load_slate(const SlateLds& t, const float* __restrict__ sp, const float* __restrict__ yp,
                                          int L, float pad, int k, int gain_powered, int* redi) {
  int cnt = 0;
  for (int i = threadIdx.x; i < L; i += blockDim.x) cnt += (yp[i] != pad);
  const int n = block_sum_i(cnt, redi);
  const int kk = k;                      // rows (ranks) >= kk carry no discount; the callers resolve k=None / k_rows
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float y = yp[i];
    if (y == pad) continue;
    int c = 0;
    for (int j = 0; j < i; ++j) c += (yp[j] != pad);
    t.vidx[c] = i;
    t.sc[c] = sp[i];
    t.gc[c] = gain_powered ? exp2f(y) - 1.0f : y;
    // row c of the compacted matrix is the reference's row i (its mask and scaling are indexed by i)
    t.scal[c] = (i < n) ? (float)(n + 1 - 2 * (i + 1)) : 0.f;
    t.dk[c] = (i < kk) ? 1.0f / log2f((float)i + 2.0f) : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float sj = t.sc[j];
    float a = 0.f;
    for (int m = 0; m < n; ++m) a += fabsf(sj - t.sc[m]);
    t.bs[j] = a;
  }
  __syncthreads();
  return n;
}

// ---------------------------------------------------------------------------------------------------------
// step 2a: NeuralSort + max_iter Sinkhorn steps, recording normalisers and residuals
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) ltrx_neural_forward_kernel(const float* __restrict__ y_pred,
                                                                   const float* __restrict__ y_true, int L, float pad,
                                                                   float tau, int max_iter,
                                                                   float* __restrict__ Sws, float* __restrict__ cnws,
                                                                   float* __restrict__ rnws, float* __restrict__ resws) {
  extern __shared__ float lds[];
  __shared__ float red[LTRX_MAX_WAVES];
  __shared__ int redi[LTRX_MAX_WAVES];
  const SlateLds t = carve_lds(lds, L);
  const int b = blockIdx.x;
  const int n = load_slate(t, y_pred + (size_t)b * L, y_true + (size_t)b * L, L, pad, 0, 0, redi);
  float* S = Sws + (size_t)b * L * L;
  float* cn = cnws + (size_t)b * max_iter * L;
  float* rn = rnws + (size_t)b * max_iter * L;
  float* res = resws + (size_t)b * max_iter;
  const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;
  if (n == 0) {
    for (int it = threadIdx.x; it < max_iter; it += blockDim.x) res[it] = 0.f;
    return;
  }
  // ---- P0 = row softmax (one wave per row) ----
  for (int i = wave; i < n; i += nw) {
    const float sc_i = t.scal[i];
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, (sc_i * t.sc[j] - t.bs[j]) / tau);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) {
      const float e = expf((sc_i * t.sc[j] - t.bs[j]) / tau - m);
      S[(size_t)i * n + j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < n; j += 64) S[(size_t)i * n + j] /= sum;
  }
  __syncthreads();
  // ---- Sinkhorn ----
  float rowres_prev = 0.f;
  for (int it = 0; it <= max_iter; ++it) {
    // column sums of the current state (thread per column; consecutive lanes -> consecutive addresses)
    float cres = 0.f;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      float c = 0.f;
      for (int i = 0; i < n; ++i) c += S[(size_t)i * n + j];
      cres = fmaxf(cres, fabsf(c - 1.0f));
      if (it < max_iter) {
        c = fmaxf(c, kSinkEps);
        cn[(size_t)it * L + j] = c;
        for (int i = 0; i < n; ++i) S[(size_t)i * n + j] /= c;
      }
    }
    if (it > 0) {   // residual of step it-1: its row sums (after the row scaling) and these column sums
      cres = block_max(cres, red);
      if (threadIdx.x == 0) res[it - 1] = fmaxf(cres, rowres_prev);
    }
    if (it == max_iter) break;
    __syncthreads();
    float rres = 0.f;
    for (int i = wave; i < n; i += nw) {
      float r = 0.f;
      for (int j = lane; j < n; j += 64) r += S[(size_t)i * n + j];
      r = fmaxf(wave_sum(r), kSinkEps);
      float r2 = 0.f;
      for (int j = lane; j < n; j += 64) {
        const float v = S[(size_t)i * n + j] / r;
        S[(size_t)i * n + j] = v;
        r2 += v;
      }
      r2 = wave_sum(r2);
      if (lane == 0) rn[(size_t)it * L + i] = r;
      rres = fmaxf(rres, fabsf(r2 - 1.0f));
    }
    rowres_prev = block_max(rres, red);   // (barriers: publishes the row-scaled S)
  }
}

// T* = 1 + first step whose residual is below tol for every slate, else max_iter (loss_utils.py:20-26)
__global__ void __launch_bounds__(256) ltrx_neural_pick_iter_kernel(const float* __restrict__ res, int B, int max_iter,
                                                                    float tol, int* __restrict__ titer,
                                                                    int32_t* __restrict__ iters_out) {
  __shared__ float red[LTRX_MAX_WAVES];
  int T = max_iter;
  for (int it = 0; it < max_iter; ++it) {
    float m = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) m = fmaxf(m, res[(size_t)b * max_iter + it]);
    m = block_max(m, red);
    if (m < tol) {
      T = it + 1;
      break;   // m is uniform across the block
    }
  }
  if (threadIdx.x == 0) {
    titer[0] = T;
    if (iters_out) iters_out[0] = T;
  }
}

// ---------------------------------------------------------------------------------------------------------
// step 2b: rewind to T*, read the value out, backward
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) ltrx_neural_backward_kernel(
    const float* __restrict__ y_pred, const float* __restrict__ y_true, const float* __restrict__ idcg,
    const float* __restrict__ nonzero_count, int L, float pad, float inv_tau, int gain_powered, int k,
    const int* __restrict__ k_rows, int max_iter,
    const int* __restrict__ titer, float* __restrict__ Sws, float* __restrict__ Aws, const float* __restrict__ cnws,
    const float* __restrict__ rnws, float* __restrict__ per_ws, float* __restrict__ per_out, float* __restrict__ grad) {
  extern __shared__ float lds[];
  __shared__ float red[LTRX_MAX_WAVES];
  __shared__ int redi[LTRX_MAX_WAVES];
  const SlateLds t = carve_lds(lds, L);
  const int b = blockIdx.x;
  int kk = (k <= 0 || k > L) ? L : k;
  if (k_rows) kk = min(kk, k_rows[b]);
  const int n = load_slate(t, y_pred + (size_t)b * L, y_true + (size_t)b * L, L, pad, kk, gain_powered, redi);
  float* S = Sws + (size_t)b * L * L;
  float* A = Aws + (size_t)b * L * L;
  const float* cn = cnws + (size_t)b * max_iter * L;
  const float* rn = rnws + (size_t)b * max_iter * L;
  const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;
  const int T = titer[0];
  float* gp = grad ? grad + (size_t)b * L : nullptr;
  if (gp)
    for (int i = threadIdx.x; i < L; i += blockDim.x) gp[i] = 0.f;
  const float id = idcg[b];
  const float cnt = nonzero_count[0];
  if (n == 0 || id == 0.f) {   // neuralNDCG.py:62-63: excluded from the mean, zero gradient
    if (threadIdx.x == 0) {
      per_ws[b] = 0.f;
      if (per_out) per_out[b] = 0.f;
    }
    return;
  }
  // ---- rewind the extra steps max_iter-1 .. T (undo row scaling, then column scaling) ----
  for (int it = max_iter - 1; it >= T; --it) {
    for (int i = wave; i < n; i += nw) {
      const float r = rn[(size_t)it * L + i];
      for (int j = lane; j < n; j += 64) S[(size_t)i * n + j] = (S[(size_t)i * n + j] * r) * cn[(size_t)it * L + j];
    }
  }
  __syncthreads();
  // ---- read-out: value_b = sum_i dk_i sum_j S[i][j] g_j / (idcg + eps) ----
  float v = 0.f;
  for (int i = wave; i < n; i += nw) {
    const float d = t.dk[i];
    if (d == 0.f) continue;
    float a = 0.f;
    for (int j = lane; j < n; j += 64) a += S[(size_t)i * n + j] * t.gc[j];
    a = wave_sum(a);
    if (lane == 0) v += d * a;
  }
  v = block_sum(v, red);
  const float value = v / (id + kSinkEps);
  if (threadIdx.x == 0) {
    per_ws[b] = value;
    if (per_out) per_out[b] = value;
  }
  if (!gp) return;
  // ---- adjoint seed: d loss / d S[i][j] = -(1/cnt) dk_i g_j / (idcg + eps) ----
  const float coef = -1.0f / (cnt * (id + kSinkEps));
  for (int i = wave; i < n; i += nw) {
    const float d = coef * t.dk[i];
    for (int j = lane; j < n; j += 64) A[(size_t)i * n + j] = d * t.gc[j];
  }
  __syncthreads();
  // ---- reverse Sinkhorn ----
  for (int it = T - 1; it >= 0; --it) {
    // row step:  Y2 = Y1 / r
    for (int i = wave; i < n; i += nw) {
      const float r = rn[(size_t)it * L + i];
      float d = 0.f;
      for (int j = lane; j < n; j += 64) d += A[(size_t)i * n + j] * S[(size_t)i * n + j];
      d = (r > kSinkEps) ? wave_sum(d) : 0.f;
      for (int j = lane; j < n; j += 64) {
        const size_t o = (size_t)i * n + j;
        A[o] = (A[o] - d) / r;
        S[o] = S[o] * r;
      }
    }
    __syncthreads();
    // column step:  Y1 = X / c
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const float c = cn[(size_t)it * L + j];
      float d = 0.f;
      for (int i = 0; i < n; ++i) d += A[(size_t)i * n + j] * S[(size_t)i * n + j];
      if (!(c > kSinkEps)) d = 0.f;
      for (int i = 0; i < n; ++i) {
        const size_t o = (size_t)i * n + j;
        A[o] = (A[o] - d) / c;
        S[o] = S[o] * c;
      }
    }
    __syncthreads();
  }
  // ---- row softmax backward: gz = P0 * (g0 - <g0, P0>) / tau   (S now holds P0 again) ----
  for (int i = wave; i < n; i += nw) {
    float d = 0.f;
    for (int j = lane; j < n; j += 64) d += A[(size_t)i * n + j] * S[(size_t)i * n + j];
    d = wave_sum(d);
    for (int j = lane; j < n; j += 64) {
      const size_t o = (size_t)i * n + j;
      A[o] = S[o] * (A[o] - d) * inv_tau;
    }
  }
  __syncthreads();
  // ---- P_max[i][j] = scaling_i s_j - Bsum_j:  column sums of gz ----
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int i = 0; i < n; ++i) {
      const float g = A[(size_t)i * n + j];
      a += g * t.scal[i];
      q += g;
    }
    t.bs[j] = a;   // Bsum is no longer needed: reuse as the direct term
    t.q[j] = q;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float sj = t.sc[j];
    float sgn_sum = 0.f, cross = 0.f;
    for (int m = 0; m < n; ++m) {
      const float dlt = sj - t.sc[m];
      const float sg = (dlt > 0.f) ? 1.0f : ((dlt < 0.f) ? -1.0f : 0.f);
      sgn_sum += sg;
      cross -= t.q[m] * sg;     // sign(s_m - s_j) = -sign(s_j - s_m)
    }
    gp[t.vidx[j]] = t.bs[j] - t.q[j] * sgn_sum + cross;
  }
}


// ---------------------------------------------------------------------------------------------------------
// Block-resident path (round 3; L <= 240): the n x n Sinkhorn state in VGPRs as a 2-D BLOCK decomposition.
// (Round 2 kept the state in registers too, but gave a wave whole rows (lane = column): every row sum was a 6-step DPP chain (15 per
// wave and iteration, serialised to stay under 128 VGPRs), every column sum went through a 16 KB LDS exchange that 768 of the 1024
// threads waited for, and the batch-global residual cost two more block reductions -- 6 barriers and ~18 k cycles per iteration for
// 57.6 k elements: forward 459 us, backward 1173 us at 256 x 240, profiles/r03_bench_attn_neuralndcg_kernel_stats.md "before";
// same-box A/B of the whole plugin call 2074 -> 686 us, profiles/r03_neuralndcg_ab.md.)
// Here the 16 waves form a 4 x 4 grid of blocks and the 64 lanes of a wave an LRN x LCN grid of TBR x TBC register tiles: thread
// (wr, wc, lr, lc) owns rows LRN TBR wr + TBR lr + (0..TBR-1) and columns LCN TBC wc + TBC lc + (0..TBC-1).  Geometries: 8 x 8
// lanes with square tiles of 2 / 4 / 6 on 4 x 4 waves (L <= 64 / 128 / 192); 4 x 16 lanes with 15 x 5 tiles on 4 x 3 waves (768
// threads, 170 registers each) for L <= 240: an exact fit of the WEB30K slate length.
//   * a row sum = in-thread adds, a DPP all-reduce over the LCN consecutive lanes lc, and 4 wave partials through LDS; a column
//     sum the same over the LRN lanes lr (stride LCN: row_ror:8 / xor 16 / xor 32 exchanges);
//   * every thread adds the four wave partials of its own rows / columns in the SAME fixed order (bitwise identical normalisers
//     in every thread, deterministic) and takes the reciprocal itself: no combine phase, two barriers per iteration;
//   * the residual trace of the batch-global early exit (loss_utils.py:25) is NOT computed in the loop: the normaliser trace that
//     the backward needs anyway (column sums -- stored unclamped, plus the sums after the last step -- and clamped row sums)
//     determines it, and ltrx_neural_residual_kernel derives res[b][it] from it afterwards.
// Same arithmetic per element as the kernels above (x * (1/c), clamp at 1e-10, softmax as exp(z - max) / sum); only the order of
// the partial sums inside a row / column differs.  Workspace: cn has max_iter + 1 rows per slate on this path.
// ---------------------------------------------------------------------------------------------------------
// (1 / c is an IEEE division, ~10 instructions: the hardware reciprocal v_rcp_f32 halves the VALU count of a forward iteration
//  (700 -> 360 per wave) but only buys 5 % of the call -- the loop is bound by barrier / DPP / LDS latency, not by VALU issue --
//  and costs another ulp per normalisation; measured and not adopted, profiles/NOTES.md round 3)
#define LTRX_NEURAL_RCP(x) (1.0f / (x))
template <int LCN>
__device__ __forceinline__ float sum_lc(float v) {       // all-reduce over the LCN consecutive lanes that share lr
  v += LTRX_DPP_F(0.f, v, 0xB1, 0xF, true);               // quad_perm [1,0,3,2]
  v += LTRX_DPP_F(0.f, v, 0x4E, 0xF, true);               // quad_perm [2,3,0,1]
  v += LTRX_DPP_F(0.f, v, 0x141, 0xF, true);              // row_half_mirror: the other quad of the 8-lane group
  if (LCN == 16) v += LTRX_DPP_F(0.f, v, 0x140, 0xF, true);   // row_mirror: the other half of the 16-lane row
  return v;
}
template <int LCN>
__device__ __forceinline__ float max_lc(float v) {
  v = fmaxf(v, LTRX_DPP_F(v, v, 0xB1, 0xF, false));
  v = fmaxf(v, LTRX_DPP_F(v, v, 0x4E, 0xF, false));
  v = fmaxf(v, LTRX_DPP_F(v, v, 0x141, 0xF, false));
  if (LCN == 16) v = fmaxf(v, LTRX_DPP_F(v, v, 0x140, 0xF, false));
  return v;
}
template <int LCN>
__device__ __forceinline__ float sum_lr(float v) {       // all-reduce over the 64 / LCN lanes lc, lc + LCN, ... (same lc)
  if (LCN == 8) v += LTRX_DPP_F(0.f, v, 0x128, 0xF, true);    // row_ror:8 = lane ^ 8 inside a 16-lane row
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <int LRN, int TBR, int TBC, int NWC>
struct BlkGeom {
  static constexpr int LCN = 64 / LRN;
  static constexpr int WR = 4 * LRN * TBR;      // rows covered (4 wave rows)
  static constexpr int WC = NWC * LCN * TBC;    // columns covered (NWC wave columns)
  int wr, wc, lr, lc, r0, c0;
  __device__ __forceinline__ BlkGeom() {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    wr = wave / NWC;
    wc = wave % NWC;
    lr = lane / LCN;
    lc = lane % LCN;
    r0 = LRN * TBR * wr + TBR * lr;
    c0 = LCN * TBC * wc + TBC * lc;
  }
};
#define LTRX_PART4(part, idx) (((part[0][idx] + part[1][idx]) + part[2][idx]) + part[3][idx])     /* fixed order */
// the NWC wave partials of a row (fixed order)
template <int NWC, int W>
__device__ __forceinline__ float part_rows(const float (*part)[W], int idx) {
  float a = part[0][idx] + part[1][idx];
  if (NWC >= 3) a += part[2][idx];
  if (NWC == 4) a += part[3][idx];
  return a;
}
template <int NWC, int W>
__device__ __forceinline__ float part_rows_max(const float (*part)[W], int idx) {
  float a = fmaxf(part[0][idx], part[1][idx]);
  if (NWC >= 3) a = fmaxf(a, part[2][idx]);
  if (NWC == 4) a = fmaxf(a, part[3][idx]);
  return a;
}

#ifdef LTRX_NEURAL_STAMP     // lab builds only (tools/lab/lib_variant.sh): cycle stamps of one workgroup of the forward kernel
__device__ unsigned long long g_neural_stamps[12][8][8];
#define NSTAMP(it, ph)                                                                 \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if (blockIdx.x == LTRX_NEURAL_STAMP && (threadIdx.x & 63) == 0 && (it) >= 20 && (it) < 28)    \
      g_neural_stamps[threadIdx.x >> 6][(it) - 20][ph] = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
extern "C" int ltrx_debug_neural_stamps(unsigned long long* host_dst) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_neural_stamps), sizeof(unsigned long long) * 12 * 8 * 8) == hipSuccess ? 0 : 1;
}
#else
#define NSTAMP(it, ph)
#endif

template <int LRN, int TBR, int TBC, int NWC>
__global__ void __launch_bounds__(256 * NWC) ltrx_neural_forward_blk_kernel(const float* __restrict__ y_pred,
                                                                       const float* __restrict__ y_true, int L, float pad,
                                                                       float tau, int max_iter, float* __restrict__ Sws,
                                                                       float* __restrict__ cnws, float* __restrict__ rnws) {
  typedef BlkGeom<LRN, TBR, TBC, NWC> G;
  constexpr int LCN = G::LCN, W = (G::WC > G::WR ? G::WC : G::WR);
  extern __shared__ float lds[];
  __shared__ int redi[LTRX_MAX_WAVES];
  __shared__ __attribute__((aligned(16))) float part_r[NWC][W];    // row partials: one per wave column
  __shared__ __attribute__((aligned(16))) float part_c[4][W];      // column partials: one per wave row (also the softmax sums)
  __shared__ float dump[256 * NWC];                                // where the lanes that do not publish a partial sum write
  const SlateLds t = carve_lds(lds, L);
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = load_slate(t, y_pred + (size_t)b * L, y_true + (size_t)b * L, L, pad, 0, 0, redi);
  if (n == 0) return;                                   // (the residual kernel writes res = 0 for an empty slate)
  float* S = Sws + (size_t)b * L * L;
  float* cn = cnws + (size_t)b * (max_iter + 1) * L;
  float* rn = rnws + (size_t)b * max_iter * L;
  const G g;
  // partial sums are published by the lanes lc == 0 (rows) / lr == 0 (columns); the others store to `dump`: a store whose ADDRESS
  // is selected keeps the loop free of exec-mask changes, so the 15 (5) independent reduction chains of a phase interleave
  // (cycle stamps, profiles/NOTES.md round 3: the row phase of the youngest wave of a SIMD 3100 -> see there)
  float* const rdst = (g.lc == 0) ? &part_r[g.wc][g.r0] : &dump[tid];
  float* const cdst = (g.lr == 0) ? &part_c[g.wr][g.c0] : &dump[tid];
  const int rstep = (g.lc == 0) ? 1 : 0, cstep = (g.lr == 0) ? 1 : 0;
  float m[TBR][TBC];
  // ---- P0 = row softmax of (scaling_i s_j - Bsum_j) / tau, two passes over the same expression (clamped table indices +
  //      selects: no branches around the reads) ----
  {
#pragma unroll
    for (int i = 0; i < TBR; ++i) {
      const float sc_i = t.scal[min(g.r0 + i, n - 1)];
      float a = -INFINITY;
#pragma unroll
      for (int j = 0; j < TBC; ++j) {
        const int jc = min(g.c0 + j, n - 1);
        const float z0 = (sc_i * t.sc[jc] - t.bs[jc]) / tau;
        a = fmaxf(a, (g.r0 + i < n && g.c0 + j < n) ? z0 : -INFINITY);
      }
      a = max_lc<LCN>(a);
      if (g.lc == 0) part_r[g.wc][g.r0 + i] = a;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TBR; ++i) {
      const float sc_i = t.scal[min(g.r0 + i, n - 1)];
      const float mx = part_rows_max<NWC, W>(part_r, g.r0 + i);
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < TBC; ++j) {
        const int jc = min(g.c0 + j, n - 1);
        const float z0 = (sc_i * t.sc[jc] - t.bs[jc]) / tau;
        const float z = (g.r0 + i < n && g.c0 + j < n) ? z0 : -INFINITY;
        const float e = (z == -INFINITY) ? 0.f : expf(z - mx);
        m[i][j] = e;
        a += e;
      }
      a = sum_lc<LCN>(a);
      if (g.lc == 0) part_c[g.wc][g.r0 + i] = a;       // (second exchange buffer: part_r is still being read)
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TBR; ++i) {
      const float sum = part_rows<NWC, W>(part_c, g.r0 + i);
#pragma unroll
      for (int j = 0; j < TBC; ++j) m[i][j] = (sum > 0.f) ? m[i][j] / sum : 0.f;
    }
    __syncthreads();                                    // part_c is rewritten by the first column phase
  }
  // ---- Sinkhorn ----
  for (int it = 0; it <= max_iter; ++it) {
    NSTAMP(it, 0);
#pragma unroll
    for (int j = 0; j < TBC; ++j) {
      float a = m[0][j];
#pragma unroll
      for (int i = 1; i < TBR; ++i) a += m[i][j];
      cdst[j * cstep] = sum_lr<LCN>(a);
    }
    NSTAMP(it, 1);
    __syncthreads();
    NSTAMP(it, 2);
    if (tid < n) cn[(size_t)it * L + tid] = LTRX_PART4(part_c, tid);      // BEFORE the clamp: normaliser trace + residual trace
    if (it == max_iter) break;
#pragma unroll
    for (int j = 0; j < TBC; ++j) {
      const float rc = LTRX_NEURAL_RCP(fmaxf(LTRX_PART4(part_c, g.c0 + j), kSinkEps));
#pragma unroll
      for (int i = 0; i < TBR; ++i) m[i][j] *= rc;
    }
    NSTAMP(it, 3);
#pragma unroll
    for (int i = 0; i < TBR; ++i) {
      float a = m[i][0];
#pragma unroll
      for (int j = 1; j < TBC; ++j) a += m[i][j];
      rdst[i * rstep] = sum_lc<LCN>(a);
    }
    NSTAMP(it, 4);
    __syncthreads();
    NSTAMP(it, 5);
    if (tid >= 256 && tid - 256 < n) rn[(size_t)it * L + tid - 256] = fmaxf(part_rows<NWC, W>(part_r, tid - 256), kSinkEps);
#pragma unroll
    for (int i = 0; i < TBR; ++i) {
      const float rr = LTRX_NEURAL_RCP(fmaxf(part_rows<NWC, W>(part_r, g.r0 + i), kSinkEps));
#pragma unroll
      for (int j = 0; j < TBC; ++j) m[i][j] *= rr;
    }
    NSTAMP(it, 6);
  }
  // ---- publish the final state for the backward kernel ----
#pragma unroll
  for (int i = 0; i < TBR; ++i)
#pragma unroll
    for (int j = 0; j < TBC; ++j)
      if (g.r0 + i < n && g.c0 + j < n) S[(size_t)(g.r0 + i) * n + g.c0 + j] = m[i][j];
}

// Backward of the block path.  Two matrices have to stay on chip: the state m and, instead of the adjoint a itself, the
// elementwise product b = a (.) m, in which the reversed Sinkhorn needs no division and no product per dot:
//   row step  Y2 = Y1 / r:  d_i = [r_i > eps] sum_j b_ij;  b_ij <- b_ij - d_i m_ij;  m_ij <- m_ij r_i      (a1 (.) Y1 = (a2 - d) (.) Y2)
//   col step  Y1 = X / c :  d_j = [c_j > eps] sum_i b_ij;  b_ij <- b_ij - d_j m_ij;  m_ij <- m_ij c_j
//   softmax   gz_ij = (b_ij - P0_ij sum_j b_ij) / tau                                                     (= P0 (a - <a, P0>) / tau)
// Two tiles per thread do not fit the 128 registers of a 1024-thread workgroup at L > 192, so TLC of the TBC columns of b live
// in LDS there (b_lds[row i][tid][TLC]: consecutive threads -> consecutive banks; 120 KB at 15 x 4, TLC = 2).  Both sweeps of a step
// go row by row; the row step accumulates the column sums of the new b, the column step publishes its row sums per row, so
// every element of b is read and written once per half-step and the last sweep leaves exactly the row sums the softmax backward
// starts from.  Padded rows / columns use r = c = 1, d = 0.
template <int LRN, int TBR, int TBC, int NWC, int TLC>
__global__ void __launch_bounds__(256 * NWC) ltrx_neural_backward_blk_kernel(
    const float* __restrict__ y_pred, const float* __restrict__ y_true, const float* __restrict__ idcg,
    const float* __restrict__ nonzero_count, int L, float pad, float inv_tau, int gain_powered, int k,
    const int* __restrict__ k_rows, int max_iter, const int* __restrict__ titer, const float* __restrict__ Sws,
    const float* __restrict__ cnws, const float* __restrict__ rnws, float* __restrict__ per_ws, float* __restrict__ per_out,
    float* __restrict__ grad) {
  typedef BlkGeom<LRN, TBR, TBC, NWC> G;
  constexpr int LCN = G::LCN, W = (G::WC > G::WR ? G::WC : G::WR), NT = 256 * NWC;
  constexpr int TRC = TBC - TLC;                        // columns of b kept in registers
  __shared__ float tables[7 * W];
  __shared__ float red[LTRX_MAX_WAVES];
  __shared__ int redi[LTRX_MAX_WAVES];
  __shared__ __attribute__((aligned(16))) float part_r[4][W];      // row partials: NWC of the 4 slots (all 4 in the epilogue)
  __shared__ __attribute__((aligned(16))) float part_c[4][W];
  __shared__ float cvec[W];
  __shared__ float dvec[W];
  __shared__ float nrm[2][2][W];                        // [step parity][row normalisers | unclamped column sums] of a step
  __shared__ float dump[NT];                            // where the lanes that do not publish a partial sum write (branch-free stores)
  __shared__ __attribute__((aligned(16))) float b_lds[TBR * NT * (TLC > 0 ? TLC : 1)];
  const SlateLds t = carve_lds(tables, L);
  const int b = blockIdx.x, tid = threadIdx.x;
  int kk = (k <= 0 || k > L) ? L : k;
  if (k_rows) kk = min(kk, k_rows[b]);
  const int n = load_slate(t, y_pred + (size_t)b * L, y_true + (size_t)b * L, L, pad, kk, gain_powered, redi);
  const float* S = Sws + (size_t)b * L * L;
  const float* cn = cnws + (size_t)b * (max_iter + 1) * L;
  const float* rn = rnws + (size_t)b * max_iter * L;
  const int T = titer[0];
  float* gp = grad ? grad + (size_t)b * L : nullptr;
  if (gp)
    for (int i = tid; i < L; i += blockDim.x) gp[i] = 0.f;
  const float id = idcg[b];
  const float cnt = nonzero_count[0];
  if (n == 0 || id == 0.f) {                            // neuralNDCG.py:62-63: excluded from the mean, zero gradient
    if (tid == 0) {
      per_ws[b] = 0.f;
      if (per_out) per_out[b] = 0.f;
    }
    return;
  }
  const G g;
  // partial sums are published by the lanes lc == 0 (rows) / lr == 0 (columns); the others store to `dump`: a store whose
  // ADDRESS is selected keeps the sweeps free of branches (with a branch per row the compiler moves the state updates of all rows
  // into one block and doubles their register footprint)
  float* const rdst = (g.lc == 0) ? &part_r[g.wc][g.r0] : &dump[tid];
  const int rstep = (g.lc == 0) ? 1 : 0;
  float* const cdst = (g.lr == 0) ? &part_c[g.wr][g.c0] : &dump[tid];
  const int cstep = (g.lr == 0) ? 1 : 0;
  float m[TBR][TBC];
  float br[TBR][TRC > 0 ? TRC : 1];
#define LTRX_BL(i, jj) b_lds[((i) * NT + tid) * TLC + (jj)]
#define LTRX_B_GET(i, j) ((j) < TRC ? br[i][(j) < TRC ? (j) : 0] : LTRX_BL(i, (j) - TRC))
#define LTRX_B_SET(i, j, val)                                   \
  do {                                                          \
    if ((j) < TRC) br[i][(j) < TRC ? (j) : 0] = (val);          \
    else LTRX_BL(i, (j) - TRC) = (val);                         \
  } while (0)
#pragma unroll
  for (int i = 0; i < TBR; ++i)
#pragma unroll
    for (int j = 0; j < TBC; ++j) {
      const float sv = S[(size_t)min(g.r0 + i, n - 1) * n + min(g.c0 + j, n - 1)];      // clamped address + select: no branch
      m[i][j] = (g.r0 + i < n && g.c0 + j < n) ? sv : 0.f;
    }
  // ---- rewind the steps max_iter-1 .. T (batch-global early exit): undo the row scaling, then the column scaling ----
  for (int it = max_iter - 1; it >= T; --it) {
#pragma unroll
    for (int i = 0; i < TBR; ++i) {
      const float r = rn[(size_t)it * L + min(g.r0 + i, n - 1)];
#pragma unroll
      for (int j = 0; j < TBC; ++j) {
        const float c = fmaxf(cn[(size_t)it * L + min(g.c0 + j, n - 1)], kSinkEps);
        m[i][j] = (m[i][j] * r) * c;                     // (padded entries are 0 and stay 0)
      }
    }
  }
  // ---- read-out: value_b = sum_i dk_i sum_j m_ij g_j / (idcg + eps) ----
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < TBR; ++i) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < TBC; ++j) acc += m[i][j] * t.gc[min(g.c0 + j, n - 1)];          // (m is 0 on padded entries)
    v += ((g.r0 + i < n) ? t.dk[min(g.r0 + i, n - 1)] : 0.f) * acc;
  }
  v = block_sum(v, red);
  const float value = v / (id + kSinkEps);
  if (tid == 0) {
    per_ws[b] = value;
    if (per_out) per_out[b] = value;
  }
  if (!gp) return;
  const float coef = -1.0f / (cnt * (id + kSinkEps));
  // seed b = a (.) m and publish the row sums of b (per wave) that the first row step -- or, with T == 0, the softmax backward -- reduces
#pragma unroll
  for (int i = 0; i < TBR; ++i) {
    const float d = (g.r0 + i < n) ? coef * t.dk[min(g.r0 + i, n - 1)] : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < TBC; ++j) {
      const float a = d * ((g.c0 + j < n) ? t.gc[min(g.c0 + j, n - 1)] : 0.f);
      const float bv = a * m[i][j];
      LTRX_B_SET(i, j, bv);
      acc += bv;
    }
    acc = sum_lc<LCN>(acc);
    rdst[i * rstep] = acc;
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- reverse Sinkhorn ----
  for (int it = T - 1; it >= 0; --it) {
    // this step's normalisers -> LDS (one global load per thread, in flight until the barrier; padded rows / columns: 1);
    // double-buffered by step parity: a wave that is ahead writes the other buffer than the one a slower wave still reads
    float* rvec = nrm[it & 1][0];
    float* cnv = nrm[it & 1][1];
    if (tid < W && tid < 256) rvec[tid] = (tid < n) ? rn[(size_t)it * L + min(tid, n - 1)] : 1.0f;
    else if (tid >= 256 && tid - 256 < W) cnv[tid - 256] = (tid - 256 < n) ? cn[(size_t)it * L + min(tid - 256, n - 1)] : 1.0f;
    __syncthreads();                                    // row sums of b (part_r) and the normalisers are complete
    // row step: b_ij -= d_i m_ij; m_ij *= r_i; accumulate the column sums of the new b
    float colsum[TBC];
#pragma unroll
    for (int j = 0; j < TBC; ++j) colsum[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TBR; ++i) {
      const float r = rvec[g.r0 + i];
      float d = part_rows<NWC, W>(part_r, g.r0 + i);
      d = (r > kSinkEps) ? d : 0.f;
#pragma unroll
      for (int j = 0; j < TBC; ++j) {
        const float b1 = LTRX_B_GET(i, j) - d * m[i][j];
        LTRX_B_SET(i, j, b1);
        m[i][j] *= r;
        colsum[j] += b1;
      }
      __builtin_amdgcn_sched_barrier(0);                 // rows one after the other: no register room for several in flight
    }
#pragma unroll
    for (int j = 0; j < TBC; ++j) {
      cdst[j * cstep] = sum_lr<LCN>(colsum[j]);
    }
    __syncthreads();
    // column step: b_ij -= d_j m_ij; m_ij *= c_j; publish the row sums of the new b
    float dj[TBC], cj[TBC];
#pragma unroll
    for (int j = 0; j < TBC; ++j) {
      const float craw = cnv[g.c0 + j];
      const float d = LTRX_PART4(part_c, g.c0 + j);
      dj[j] = (craw > kSinkEps) ? d : 0.f;               // clamp active: no dependence on the column sum
      cj[j] = fmaxf(craw, kSinkEps);
    }
#pragma unroll
    for (int i = 0; i < TBR; ++i) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < TBC; ++j) {
        const float b1 = LTRX_B_GET(i, j) - dj[j] * m[i][j];
        LTRX_B_SET(i, j, b1);
        m[i][j] *= cj[j];
        acc += b1;
      }
      acc = sum_lc<LCN>(acc);
      rdst[i * rstep] = acc;                             // (part_r: its readers of this step are behind the barrier above)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- row softmax backward (m holds P0 again, part_r the row sums of b per wave): gz = (b - P0 sum_j b) / tau; its column sums
  //      weighted by the row scaling (direct term) and plain (Q) ----
  {
    __syncthreads();
    float c1[TBC], c2[TBC];
#pragma unroll
    for (int j = 0; j < TBC; ++j) c1[j] = c2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TBR; ++i) {
      const float sc = (g.r0 + i < n) ? t.scal[min(g.r0 + i, n - 1)] : 0.f;
      const float d = part_rows<NWC, W>(part_r, g.r0 + i);
#pragma unroll
      for (int j = 0; j < TBC; ++j) {
        const float gz = (LTRX_B_GET(i, j) - m[i][j] * d) * inv_tau;
        c1[j] += gz * sc;
        c2[j] += gz;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < TBC; ++j) {
      c1[j] = sum_lr<LCN>(c1[j]);
      c2[j] = sum_lr<LCN>(c2[j]);
    }
    __syncthreads();                                    // part_r readers (just above) and the last part_c readers are done
    if (g.lr == 0) {
#pragma unroll
      for (int j = 0; j < TBC; ++j) {
        part_c[g.wr][g.c0 + j] = c1[j];
        part_r[g.wr][g.c0 + j] = c2[j];                 // (part_r reused, indexed by column here)
      }
    }
    __syncthreads();
    if (tid < W) {
      cvec[tid] = LTRX_PART4(part_c, tid);
      dvec[tid] = LTRX_PART4(part_r, tid);
    }
    __syncthreads();
  }
  for (int j = tid; j < n; j += blockDim.x) {
    const float sj = t.sc[j];
    float sgn_sum = 0.f, cross = 0.f;
    for (int mm = 0; mm < n; ++mm) {
      const float dlt = sj - t.sc[mm];
      const float sg = (dlt > 0.f) ? 1.0f : ((dlt < 0.f) ? -1.0f : 0.f);
      sgn_sum += sg;
      cross -= dvec[mm] * sg;
    }
    gp[t.vidx[j]] = cvec[j] - dvec[j] * sgn_sum + cross;
  }
#undef LTRX_BL
#undef LTRX_B_GET
#undef LTRX_B_SET
}

// residual trace of the block path: res[b][it] = max( max_j |colsum_{it+1}[j] - 1|, max_i |r_it[i] * (1 / r_it[i]) - 1| ) -- the
// column sums of the state after step `it` are the ones step it + 1 starts from (slot max_iter = after the last step)
__global__ void __launch_bounds__(256) ltrx_neural_residual_kernel(const float* __restrict__ y_true, int L, float pad, int max_iter,
                                                                   const float* __restrict__ cnws, const float* __restrict__ rnws,
                                                                   float* __restrict__ resws) {
  __shared__ int redi[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  int cnt = 0;
  for (int i = threadIdx.x; i < L; i += blockDim.x) cnt += (y_true[(size_t)b * L + i] != pad);
  const int n = block_sum_i(cnt, redi);
  const float* cn = cnws + (size_t)b * (max_iter + 1) * L;
  const float* rn = rnws + (size_t)b * max_iter * L;
  const int wave = wave_id(), lane = lane_id(), nw = blockDim.x >> 6;
  for (int it = wave; it < max_iter; it += nw) {        // one wave per step
    float a = 0.f;
    for (int j = lane; j < n; j += 64) {
      a = fmaxf(a, fabsf(cn[(size_t)(it + 1) * L + j] - 1.0f));
      const float r = rn[(size_t)it * L + j];
      a = fmaxf(a, fabsf(r * (1.0f / r) - 1.0f));
    }
    a = wave_max(a);
    if (lane == 0) resws[(size_t)b * max_iter + it] = (n > 0) ? a : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------

extern "C" size_t ltrx_neuralndcg_workspace_bytes(int B, int L, int max_iter) {
  if (B <= 0 || L <= 0 || max_iter < 0) return 0;
  const int mi = max_iter > 0 ? max_iter : 1;
  return align64((size_t)B * 4) + align64((size_t)B * mi * 4) + 64 + align64((size_t)B * (mi + 1) * L * 4) +
         align64((size_t)B * mi * L * 4) + 2 * align64((size_t)B * L * L * 4) + 64;
}

extern "C" int ltrx_neuralndcg_prepare(const float* y_true, int B, int L, float pad_value, int k, int idcg_powered,
                                       float* idcg_out, float* nonzero_count_out, void* ws, ltrx_stream_t stream) {
  if (!y_true || !idcg_out || !nonzero_count_out || !ws || B <= 0 || L <= 0) return LTRX_EINVAL;
  if (L > LTRX_MAX_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* nz = (float*)ws;   // [B] (aliases NeuralWs::per, which is not live yet)
  hipLaunchKernelGGL(ltrx_neural_idcg_kernel, dim3(B), dim3(256), (size_t)L * sizeof(float), s, y_true, L, pad_value, k,
                     idcg_powered, idcg_out, nz);
  LTRX_LAUNCH_CHECK();
  return ltrx_launch_finalize_sum(nz, B, 1.0f, nonzero_count_out, s);
}

extern "C" int ltrx_neuralndcg_fwd_bwd(const float* y_pred, const float* y_true, const float* idcg,
                                       const float* nonzero_count, int B, int L, float pad_value, float temperature,
                                       int powered_relevancies, int k, const int32_t* k_rows, int transposed,
                                       int max_iter, float tol, float* loss_out, float* per_slate_out, float* grad_out, int32_t* iters_out,
                                       int path, void* ws, ltrx_stream_t stream) {
  if (!y_pred || !y_true || !idcg || !nonzero_count || !loss_out || !ws || B <= 0 || L <= 0) return LTRX_EINVAL;
  if (!(temperature > 0.f) || max_iter < 0 || path < 0 || path > 1) return LTRX_EINVAL;
  if (L > LTRX_MAX_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int mi = max_iter > 0 ? max_iter : 1;
  NeuralWs w = carve(ws, B, L, mi);
  // neuralNDCG.py:48-49 vs :118-124: plain uses 2^y-1 or y (padded -> 0 either way); transposed uses y for the
  // non-powered case, whose padded entries meet zero columns -- identical on the valid block.
  (void)transposed;
  const int threads = 1024;   // 16 waves per slate
  const size_t lds = LTRX_NEURAL_LDS_FLOATS(L) * sizeof(float);
  // path 0: block-resident kernels (L <= 240, the WEB30K slate length; longer slates run the general kernels); 1: always the
  // general L2-streaming kernels
  const bool blk = (L <= 240) && path == 0;
#define LTRX_NEURAL_FWD_BLK(LRN_, TBR_, TBC_, NWC_)                                                                        \
  hipLaunchKernelGGL((ltrx_neural_forward_blk_kernel<LRN_, TBR_, TBC_, NWC_>), dim3(B), dim3(256 * NWC_), lds, s, y_pred,     \
                     y_true, L, pad_value, temperature, max_iter, w.S, w.cn, w.rn)
  if (blk) {       // 16 waves of small square tiles up to L = 192; above, 12 waves (4 x 3) of 15 x 5 tiles at 170 registers per thread
    if (L <= 64) LTRX_NEURAL_FWD_BLK(8, 2, 2, 4);
    else if (L <= 128) LTRX_NEURAL_FWD_BLK(8, 4, 4, 4);
    else if (L <= 192) LTRX_NEURAL_FWD_BLK(8, 6, 6, 4);
    else LTRX_NEURAL_FWD_BLK(4, 15, 5, 3);         // (8 waves of 15 x 8 tiles: 2 % slower, profiles/NOTES.md round 3)
    LTRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(ltrx_neural_residual_kernel, dim3(B), dim3(256), 0, s, y_true, L, pad_value, max_iter, w.cn, w.rn, w.res);
  } else {
    hipLaunchKernelGGL(ltrx_neural_forward_kernel, dim3(B), dim3(threads), lds, s, y_pred, y_true, L, pad_value,
                       temperature, max_iter, w.S, w.cn, w.rn, w.res);
  }
#undef LTRX_NEURAL_FWD_BLK
  LTRX_LAUNCH_CHECK();
  hipLaunchKernelGGL(ltrx_neural_pick_iter_kernel, dim3(1), dim3(256), 0, s, w.res, B, max_iter, tol, w.titer, iters_out);
  LTRX_LAUNCH_CHECK();
#define LTRX_NEURAL_BWD_BLK(LRN_, TBR_, TBC_, NWC_, TLC_)                                                                  \
  hipLaunchKernelGGL((ltrx_neural_backward_blk_kernel<LRN_, TBR_, TBC_, NWC_, TLC_>), dim3(B), dim3(256 * NWC_), 0, s, y_pred, \
                     y_true, idcg, nonzero_count, L, pad_value, 1.0f / temperature, powered_relevancies, k, k_rows, max_iter,       \
                     w.titer, w.S, w.cn, w.rn, w.per, per_slate_out, grad_out)
  if (blk) {
    if (L <= 64) LTRX_NEURAL_BWD_BLK(8, 2, 2, 4, 0);
    else if (L <= 128) LTRX_NEURAL_BWD_BLK(8, 4, 4, 4, 0);
    else if (L <= 192) LTRX_NEURAL_BWD_BLK(8, 6, 6, 4, 0);
    else LTRX_NEURAL_BWD_BLK(4, 15, 5, 3, 2);
  } else {
    hipLaunchKernelGGL(ltrx_neural_backward_kernel, dim3(B), dim3(threads), lds, s, y_pred, y_true, idcg, nonzero_count, L,
                       pad_value, 1.0f / temperature, powered_relevancies, k, k_rows, max_iter, w.titer, w.S, w.A, w.cn, w.rn,
                       w.per, per_slate_out, grad_out);
  }
#undef LTRX_NEURAL_BWD_BLK
  LTRX_LAUNCH_CHECK();
  // loss = -sum_b value_b / nonzero_count  (device scalar) -> two tiny kernels: sum, then scale
  int rc = ltrx_launch_finalize_sum(w.per, B, -1.0f, loss_out, s);
  if (rc != LTRX_OK) return rc;
  return ltrx_launch_div_by_device_scalar(loss_out, nonzero_count, s);
}
