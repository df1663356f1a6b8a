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
// Layout: one workgroup per slate; the n x n state S and adjoint A live in a global workspace (L2 / Infinity
// Cache resident: 2 * 230 KB per slate at L = 240).  Column phases: one thread per column, lanes read
// consecutive addresses of a row (coalesced).  Row phases: one wave per row, lane-strided, wave-shuffle sums.
// (A register-resident variant for L <= 256 is the planned optimisation; this kernel is general in L.)
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
  w.cn = (float*)p;    p += align64((size_t)B * max_iter * L * 4);
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
// Register-resident fast path (n_valid <= 16 * NR, L <= 64 * NC; L <= 240 -> NR = 15, NC = 4):
// a 1024-thread workgroup (16 waves) holds the whole n x n Sinkhorn state in VGPRs -- wave w owns rows
// w, w+16, ..., lane l owns columns l, l+64, ... (60 registers at L = 240; 240^2 fp32 = 230 KB would not fit the
// 160 KB LDS).  Row sums are lane-local adds + one wave shuffle reduction per row; column sums are per-wave partials
// exchanged through a 16 KB LDS array and combined in a fixed order.  The backward kernel carries the adjoint the
// same way (120 state registers).  Normalisers are applied as x * (1/c) (<= 1 ulp from the reference's x / c).
// Same outputs / workspace contract as the general kernels above (cn, rn, res, S), so pick_iter and the batch-global
// early-exit replay are shared.
// ---------------------------------------------------------------------------------------------------------
template <int NR, int NC>
struct RegMat {
  float v[NR][NC];
};

#define LTRX_FOR_RC for (int ri = 0; ri < NR; ++ri) for (int cj = 0; cj < NC; ++cj)

template <int NR, int NC>
__global__ void __launch_bounds__(1024) ltrx_neural_forward_reg_kernel(const float* __restrict__ y_pred,
                                                                       const float* __restrict__ y_true, int L, float pad,
                                                                       float tau, int max_iter, float* __restrict__ Sws,
                                                                       float* __restrict__ cnws, float* __restrict__ rnws,
                                                                       float* __restrict__ resws) {
  extern __shared__ float lds[];
  __shared__ float red[LTRX_MAX_WAVES];
  __shared__ int redi[LTRX_MAX_WAVES];
  __shared__ float colpart[16][64 * NC];
  __shared__ float cvec[64 * NC];
  const SlateLds t = carve_lds(lds, L);
  const int b = blockIdx.x;
  const int n = load_slate(t, y_pred + (size_t)b * L, y_true + (size_t)b * L, L, pad, 0, 0, redi);
  float* S = Sws + (size_t)b * L * L;
  float* cn = cnws + (size_t)b * max_iter * L;
  float* rn = rnws + (size_t)b * max_iter * L;
  float* res = resws + (size_t)b * max_iter;
  const int lane = lane_id(), w = wave_id();
  if (n == 0) {
    for (int it = threadIdx.x; it < max_iter; it += blockDim.x) res[it] = 0.f;
    return;
  }
  RegMat<NR, NC> m;
  // ---- P0 = row softmax ----
#pragma unroll
  for (int ri = 0; ri < NR; ++ri) {
    const int i = w + 16 * ri;
    const float sc_i = (i < n) ? t.scal[i] : 0.f;
    float z[NC];
    float mx = -INFINITY;
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) {
      const int j = lane + 64 * cj;
      z[cj] = (i < n && j < n) ? (sc_i * t.sc[j] - t.bs[j]) / tau : -INFINITY;
      mx = fmaxf(mx, z[cj]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) {
      const float e = (z[cj] == -INFINITY) ? 0.f : expf(z[cj] - mx);
      m.v[ri][cj] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = (sum > 0.f) ? 1.0f / sum : 0.f;
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) m.v[ri][cj] = (sum > 0.f) ? m.v[ri][cj] / sum : 0.f;
    (void)inv;
    __builtin_amdgcn_sched_barrier(0);     // keep the rows sequential: interleaving all 15 blows the 128-VGPR budget
  }
  // ---- Sinkhorn ----
  float rowres_prev = 0.f;
  for (int it = 0; it <= max_iter; ++it) {
    float* cn_it = cn + (size_t)it * L;
    float* rn_it = rn + (size_t)it * L;
    // column sums: per-wave partials -> LDS -> fixed-order combine
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) {
      float a = 0.f;
#pragma unroll
      for (int ri = 0; ri < NR; ++ri) a += m.v[ri][cj];
      colpart[w][lane + 64 * cj] = a;
    }
    __syncthreads();
    float cres = 0.f;
    if (threadIdx.x < 64 * NC) {
      const int j = threadIdx.x;
      float c = 0.f;
#pragma unroll
      for (int ww = 0; ww < 16; ++ww) c += colpart[ww][j];
      if (j < n) {
        cres = fabsf(c - 1.0f);
        if (it < max_iter) {
          c = fmaxf(c, kSinkEps);
          cn_it[j] = c;
        }
      }
      cvec[j] = (j < n) ? 1.0f / c : 0.f;
    }
    if (it > 0) {
      cres = block_max(cres, red);
      if (threadIdx.x == 0) res[it - 1] = fmaxf(cres, rowres_prev);
    }
    if (it == max_iter) break;
    __syncthreads();
    float rcj[NC];
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) rcj[cj] = cvec[lane + 64 * cj];
    float rres = 0.f;
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
      const int i = w + 16 * ri;
      float r = 0.f;
#pragma unroll
      for (int cj = 0; cj < NC; ++cj) {
        m.v[ri][cj] *= rcj[cj];
        r += m.v[ri][cj];
      }
      r = fmaxf(wave_sum(r), kSinkEps);
      const float rr = 1.0f / r;
#pragma unroll
      for (int cj = 0; cj < NC; ++cj) m.v[ri][cj] *= rr;
      if (i < n) {
        if (lane == 0) rn_it[i] = r;
        rres = fmaxf(rres, fabsf(r * rr - 1.0f));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    rowres_prev = block_max(rres, red);     // (barriers: colpart / cvec may be overwritten next iteration)
  }
  // ---- publish the final state for the backward kernel ----
#pragma unroll
  for (int ri = 0; ri < NR; ++ri) {
    const int i = w + 16 * ri;
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) {
      const int j = lane + 64 * cj;
      if (i < n && j < n) S[(size_t)i * n + j] = m.v[ri][cj];
    }
  }
}

// Backward: state m (NR x NC registers) + adjoint.  NCL of the NC adjoint column slots live in LDS (a_lds[slot][tid],
// conflict-free: consecutive threads -> consecutive banks) so that the 1024-thread workgroup stays under its
// 128-VGPR budget: at L = 240 (NR 15, NC 4, NCL 2) that is 60 + 30 registers + 123 KB of the 160 KB LDS.
template <int NR, int NC, int NCL>
__global__ void __launch_bounds__(1024) ltrx_neural_backward_reg_kernel(
    const float* __restrict__ y_pred, const float* __restrict__ y_true, const float* __restrict__ idcg,
    const float* __restrict__ nonzero_count, int L, float pad, float inv_tau, int gain_powered, int k,
    const int* __restrict__ k_rows, int max_iter,
    const int* __restrict__ titer, const float* __restrict__ Sws, const float* __restrict__ cnws,
    const float* __restrict__ rnws, float* __restrict__ per_ws, float* __restrict__ per_out, float* __restrict__ grad) {
  constexpr int NCR = NC - NCL;                         // adjoint column slots kept in registers
  __shared__ float tables[7 * 64 * NC];
  __shared__ float red[LTRX_MAX_WAVES];
  __shared__ int redi[LTRX_MAX_WAVES];
  __shared__ float colpart[16][64 * NC];
  __shared__ float cvec[64 * NC];
  __shared__ float dvec[64 * NC];
  __shared__ float a_lds[(NCL > 0 ? NR * NCL : 1) * 1024];
  const SlateLds t = carve_lds(tables, L);
  const int b = blockIdx.x;
  int kk = (k <= 0 || k > L) ? L : k;
  if (k_rows) kk = min(kk, k_rows[b]);
  const int n = load_slate(t, y_pred + (size_t)b * L, y_true + (size_t)b * L, L, pad, kk, gain_powered, redi);
  const float* S = Sws + (size_t)b * L * L;
  const float* cn = cnws + (size_t)b * max_iter * L;
  const float* rn = rnws + (size_t)b * max_iter * L;
  const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
  const int T = titer[0];
  float* gp = grad ? grad + (size_t)b * L : nullptr;
  if (gp)
    for (int i = tid; i < L; i += blockDim.x) gp[i] = 0.f;
  const float id = idcg[b];
  const float cnt = nonzero_count[0];
  if (n == 0 || id == 0.f) {
    if (tid == 0) {
      per_ws[b] = 0.f;
      if (per_out) per_out[b] = 0.f;
    }
    return;
  }
  float m[NR][NC];
  float ar[NR][NCR > 0 ? NCR : 1];
#define LTRX_A_GET(ri, cj) ((cj) < NCR ? ar[ri][(cj) < NCR ? (cj) : 0] : a_lds[((ri) * NCL + ((cj) - NCR)) * 1024 + tid])
#define LTRX_A_SET(ri, cj, val)                                        \
  do {                                                                 \
    if ((cj) < NCR) ar[ri][(cj) < NCR ? (cj) : 0] = (val);             \
    else a_lds[((ri) * NCL + ((cj) - NCR)) * 1024 + tid] = (val);      \
  } while (0)
#pragma unroll
  for (int ri = 0; ri < NR; ++ri)
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) {
      const int i = w + 16 * ri, j = lane + 64 * cj;
      m[ri][cj] = (i < n && j < n) ? S[(size_t)i * n + j] : 0.f;
    }
  // ---- rewind the steps max_iter-1 .. T ----
  for (int it = max_iter - 1; it >= T; --it) {
    const float* cn_it = cn + (size_t)it * L;
    const float* rn_it = rn + (size_t)it * L;
    float cj_[NC];
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) cj_[cj] = (lane + 64 * cj < n) ? cn_it[lane + 64 * cj] : 0.f;
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
      const int i = w + 16 * ri;
      const float r = (i < n) ? rn_it[i] : 0.f;
#pragma unroll
      for (int cj = 0; cj < NC; ++cj) m[ri][cj] = (m[ri][cj] * r) * cj_[cj];
    }
  }
  // ---- read-out ----
  float v = 0.f;
  {
    float gcj[NC];
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) gcj[cj] = (lane + 64 * cj < n) ? t.gc[lane + 64 * cj] : 0.f;
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
      const int i = w + 16 * ri;
      float acc = 0.f;
#pragma unroll
      for (int cj = 0; cj < NC; ++cj) acc += m[ri][cj] * gcj[cj];
      acc = wave_sum(acc);
      if (lane == 0 && i < n) v += t.dk[i] * acc;
    }
    v = block_sum(v, red);
    const float value = v / (id + kSinkEps);
    if (tid == 0) {
      per_ws[b] = value;
      if (per_out) per_out[b] = value;
    }
    if (!gp) return;
    const float coef = -1.0f / (cnt * (id + kSinkEps));
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
      const int i = w + 16 * ri;
      const float d = (i < n) ? coef * t.dk[i] : 0.f;
#pragma unroll
      for (int cj = 0; cj < NC; ++cj) LTRX_A_SET(ri, cj, d * gcj[cj]);
    }
  }
  // ---- reverse Sinkhorn ----
  for (int it = T - 1; it >= 0; --it) {
    const float* cn_it = cn + (size_t)it * L;
    const float* rn_it = rn + (size_t)it * L;
    // row step:  Y2 = Y1 / r
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
      const int i = w + 16 * ri;
      const float r = (i < n) ? rn_it[i] : 1.0f;
      float av[NC];
      float d = 0.f;
#pragma unroll
      for (int cj = 0; cj < NC; ++cj) {
        av[cj] = LTRX_A_GET(ri, cj);
        d += av[cj] * m[ri][cj];
      }
      d = (r > kSinkEps) ? wave_sum(d) : 0.f;
      const float rr = 1.0f / r;
#pragma unroll
      for (int cj = 0; cj < NC; ++cj) {
        const bool live = (i < n) && (lane + 64 * cj < n);
        LTRX_A_SET(ri, cj, live ? (av[cj] - d) * rr : 0.f);
        m[ri][cj] *= r;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // column step:  Y1 = X / c
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) {
      float d = 0.f;
#pragma unroll
      for (int ri = 0; ri < NR; ++ri) d += LTRX_A_GET(ri, cj) * m[ri][cj];
      colpart[w][lane + 64 * cj] = d;
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (tid < 64 * NC) {
      float d = 0.f;
#pragma unroll
      for (int ww = 0; ww < 16; ++ww) d += colpart[ww][tid];
      const float c = (tid < n) ? cn_it[tid] : 1.0f;
      dvec[tid] = (c > kSinkEps) ? d : 0.f;
      cvec[tid] = c;
    }
    __syncthreads();
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) {
      const int j = lane + 64 * cj;
      const float c = cvec[j], d = dvec[j];
      const float rc = 1.0f / c;
#pragma unroll
      for (int ri = 0; ri < NR; ++ri) {
        const bool live = (w + 16 * ri < n) && (j < n);
        LTRX_A_SET(ri, cj, live ? (LTRX_A_GET(ri, cj) - d) * rc : 0.f);
        m[ri][cj] *= c;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();      // colpart / cvec / dvec are rewritten by the next step
  }
  // ---- row softmax backward (m holds P0 again): gz = P0 (g0 - <g0, P0>) / tau ----
#pragma unroll
  for (int ri = 0; ri < NR; ++ri) {
    float av[NC];
    float d = 0.f;
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) {
      av[cj] = LTRX_A_GET(ri, cj);
      d += av[cj] * m[ri][cj];
    }
    d = wave_sum(d);
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) LTRX_A_SET(ri, cj, m[ri][cj] * (av[cj] - d) * inv_tau);
  }
  // ---- column sums of gz: weighted by the row scaling (direct term), then plain (Q) ----
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int cj = 0; cj < NC; ++cj) {
      float s1 = 0.f;
#pragma unroll
      for (int ri = 0; ri < NR; ++ri) {
        const int i = w + 16 * ri;
        const float g = LTRX_A_GET(ri, cj);
        s1 += (pass == 0) ? ((i < n) ? g * t.scal[i] : 0.f) : g;
      }
      colpart[w][lane + 64 * cj] = s1;
    }
    __syncthreads();
    if (tid < 64 * NC) {
      float s1 = 0.f;
#pragma unroll
      for (int ww = 0; ww < 16; ++ww) s1 += colpart[ww][tid];
      if (pass == 0) cvec[tid] = s1; else dvec[tid] = s1;
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
#undef LTRX_A_GET
#undef LTRX_A_SET
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------

extern "C" size_t ltrx_neuralndcg_workspace_bytes(int B, int L, int max_iter) {
  if (B <= 0 || L <= 0 || max_iter < 0) return 0;
  const int mi = max_iter > 0 ? max_iter : 1;
  return align64((size_t)B * 4) + align64((size_t)B * mi * 4) + 64 + 2 * align64((size_t)B * mi * L * 4) +
         2 * align64((size_t)B * L * L * 4) + 64;
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
  const bool fast = (L <= 240) && path == 0;     // path 1: the general (L2-streaming) kernels for every L
#define LTRX_NEURAL_FWD(NR, NC)                                                                                          \
  hipLaunchKernelGGL((ltrx_neural_forward_reg_kernel<NR, NC>), dim3(B), dim3(1024), lds, s, y_pred, y_true, L, pad_value, \
                     temperature, max_iter, w.S, w.cn, w.rn, w.res)
  if (fast) {
    if (L <= 64) LTRX_NEURAL_FWD(4, 1);
    else if (L <= 128) LTRX_NEURAL_FWD(8, 2);
    else if (L <= 192) LTRX_NEURAL_FWD(12, 3);
    else LTRX_NEURAL_FWD(15, 4);
  } else {
    hipLaunchKernelGGL(ltrx_neural_forward_kernel, dim3(B), dim3(threads), lds, s, y_pred, y_true, L, pad_value,
                       temperature, max_iter, w.S, w.cn, w.rn, w.res);
  }
#undef LTRX_NEURAL_FWD
  LTRX_LAUNCH_CHECK();
  hipLaunchKernelGGL(ltrx_neural_pick_iter_kernel, dim3(1), dim3(256), 0, s, w.res, B, max_iter, tol, w.titer, iters_out);
  LTRX_LAUNCH_CHECK();
#define LTRX_NEURAL_BWD(NR, NC, NCL)                                                                                      \
  hipLaunchKernelGGL((ltrx_neural_backward_reg_kernel<NR, NC, NCL>), dim3(B), dim3(1024), 0, s, y_pred, y_true, idcg,      \
                     nonzero_count, L, pad_value, 1.0f / temperature, powered_relevancies, k, k_rows, max_iter, w.titer, w.S, w.cn, \
                     w.rn, w.per, per_slate_out, grad_out)
  if (fast) {
    if (L <= 64) LTRX_NEURAL_BWD(4, 1, 0);
    else if (L <= 128) LTRX_NEURAL_BWD(8, 2, 0);
    else if (L <= 192) LTRX_NEURAL_BWD(12, 3, 0);
    else LTRX_NEURAL_BWD(15, 4, 2);
  } else {
    hipLaunchKernelGGL(ltrx_neural_backward_kernel, dim3(B), dim3(threads), lds, s, y_pred, y_true, idcg, nonzero_count, L,
                       pad_value, 1.0f / temperature, powered_relevancies, k, k_rows, max_iter, w.titer, w.S, w.A, w.cn, w.rn,
                       w.per, per_slate_out, grad_out);
  }
#undef LTRX_NEURAL_BWD
  LTRX_LAUNCH_CHECK();
  // loss = -sum_b value_b / nonzero_count  (device scalar) -> two tiny kernels: sum, then scale
  int rc = ltrx_launch_finalize_sum(w.per, B, -1.0f, loss_out, s);
  if (rc != LTRX_OK) return rc;
  return ltrx_launch_div_by_device_scalar(loss_out, nonzero_count, s);
}
