// Fused LambdaLoss forward + backward, all 8 weighing schemes.  Reference: allrank/models/losses/lambdaLoss.py:7-114.
//
// In the reference everything lives in "sorted by prediction" space.  Only the sorted POSITION of an item
// matters (discount D = log2(1 + position), the top-k mask), so the kernel keeps the original item order and
// gives every valid item its stable-descending rank by a counting rank out of LDS:
//     rank_i = #{ j valid : s_j > s_i  or (s_j == s_i and j < i) }            (position = rank + 1)
// Pair (i, j) is selected (lambdaLoss.py:39-45,75) iff both valid, rank_i < k, rank_j < k and
// (scheme == ndcgLoss1 ? true : y_i > y_j)   -- ndcgLoss1 keeps the diagonal, like the reference.
//     q_ij   = max(sigmoid(sigma (s_i - s_j)), eps)
//     l_ij   = log_b( max(q_ij ^ w_ij, eps) ) = max(w_ij * ln q_ij, ln eps) / ln b          (:66-72)
//     loss   = -sum l_ij            (reduction sum)   |   -sum l_ij / #selected   (mean, batch-global count)
//     dl_ij/d(s_i - s_j) = w_ij * sigma * (1 - sigmoid) / ln b    where neither clamp is active, else 0;
// the weights depend on scores only through the (piecewise constant) ranks -> no gradient through them.
// Each thread owns items i = tid, tid+T, ... and walks j over the slate in LDS (broadcast reads), evaluating
// both orientations (i,j) and (j,i) (at most one is live except for ndcgLoss1).  The L x L matrices of the reference never exist.
#include "ltrx_device.h"

using namespace ltrx;

namespace {

struct PairCtx {
  int scheme;
  float mu, sigma, eps, ln_eps, inv_lnb;
  const float* invD;  // invD[p] = 1/log2(1+p), p = 0..L+1 (invD[0] unused)
};

// weight of the ordered pair (first = a, second = b); ra/rb are 0-based ranks.
__device__ __forceinline__ float pair_weight(const PairCtx& c, float Ga, float Gb, int ra, int rb, float ya, float yb) {
  switch (c.scheme) {
    case LTRX_SCHEME_NDCGLOSS1:
      return Ga * c.invD[ra + 1];                                   // (G / D)[:, :, None]      lambdaLoss.py:85
    case LTRX_SCHEME_NDCGLOSS2: {
      const int d = abs(ra - rb);
      const float del = (d == 0) ? 0.f : fabsf(c.invD[d] - c.invD[d + 1]);   // :88-92
      return del * fabsf(Ga - Gb);
    }
    case LTRX_SCHEME_LAMBDARANK:
      return fabsf(c.invD[ra + 1] - c.invD[rb + 1]) * fabsf(Ga - Gb);       // :97-98
    case LTRX_SCHEME_NDCGLOSS2PP: {
      const int d = abs(ra - rb);
      const float del = (d == 0) ? 0.f : fabsf(c.invD[d] - c.invD[d + 1]);
      const float dg = fabsf(Ga - Gb);
      return c.mu * (del * dg) + fabsf(c.invD[ra + 1] - c.invD[rb + 1]) * dg;   // :101-102
    }
    case LTRX_SCHEME_RANKNET_GTDIFF:
      return fabsf(ya - yb);                                                 // :109-110
    case LTRX_SCHEME_RANKNET_GTDIFF_POWED:
      return fabsf(ya * ya - yb * yb);                                       // :113-114
    default:
      return 1.0f;                                                           // None / rankNet_scheme
  }
}

// loss term and d/d(s_a - s_b) of the ordered pair given sig = sigmoid(sigma (s_a - s_b)).
__device__ __forceinline__ void pair_terms(const PairCtx& c, float w, float sig, float& l, float& g) {
  const float q = fmaxf(sig, c.eps);
  const float wl = w * logf(q);
  const bool live = (sig >= c.eps) && (wl >= c.ln_eps);
  l = fmaxf(wl, c.ln_eps) * c.inv_lnb;
  g = live ? w * c.sigma * (1.0f - sig) * c.inv_lnb : 0.f;
}

}  // namespace

// 1024 threads per slate: thread (i, q) = (tid & 255 [+256 ...], tid >> 8) owns item i and a quarter of the partner range j;
// the four partial sums per item are combined through LDS (16 waves per CU instead of 4 hide the exp/log latency).
// GWS: the thirteen work arrays live in a global workspace (slates too long for LDS; ltrx_device.h)
template <bool GWS>
__global__ void __launch_bounds__(1024) ltrx_lambdaloss_kernel(const float* __restrict__ y_pred,
                                                              const float* __restrict__ y_true, int L, float eps,
                                                              float pad, int scheme, int k, float sigma, float mu,
                                                              int logbase, float* __restrict__ per_loss,
                                                              float* __restrict__ per_cnt, float* __restrict__ grad,
                                                              int64_t* __restrict__ order_out, float* gws, size_t gws_stride) {
  extern __shared__ float lds[];
  float* base = GWS ? gws + (size_t)blockIdx.x * gws_stride : lds;
  float* ss = base;                    // [L] scores
  float* ys = base + L;                // [L] labels (pad kept)
  float* Gs = base + 2 * L;            // [L] gains / maxDCG
  int* rk = (int*)(base + 3 * L);      // [L] rank by score (valid items; padded get L)
  float* invD = base + 4 * L;          // [L+2]
  float* part = base + 5 * L + 2;      // [4][L] float partials
  int* parti = (int*)(base + 9 * L + 2);   // [4][L] int partials (score-rank and label-rank counts packed: rs | ry << 16)
  __shared__ float red[LTRX_MAX_WAVES];
  __shared__ int redi[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  const float* sp = y_pred + (size_t)b * L;
  const float* yp = y_true + (size_t)b * L;
  int nv = 0;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    ss[i] = sp[i];
    const float y = yp[i];
    ys[i] = y;
    nv += (y != pad);
  }
  for (int p = threadIdx.x; p < L + 2; p += blockDim.x) invD[p] = (p == 0) ? 0.f : 1.0f / log2f(1.0f + (float)p);
  nv = block_sum_i(nv, redi);         // barriers publish ss/ys/invD
  const int kk = (k <= 0 || k > L) ? L : k;

  const int q = threadIdx.x >> 8, i0 = threadIdx.x & 255;
  const int lq = (L + 3) >> 2;
  const int j0 = q * lq, j1 = min(L, j0 + lq);
  // ---- ranks by score, ideal DCG@k by label rank ----
  for (int i = i0; i < L; i += 256) {
    const float yi = ys[i], si = ss[i];
    int rs = 0, ry = 0;
    if (yi != pad)
      for (int j = j0; j < j1; ++j) {
        const float yj = ys[j];
        if (yj == pad) continue;
        const float sj = ss[j];
        rs += (sj > si) || (sj == si && j < i);
        ry += (yj > yi) || (yj == yi && j < i);
      }
    parti[q * L + i] = rs | (ry << 16);            // a quarter of L <= LTRX_MAX_LONG_SLATE_LEN = 16384 partners: both counts fit 15 bits
  }
  __syncthreads();
  float dsum = 0.f;
  if (q == 0)
    for (int i = i0; i < L; i += 256) {
      const float yi = ys[i];
      if (yi == pad) {
        rk[i] = L;
        continue;
      }
      const int a0 = parti[i], a1 = parti[L + i], a2 = parti[2 * L + i], a3 = parti[3 * L + i];
      const int rs = (a0 & 0xFFFF) + (a1 & 0xFFFF) + (a2 & 0xFFFF) + (a3 & 0xFFFF);
      const int ry = (a0 >> 16) + (a1 >> 16) + (a2 >> 16) + (a3 >> 16);
      rk[i] = rs;
      if (ry < kk) dsum += (exp2f(fmaxf(yi, 0.f)) - 1.0f) * invD[ry + 1];     // lambdaLoss.py:54
    }
  const float maxdcg = fmaxf(block_sum(dsum, red), eps);
  for (int i = threadIdx.x; i < L; i += blockDim.x) Gs[i] = (ys[i] == pad) ? 0.f : (exp2f(fmaxf(ys[i], 0.f)) - 1.0f) / maxdcg;
  __syncthreads();
  if (order_out) {   // stable descending argsort of the masked predictions: valid by rank, padded after, in index order
    int64_t* op = order_out + (size_t)b * L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
      if (ys[i] != pad) {
        op[rk[i]] = i;
      } else {
        int before = 0;
        for (int j = 0; j < i; ++j) before += (ys[j] == pad);
        op[nv + before] = i;
      }
    }
  }

  PairCtx c;
  c.scheme = scheme;
  c.mu = mu;
  c.sigma = sigma;
  c.eps = eps;
  c.ln_eps = logf(eps);
  c.inv_lnb = (logbase == LTRX_LOG_NATURAL) ? 1.0f : 1.4426950408889634f;
  c.invD = invD;
  const bool all_pairs = (scheme == LTRX_SCHEME_NDCGLOSS1);

  float lsum = 0.f, csum = 0.f;
  float* gp = grad ? grad + (size_t)b * L : nullptr;
  for (int i = i0; i < L; i += 256) {
    const float yi = ys[i];
    const int ri = rk[i];
    float gacc = 0.f;
    if (yi != pad && ri < kk) {
      const float si = ss[i], Gi = Gs[i], yci = fmaxf(yi, 0.f);
      for (int j = j0; j < j1; ++j) {
        const float yj = ys[j];
        const int rj = rk[j];
        if (yj == pad || rj >= kk) continue;
        const bool fwd = all_pairs || (yi > yj);          // pair (i, j): i is the "first" element
        const bool bwd = (j != i) && (all_pairs || (yj > yi));   // pair (j, i): i is the "second" element
        if (!(fwd || bwd)) continue;
        const float Gj = Gs[j], ycj = fmaxf(yj, 0.f);
        const float dx = sigma * (si - ss[j]);
        if (fwd) {
          float l, g;
          pair_terms(c, pair_weight(c, Gi, Gj, ri, rj, yci, ycj), 1.0f / (1.0f + expf(-dx)), l, g);
          lsum += l;
          csum += 1.0f;
          if (j != i) gacc -= g;   // d(-l_ij)/d s_i ; the diagonal pair (ndcgLoss1 only) has no score dependence
        }
        if (bwd) {
          // sigmoid(sigma (s_j - s_i)) evaluated directly (1 - sigmoid(dx) would cancel catastrophically for large dx)
          float l, g;
          pair_terms(c, pair_weight(c, Gj, Gi, rj, ri, ycj, yci), 1.0f / (1.0f + expf(dx)), l, g);
          gacc += g;               // d(-l_ji)/d s_i   (s_i enters pair (j,i) with a minus sign)
        }
      }
    }
    part[q * L + i] = gacc;
  }
  __syncthreads();
  if (gp && q == 0)
    for (int i = i0; i < L; i += 256) gp[i] = (part[i] + part[L + i]) + (part[2 * L + i] + part[3 * L + i]);
  lsum = block_sum(lsum, red);
  csum = block_sum(csum, red);
  if (threadIdx.x == 0) {
    per_loss[b] = lsum;
    per_cnt[b] = csum;
  }
}

// loss_out = -sum / (mean ? count : 1);   count_out = #selected pairs.
__global__ void __launch_bounds__(256) ltrx_lambdaloss_finalize_kernel(const float* __restrict__ per_loss,
                                                                       const float* __restrict__ per_cnt, int B,
                                                                       int reduction,
                                                                       const float* __restrict__ ext_count,
                                                                       float* __restrict__ loss_out,
                                                                       float* __restrict__ cnt_out,
                                                                       float* __restrict__ scale_ws) {
  __shared__ float red[LTRX_MAX_WAVES];
  float a = 0.f, c = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    a += per_loss[b];
    c += per_cnt[b];
  }
  a = block_sum(a, red);
  c = block_sum(c, red);
  if (threadIdx.x == 0) {
    const float div = (reduction == LTRX_REDUCE_MEAN) ? (ext_count ? ext_count[0] : c) : 1.0f;
    loss_out[0] = -a / div;     // mean over an empty selection -> NaN, like torch.mean of an empty tensor
    if (cnt_out) cnt_out[0] = c;
    scale_ws[0] = (div > 0.f) ? 1.0f / div : 0.f;
  }
}

__global__ void __launch_bounds__(256) ltrx_scale_by_device_scalar_kernel(float* __restrict__ x, size_t n,
                                                                          const float* __restrict__ scale) {
  const float sc = scale[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= sc;
}

static size_t lambda_per_floats(int B) { return ((size_t)(2 * (B > 0 ? B : 0) + 4) + 3) & ~(size_t)3; }
extern "C" size_t ltrx_lambdaloss_workspace_bytes(int B, int L) {
  return (lambda_per_floats(B) + ltrx_array_ws_floats(13, 2, B > 0 ? B : 0, L > 0 ? L : 0)) * sizeof(float);
}

extern "C" int ltrx_lambdaloss_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps,
                                       float pad_value, int scheme, int k, float sigma, float mu, int reduction,
                                       int logbase, const float* ext_pair_count, float* loss_out,
                                       float* pair_count_out, float* grad_out, int64_t* order_out, void* ws,
                                       ltrx_stream_t stream) {
  if (!y_pred || !y_true || !loss_out || !ws || B <= 0 || L <= 0) return LTRX_EINVAL;
  if (scheme < 0 || scheme > LTRX_SCHEME_RANKNET_GTDIFF_POWED) return LTRX_EINVAL;
  if (reduction != LTRX_REDUCE_SUM && reduction != LTRX_REDUCE_MEAN) return LTRX_EINVAL;
  if (logbase != LTRX_LOG_BINARY && logbase != LTRX_LOG_NATURAL) return LTRX_EINVAL;
  if (!(eps > 0.f)) return LTRX_EINVAL;
  if (L > LTRX_MAX_LONG_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* per_loss = (float*)ws;
  float* per_cnt = per_loss + B;
  float* scale = per_cnt + B;
  if (ltrx_arrays_in_lds(13, 2, L)) {
    const size_t lds = (size_t)(13 * L + 2) * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};   // long slates need more than the default 64 KB of dynamic LDS
    const int arc = ltrx_once_per_device(attr_done, []() {
      return hipFuncSetAttribute((const void*)ltrx_lambdaloss_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 LTRX_LDS_ARRAY_BUDGET_BYTES) == hipSuccess ? LTRX_OK : LTRX_EHIP;
    });
    if (arc != LTRX_OK) return arc;
    hipLaunchKernelGGL(ltrx_lambdaloss_kernel<false>, dim3(B), dim3(1024), lds, s, y_pred, y_true, L, eps, pad_value, scheme, k, sigma, mu,
                       logbase, per_loss, per_cnt, grad_out, order_out, (float*)nullptr, (size_t)0);
  } else {
    hipLaunchKernelGGL(ltrx_lambdaloss_kernel<true>, dim3(B), dim3(1024), 0, s, y_pred, y_true, L, eps, pad_value, scheme, k, sigma, mu,
                       logbase, per_loss, per_cnt, grad_out, order_out, per_loss + lambda_per_floats(B), ltrx_array_ws_stride(13, 2, L));
  }
  LTRX_LAUNCH_CHECK();
  hipLaunchKernelGGL(ltrx_lambdaloss_finalize_kernel, dim3(1), dim3(256), 0, s, per_loss, per_cnt, B, reduction,
                     ext_pair_count, loss_out, pair_count_out, scale);
  LTRX_LAUNCH_CHECK();
  if (grad_out && reduction == LTRX_REDUCE_MEAN) {
    const size_t n = (size_t)B * L;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ltrx_scale_by_device_scalar_kernel, dim3(blocks), dim3(256), 0, s, grad_out, n, scale);
    LTRX_LAUNCH_CHECK();
  }
  return LTRX_OK;
}
