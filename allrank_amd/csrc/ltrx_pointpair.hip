// The pointwise / pairwise members of allrank.models.losses and the MRR metric:
//   ltrx_ranknet_fwd_bwd          rankNet, rankNet_weightByGTDiff, rankNet_weightByGTDiff_pow      rankNet.py:8-79
//   ltrx_bce_fwd_bwd              bce (bce.py:8-32) and ordinal (ordinal.py:8-50; n ordinal outputs per item)
//   ltrx_pointwise_rmse_fwd_bwd   pointwise_rmse                                                    pointwise.py:6-32
//   ltrx_binary_listnet_fwd_bwd   binary_listNet                                                    binary_listNet.py:8-33
//   ltrx_mrr_at                   mrr                                                               metrics.py:80-113
// Same skeleton as the listwise kernels: one workgroup per slate, scores/labels staged once in LDS, forward value and
// d loss / d y_pred from the same pass, padded slots get an exact 0 gradient.  Losses whose normaliser is a batch-global
// COUNT (selected pairs, slates / documents with a valid item) write un-normalised gradients plus per-slate (sum, count)
// and a second, tiny pass divides by the count read from DEVICE memory -- either this call's or, under slate sharding, the
// all-reduced one handed in by the caller (same protocol as ltrx_lambdaloss_fwd_bwd).
#include "ltrx_device.h"

using namespace ltrx;

namespace {

// out[0] = sum_b a[b] / D, out2[0] = D  with D = ext ? ext[0] : sum_b c[b];  D == 0 -> NaN (torch: mean of an empty
// selection, 0/0).  One block, fixed order.
__global__ void __launch_bounds__(256) ltrx_ratio_finalize_kernel(const float* __restrict__ a, const float* __restrict__ c,
                                                                  int B, const float* __restrict__ ext,
                                                                  float* __restrict__ loss_out, float* __restrict__ count_out,
                                                                  float* __restrict__ inv_out) {
  __shared__ float red[LTRX_MAX_WAVES];
  float sa = 0.f, sc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    sa += a[b];
    sc += c[b];
  }
  sa = block_sum(sa, red);
  sc = block_sum(sc, red);
  if (threadIdx.x == 0) {
    const float d = ext ? ext[0] : sc;
    loss_out[0] = sa / d;
    if (count_out) count_out[0] = sc;
    inv_out[0] = d > 0.f ? 1.0f / d : 0.f;          // gradients of an empty selection are 0 in torch
  }
}

__global__ void __launch_bounds__(256) ltrx_scale_by_device_kernel(float* __restrict__ g, size_t n, const float* __restrict__ inv) {
  const float s = inv[0];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= s;
}

int finalize_ratio(float* per_sum, float* per_cnt, int B, const float* ext, float* loss_out, float* count_out, float* inv_ws,
                   float* grad, size_t n_grad, hipStream_t s) {
  hipLaunchKernelGGL(ltrx_ratio_finalize_kernel, dim3(1), dim3(256), 0, s, per_sum, per_cnt, B, ext, loss_out, count_out, inv_ws);
  LTRX_LAUNCH_CHECK();
  if (grad) {
    size_t blocks = (n_grad + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ltrx_scale_by_device_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grad, n_grad, inv_ws);
    LTRX_LAUNCH_CHECK();
  }
  return LTRX_OK;
}

__device__ __forceinline__ float softplus_neg(float d) {      // log(1 + exp(-d)), BCEWithLogits at target 1
  return fmaxf(-d, 0.f) + log1pf(expf(-fabsf(d)));
}
__device__ __forceinline__ float sigmoid_neg(float d) {       // 1 / (1 + exp(d))
  return 1.0f / (1.0f + expf(d));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// RankNet: pairs (i, j), both valid, y_i > y_j;  l_ij = w_ij * log(1 + exp(-(s_i - s_j)));  loss = mean over the pairs
// of the WHOLE batch (BCEWithLogitsLoss(weight) with reduction 'mean' divides by the number of pairs, rankNet.py:79).
//   weight_mode 0: 1;  1: |y_i - y_j| (:64-66);  2: |y_i^2 - y_j^2| (:67-70)
// Thread i walks the slate once and handles both orientations of every pair it belongs to.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ltrx_ranknet_kernel(const float* __restrict__ y_pred, const float* __restrict__ y_true,
                                                           int L, float pad, int wmode, float* __restrict__ per_sum,
                                                           float* __restrict__ per_cnt, float* __restrict__ grad) {
  extern __shared__ float lds[];
  float* ss = lds;
  float* ys = lds + L;
  __shared__ float red[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    ss[i] = y_pred[(size_t)b * L + i];
    ys[i] = y_true[(size_t)b * L + i];
  }
  __syncthreads();
  float lsum = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float yi = ys[i], si = ss[i];
    float g = 0.f;
    if (yi != pad) {
      for (int j = 0; j < L; ++j) {
        const float yj = ys[j];
        if (yj == pad || yj == yi) continue;
        float w = 1.0f;
        if (wmode == 1) w = fabsf(yi - yj);
        else if (wmode == 2) w = fabsf(yi * yi - yj * yj);
        if (yi > yj) {                    // pair (i, j): d = s_i - s_j
          const float d = si - ss[j];
          lsum += w * softplus_neg(d);
          cnt += 1.0f;
          g -= w * sigmoid_neg(d);
        } else {                          // pair (j, i): d = s_j - s_i
          g += w * sigmoid_neg(ss[j] - si);
        }
      }
    }
    if (grad) grad[(size_t)b * L + i] = g;
  }
  lsum = block_sum(lsum, red);
  cnt = block_sum(cnt, red);
  if (threadIdx.x == 0) {
    per_sum[b] = lsum;
    per_cnt[b] = cnt;
  }
}

extern "C" size_t ltrx_ranknet_workspace_bytes(int B, int L) {
  (void)L;
  return (size_t)(B > 0 ? B : 0) * 2 * sizeof(float) + 64;
}

extern "C" int ltrx_ranknet_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float pad_value, int weight_mode,
                                    const float* ext_pair_count, float* loss_out, float* pair_count_out, float* grad_out,
                                    void* ws, ltrx_stream_t stream) {
  if (!y_pred || !y_true || !loss_out || !ws || B <= 0 || L <= 0 || weight_mode < 0 || weight_mode > 2) return LTRX_EINVAL;
  if (L > LTRX_MAX_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* per_sum = (float*)ws;
  float* per_cnt = per_sum + B;
  float* inv = per_cnt + B;
  hipLaunchKernelGGL(ltrx_ranknet_kernel, dim3(B), dim3(256), 2 * (size_t)L * sizeof(float), s, y_pred, y_true, L, pad_value,
                     weight_mode, per_sum, per_cnt, grad_out);
  LTRX_LAUNCH_CHECK();
  return finalize_ratio(per_sum, per_cnt, B, ext_pair_count, loss_out, pair_count_out, inv, grad_out, (size_t)B * L, s);
}

// ---------------------------------------------------------------------------------------------------------------
// BCE on probabilities (torch.nn.BCELoss semantics: log clamped at -100; backward (p - t) / max(p (1 - p), 1e-12)).
//   n == 0: bce      y_pred[B,L],   target = y_true;            divisor = #slates with a valid item       (bce.py:26-30)
//   n >= 1: ordinal  y_pred[B,L,n], target_k = [y_true >= k+1]; divisor = #valid items                    (ordinal.py:20,43-48)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ltrx_bce_kernel(const float* __restrict__ y_pred, const float* __restrict__ y_true, int L,
                                                       int n, float pad, float* __restrict__ per_sum,
                                                       float* __restrict__ per_cnt, float* __restrict__ grad) {
  __shared__ float red[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  const int nn = n > 0 ? n : 1;
  float lsum = 0.f, valid = 0.f;
  for (int e = threadIdx.x; e < L * nn; e += blockDim.x) {
    const int i = e / nn, kx = e - i * nn;
    const float y = y_true[(size_t)b * L + i];
    float g = 0.f;
    if (y != pad) {
      const float p = y_pred[((size_t)b * L + i) * nn + kx];
      const float t = n > 0 ? ((y >= (float)(kx + 1)) ? 1.0f : 0.f) : y;
      lsum -= t * fmaxf(logf(p), -100.f) + (1.0f - t) * fmaxf(logf(1.0f - p), -100.f);
      g = (p - t) / fmaxf((1.0f - p) * p, 1e-12f);
      if (kx == 0) valid += 1.0f;
    }
    if (grad) grad[((size_t)b * L + i) * nn + kx] = g;
  }
  lsum = block_sum(lsum, red);
  valid = block_sum(valid, red);
  if (threadIdx.x == 0) {
    per_sum[b] = lsum;
    per_cnt[b] = n > 0 ? valid : (valid > 0.f ? 1.0f : 0.f);
  }
}

extern "C" size_t ltrx_bce_workspace_bytes(int B, int L, int n) {
  (void)L;
  (void)n;
  return (size_t)(B > 0 ? B : 0) * 2 * sizeof(float) + 64;
}

extern "C" int ltrx_bce_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, int n, float pad_value,
                                const float* ext_count, float* loss_out, float* count_out, float* grad_out, void* ws,
                                ltrx_stream_t stream) {
  if (!y_pred || !y_true || !loss_out || !ws || B <= 0 || L <= 0 || n < 0) return LTRX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* per_sum = (float*)ws;
  float* per_cnt = per_sum + B;
  float* inv = per_cnt + B;
  hipLaunchKernelGGL(ltrx_bce_kernel, dim3(B), dim3(256), 0, s, y_pred, y_true, L, n, pad_value, per_sum, per_cnt, grad_out);
  LTRX_LAUNCH_CHECK();
  return finalize_ratio(per_sum, per_cnt, B, ext_count, loss_out, count_out, inv, grad_out, (size_t)B * L * (n > 0 ? n : 1), s);
}

// ---------------------------------------------------------------------------------------------------------------
// pointwise RMSE:  e_i = y_i - levels * p_i (valid items);  rmse_b = sqrt(sum e^2 / n_valid);  loss = mean_b rmse_b
//   d/dp_i = -levels * e_i / (n_valid * rmse_b) / B          (0/0 = NaN when the slate is fitted exactly, as torch's sqrt')
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ltrx_pointwise_rmse_kernel(const float* __restrict__ y_pred,
                                                                  const float* __restrict__ y_true, int L, float levels,
                                                                  float pad, float inv_div, float* __restrict__ per,
                                                                  float* __restrict__ grad) {
  __shared__ float red[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  float sq = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float y = y_true[(size_t)b * L + i];
    if (y == pad) continue;
    const float e = y - levels * y_pred[(size_t)b * L + i];
    sq += e * e;
    cnt += 1.0f;
  }
  sq = block_sum(sq, red);
  cnt = block_sum(cnt, red);
  const float rmse = sqrtf(sq / cnt);
  if (threadIdx.x == 0) per[b] = rmse;
  if (grad) {
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
      const float y = y_true[(size_t)b * L + i];
      float g = 0.f;
      if (y != pad) g = -levels * (y - levels * y_pred[(size_t)b * L + i]) / (cnt * rmse) * inv_div;
      grad[(size_t)b * L + i] = g;
    }
  }
}

extern "C" size_t ltrx_pointwise_rmse_workspace_bytes(int B, int L) {
  (void)L;
  return (size_t)(B > 0 ? B : 0) * sizeof(float);
}

extern "C" int ltrx_pointwise_rmse_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float no_of_levels,
                                           float pad_value, float batch_divisor, float* loss_out, float* grad_out, void* ws,
                                           ltrx_stream_t stream) {
  if (!y_pred || !y_true || !loss_out || !ws || B <= 0 || L <= 0 || !(batch_divisor > 0.f)) return LTRX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* per = (float*)ws;
  hipLaunchKernelGGL(ltrx_pointwise_rmse_kernel, dim3(B), dim3(256), 0, s, y_pred, y_true, L, no_of_levels, pad_value,
                     1.0f / batch_divisor, per, grad_out);
  LTRX_LAUNCH_CHECK();
  return ltrx_launch_finalize_sum(per, B, 1.0f / batch_divisor, loss_out, s);
}

// ---------------------------------------------------------------------------------------------------------------
// binary ListNet: T = y / sum(y) (sum over valid items; a zero sum divides by 1), P = softmax(scores | padded -> -inf)
//   loss = mean_b( -sum_i T_i log(P_i + eps) );   d/ds_k = (1/B) [ P_k sum_i T_i P_i/(P_i+eps) - T_k P_k/(P_k+eps) ]
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ltrx_binary_listnet_kernel(const float* __restrict__ y_pred,
                                                                  const float* __restrict__ y_true, int L, float eps, float pad,
                                                                  float inv_div, float* __restrict__ per,
                                                                  float* __restrict__ grad) {
  extern __shared__ float lds[];
  float* ps = lds;
  float* ts = lds + L;
  __shared__ float red[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  float smax = -INFINITY, ysum = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float y = y_true[(size_t)b * L + i];
    const bool valid = (y != pad);
    const float s = valid ? y_pred[(size_t)b * L + i] : -INFINITY;
    ps[i] = s;
    ts[i] = valid ? y : 0.f;
    smax = fmaxf(smax, s);
    ysum += valid ? y : 0.f;
  }
  smax = block_max(smax, red);
  ysum = block_sum(ysum, red);
  float ssum = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float e = (ps[i] == -INFINITY) ? 0.f : expf(ps[i] - smax);
    ps[i] = e;
    ssum += e;
  }
  ssum = block_sum(ssum, red);
  const float inv_s = ssum > 0.f ? 1.0f / ssum : 0.f;         // fully padded slate: contribution defined as 0 (cf. listNet)
  const float norm = (ysum == 0.f) ? 1.0f : ysum;
  float lsum = 0.f, rsum = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float P = ps[i] * inv_s, T = ts[i] / norm;
    ps[i] = P;
    ts[i] = T;
    if (T != 0.f) lsum += T * logf(P + eps);
    rsum += T * (P / (P + eps));
  }
  lsum = block_sum(lsum, red);
  rsum = block_sum(rsum, red);
  if (threadIdx.x == 0) per[b] = -lsum;
  if (grad)
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
      const float P = ps[i], T = ts[i];
      const float r = (P > 0.f) ? P / (P + eps) : 0.f;
      grad[(size_t)b * L + i] = (P * rsum - T * r) * inv_div;
    }
}

extern "C" size_t ltrx_binary_listnet_workspace_bytes(int B, int L) {
  (void)L;
  return (size_t)(B > 0 ? B : 0) * sizeof(float);
}

extern "C" int ltrx_binary_listnet_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps, float pad_value,
                                           float batch_divisor, float* loss_out, float* grad_out, void* ws,
                                           ltrx_stream_t stream) {
  if (!y_pred || !y_true || !loss_out || !ws || B <= 0 || L <= 0 || !(batch_divisor > 0.f)) return LTRX_EINVAL;
  if (L > LTRX_MAX_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* per = (float*)ws;
  hipLaunchKernelGGL(ltrx_binary_listnet_kernel, dim3(B), dim3(256), 2 * (size_t)L * sizeof(float), s, y_pred, y_true, L, eps,
                     pad_value, 1.0f / batch_divisor, per, grad_out);
  LTRX_LAUNCH_CHECK();
  return ltrx_launch_finalize_sum(per, B, 1.0f / batch_divisor, loss_out, s);
}

// ---------------------------------------------------------------------------------------------------------------
// MRR@ats (metrics.py:80-113): labels gathered in STABLE descending order of the masked predictions (padded predictions
// -inf, padded labels 0); (value, index) = first maximum of that sequence; 1/(index+1) if index < at else 0.  The
// reference zeroes the WHOLE result when the batch sum of the maxima is 0 (metrics.py:108-109: a 0-dim mask) -- kept.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ltrx_mrr_slate_kernel(const float* __restrict__ y_pred, const float* __restrict__ y_true,
                                                             int L, float pad, float* __restrict__ best_val,
                                                             int* __restrict__ best_idx) {
  extern __shared__ float lds[];
  float* ss = lds;
  float* ys = lds + L;
  __shared__ float red[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  float ymax = -INFINITY;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float y = y_true[(size_t)b * L + i];
    const bool valid = (y != pad);
    ss[i] = valid ? y_pred[(size_t)b * L + i] : -INFINITY;
    ys[i] = valid ? y : 0.f;
    ymax = fmaxf(ymax, ys[i]);
  }
  ymax = block_max(ymax, red);
  int best = L;                                   // smallest rank among the items that carry the maximum label
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    if (ys[i] != ymax) continue;
    const float si = ss[i];
    int r = 0;
    for (int j = 0; j < L; ++j) {
      const float sj = ss[j];
      r += (sj > si) || (sj == si && j < i);
    }
    best = min(best, r);
  }
  // block min via the negated max
  const float mb = block_max(-(float)best, red);
  if (threadIdx.x == 0) {
    best_val[b] = ymax;
    best_idx[b] = (int)(-mb);
  }
}

__global__ void __launch_bounds__(256) ltrx_mrr_finalize_kernel(const float* __restrict__ best_val, const int* __restrict__ best_idx,
                                                                int B, LtrxAts ats, float* __restrict__ out) {
  const int n_ats = ats.n;
  __shared__ float red[LTRX_MAX_WAVES];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) acc += best_val[b];
  const float tot = block_sum(acc, red);
  for (int e = threadIdx.x; e < B * n_ats; e += blockDim.x) {
    const int b = e / n_ats, a = e - b * n_ats;
    const int idx = best_idx[b];
    float r = (tot == 0.f) ? 0.f : 1.0f / ((float)idx + 1.0f);
    out[e] = (idx < ats.at[a]) ? r : 0.f;
  }
}

extern "C" size_t ltrx_mrr_workspace_bytes(int B, int L, int n_ats) {
  (void)L;
  (void)n_ats;
  return (size_t)(B > 0 ? B : 0) * 8 + 64;
}

extern "C" int ltrx_mrr_at(const float* y_pred, const float* y_true, int B, int L, const int* ats, int n_ats, float pad_value,
                           float* mrr_out, void* ws, ltrx_stream_t stream) {
  if (!y_pred || !y_true || !ats || !mrr_out || !ws || B <= 0 || L <= 0 || n_ats <= 0) return LTRX_EINVAL;
  if (n_ats > LTRX_MAX_ATS || L > LTRX_MAX_METRIC_SLATE_LEN) return LTRX_EUNSUPPORTED;   // 8 B of LDS per item: 64 KB at the limit
  hipStream_t s = (hipStream_t)stream;
  float* bv = (float*)ws;
  int* bi = (int*)(bv + B);
  LtrxAts dats;
  dats.n = n_ats;
  for (int i = 0; i < n_ats; ++i) dats.at[i] = ats[i];
  hipLaunchKernelGGL(ltrx_mrr_slate_kernel, dim3(B), dim3(256), 2 * (size_t)L * sizeof(float), s, y_pred, y_true, L, pad_value,
                     bv, bi);
  LTRX_LAUNCH_CHECK();
  hipLaunchKernelGGL(ltrx_mrr_finalize_kernel, dim3(1), dim3(256), 0, s, bv, bi, B, dats, mrr_out);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
