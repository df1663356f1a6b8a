// Fused ApproxNDCG forward + backward.  Reference: allrank/models/losses/approxNDCG.py:7-53.
//
//   approx_pos_i = 1 + sum_{j != i, both valid} max(sigmoid(-alpha (s_i - s_j)), eps)
//   loss = -mean_b sum_i [ (2^{y_i} - 1) / maxDCG_b ] / log2(1 + approx_pos_i),   maxDCG clamped to >= eps
//
// The reference sorts by prediction first (approxNDCG.py:27-31); the VALUE is invariant to that sort (a sum
// over items of a function of the multiset of pairwise differences), so the kernel works in the original item
// order and needs only the label ranks for maxDCG.  Ranks come from a counting rank
// (rank_i = #{j: y_j > y_i or (y_j == y_i and j < i)}), i.e. a stable descending sort, done out of LDS with
// broadcast reads -- O(L^2) like the loss itself, no barriers inside.
//
// One workgroup per slate.  The L x L sigmoid pair matrix is never materialised: each thread owns items
// i = tid, tid+T, ... and streams the slate's scores from LDS (all lanes read the same address -> broadcast).
// Pass 1 computes approx_pos_i and w_i = d loss_b / d approx_pos_i; pass 2 the gradient
//   d loss_b / d s_k = alpha * sum_{j != k} sig'(z_kj) * ( [1 - sig_kj >= eps] w_j - [sig_kj >= eps] w_k ),  z_kj = -alpha (s_k - s_j).
// Algorithmic HBM bytes: 8 B/item in, 4 B/item out; ~2 L exp per item -> bound by the transcendental (VALU) rate.
#include "ltrx_device.h"

using namespace ltrx;

__global__ void __launch_bounds__(256) ltrx_approxndcg_kernel(const float* __restrict__ y_pred,
                                                              const float* __restrict__ y_true, int L, float eps,
                                                              float pad, float alpha, float inv_div,
                                                              float* __restrict__ per_ws, float* __restrict__ per_out,
                                                              float* __restrict__ grad) {
  extern __shared__ float lds[];
  float* ss = lds;          // [L] scores
  float* ys = lds + L;      // [L] labels (pad kept as pad)
  float* ws = lds + 2 * L;  // [L] w_i
  __shared__ float red[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  const float* sp = y_pred + (size_t)b * L;
  const float* yp = y_true + (size_t)b * L;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    ss[i] = sp[i];
    ys[i] = yp[i];
    ws[i] = 0.f;
  }
  __syncthreads();

  // ---- maxDCG: ideal DCG over all positions (approxNDCG.py:43); padded labels clamp to 0 -> zero gain ----
  float dsum = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float yi = ys[i];
    if (yi == pad) continue;
    int rank = 0;
    for (int j = 0; j < L; ++j) {
      const float yj = ys[j];
      rank += (yj != pad) && ((yj > yi) || (yj == yi && j < i));
    }
    dsum += (exp2f(fmaxf(yi, 0.f)) - 1.0f) / log2f(2.0f + (float)rank);
  }
  const float maxdcg = fmaxf(block_sum(dsum, red), eps);

  // ---- pass 1: approx positions, per-slate value, w_i ----
  float vsum = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float yi = ys[i];
    if (yi == pad) continue;
    const float si = ss[i];
    float pos = 0.f;
    for (int j = 0; j < L; ++j) {
      if (j == i || ys[j] == pad) continue;
      const float sg = 1.0f / (1.0f + expf(alpha * (si - ss[j])));   // sigmoid(-alpha (s_i - s_j))
      pos += fmaxf(sg, eps);
    }
    pos += 1.0f;
    const float G = (exp2f(fmaxf(yi, 0.f)) - 1.0f) / maxdcg;
    const float aD = log2f(1.0f + pos);
    vsum += G / aD;
    ws[i] = G / (aD * aD * (1.0f + pos) * 0.6931471805599453f);
  }
  vsum = block_sum(vsum, red);   // (contains the barriers that publish ws[])
  if (threadIdx.x == 0) {
    per_ws[b] = vsum;
    if (per_out) per_out[b] = vsum;
  }
  if (!grad) return;

  // ---- pass 2: gradient ----
  float* gp = grad + (size_t)b * L;
  for (int k = threadIdx.x; k < L; k += blockDim.x) {
    const float yk = ys[k];
    if (yk == pad) {
      gp[k] = 0.f;
      continue;
    }
    const float sk = ss[k], wk = ws[k];
    float acc = 0.f;
    for (int j = 0; j < L; ++j) {
      if (j == k || ys[j] == pad) continue;
      const float sg = 1.0f / (1.0f + expf(alpha * (sk - ss[j])));
      const float ds = sg * (1.0f - sg);
      const float a = (1.0f - sg >= eps) ? ws[j] : 0.f;
      const float c = (sg >= eps) ? wk : 0.f;
      acc += ds * (a - c);
    }
    gp[k] = alpha * acc * inv_div;
  }
}

extern "C" size_t ltrx_approxndcg_workspace_bytes(int B, int L) { (void)L; return (size_t)(B > 0 ? B : 0) * sizeof(float); }

extern "C" int ltrx_approxndcg_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps,
                                       float pad_value, float alpha, float batch_divisor, float* loss_out,
                                       float* per_slate_out, float* grad_out, void* ws, ltrx_stream_t stream) {
  if (!y_pred || !y_true || !loss_out || !ws || B <= 0 || L <= 0 || !(batch_divisor > 0.f)) return LTRX_EINVAL;
  if (L > LTRX_MAX_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* per = (float*)ws;
  hipLaunchKernelGGL(ltrx_approxndcg_kernel, dim3(B), dim3(256), 3 * (size_t)L * sizeof(float), s, y_pred, y_true, L,
                     eps, pad_value, alpha, 1.0f / batch_divisor, per, per_slate_out, grad_out);
  LTRX_LAUNCH_CHECK();
  return ltrx_launch_finalize_sum(per, B, -1.0f / batch_divisor, loss_out, s);
}
