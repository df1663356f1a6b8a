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
// One workgroup of 1024 threads per slate.  The L x L sigmoid pair matrix is never materialised: thread (i, part) owns item
// i = tid & 255 (+256, ...) and a quarter of the partner range j (part = tid >> 8), streaming the slate's scores from LDS
// (all lanes of a wave read the same address -> broadcast); the four partial sums of an item are combined through LDS.
// 16 waves per slate instead of 4: the kernel runs one workgroup per CU at the bench size, so this is what hides the
// transcendental latency (120 -> ~35 us at 256 slates x 240 items).
// Pass 1 computes approx_pos_i and w_i = d loss_b / d approx_pos_i; pass 2 the gradient
//   d loss_b / d s_k = alpha * sum_{j != k} sig'(z_kj) * ( [1 - sig_kj >= eps] w_j - [sig_kj >= eps] w_k ),  z_kj = -alpha (s_k - s_j).
// Algorithmic HBM bytes: 8 B/item in, 4 B/item out; ~2 L exp per item -> bound by the transcendental (VALU) rate.
#include "ltrx_device.h"

using namespace ltrx;

namespace {
// sigmoid(-x) = 1 / (1 + e^x) with one v_exp_f32 and one v_rcp_f32 (both ~1 ulp; x = +inf -> 0, x = -inf -> 1)
__device__ __forceinline__ float sigmoid_neg_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 1.4426950408889634f));
}
}  // namespace

// GWS: the seven work arrays live in a global workspace (slates too long for LDS; ltrx_device.h)
template <bool GWS>
__global__ void __launch_bounds__(1024) ltrx_approxndcg_kernel(const float* __restrict__ y_pred,
                                                               const float* __restrict__ y_true, int L, float eps,
                                                               float pad, float alpha, float inv_div,
                                                               float* __restrict__ per_ws, float* __restrict__ per_out,
                                                               float* __restrict__ grad, float* gws, size_t gws_stride) {
  extern __shared__ float lds[];
  float* base = GWS ? gws + (size_t)blockIdx.x * gws_stride : lds;
  float* ss = base;          // [L] scores
  float* ys = base + L;      // [L] labels (pad kept as pad)
  float* ws = base + 2 * L;  // [L] w_i
  float* part = base + 3 * L;   // [4][L] partial sums of the four partner quarters
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
  const int q = threadIdx.x >> 8, i0 = threadIdx.x & 255;
  const int lq = (L + 3) >> 2;
  const int j0 = q * lq, j1 = min(L, j0 + lq);

  // ---- maxDCG: ideal DCG over all positions (approxNDCG.py:43); padded labels clamp to 0 -> zero gain ----
  for (int i = i0; i < L; i += 256) {
    const float yi = ys[i];
    int rank = 0;
    for (int j = j0; j < j1; ++j) {
      const float yj = ys[j];
      rank += (yj != pad) && ((yj > yi) || (yj == yi && j < i));
    }
    part[q * L + i] = (float)rank;
  }
  __syncthreads();
  float dsum = 0.f;
  if (q == 0)
    for (int i = i0; i < L; i += 256) {
      const float yi = ys[i];
      if (yi == pad) continue;
      const float rank = (part[i] + part[L + i]) + (part[2 * L + i] + part[3 * L + i]);
      dsum += (exp2f(fmaxf(yi, 0.f)) - 1.0f) / log2f(2.0f + rank);
    }
  const float maxdcg = fmaxf(block_sum(dsum, red), eps);      // (barriers inside: part[] may be reused below)

  // ---- pass 1: approx positions, per-slate value, w_i ----
  for (int i = i0; i < L; i += 256) {
    const float si = ss[i];
    float pos = 0.f;
    if (ys[i] != pad)
      for (int j = j0; j < j1; ++j) {
        if (j == i || ys[j] == pad) continue;
        pos += fmaxf(sigmoid_neg_fast(alpha * (si - ss[j])), eps);   // sigmoid(-alpha (s_i - s_j))
      }
    part[q * L + i] = pos;
  }
  __syncthreads();
  float vsum = 0.f;
  if (q == 0)
    for (int i = i0; i < L; i += 256) {
      const float yi = ys[i];
      if (yi == pad) continue;
      const float pos = 1.0f + ((part[i] + part[L + i]) + (part[2 * L + i] + part[3 * L + i]));
      const float G = (exp2f(fmaxf(yi, 0.f)) - 1.0f) / maxdcg;
      const float aD = log2f(1.0f + pos);
      vsum += G / aD;
      ws[i] = G / (aD * aD * (1.0f + pos) * 0.6931471805599453f);
    }
  vsum = block_sum(vsum, red);   // (contains the barriers that publish ws[] and retire part[])
  if (threadIdx.x == 0) {
    per_ws[b] = vsum;
    if (per_out) per_out[b] = vsum;
  }
  if (!grad) return;

  // ---- pass 2: gradient ----
  for (int k = i0; k < L; k += 256) {
    float acc = 0.f;
    if (ys[k] != pad) {
      const float sk = ss[k], wk = ws[k];
      for (int j = j0; j < j1; ++j) {
        if (j == k || ys[j] == pad) continue;
        const float sg = sigmoid_neg_fast(alpha * (sk - ss[j]));
        const float ds = sg * (1.0f - sg);
        const float a = (1.0f - sg >= eps) ? ws[j] : 0.f;
        const float c = (sg >= eps) ? wk : 0.f;
        acc += ds * (a - c);
      }
    }
    part[q * L + k] = acc;
  }
  __syncthreads();
  if (q == 0) {
    float* gp = grad + (size_t)b * L;
    for (int k = i0; k < L; k += 256)
      gp[k] = (ys[k] == pad) ? 0.f : alpha * ((part[k] + part[L + k]) + (part[2 * L + k] + part[3 * L + k])) * inv_div;
  }
}

static size_t approx_per_floats(int B) { return ((size_t)(B > 0 ? B : 0) + 3) & ~(size_t)3; }
extern "C" size_t ltrx_approxndcg_workspace_bytes(int B, int L) {
  return (approx_per_floats(B) + ltrx_array_ws_floats(7, 0, B > 0 ? B : 0, L > 0 ? L : 0)) * sizeof(float);
}

extern "C" int ltrx_approxndcg_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps,
                                       float pad_value, float alpha, float batch_divisor, float* loss_out,
                                       float* per_slate_out, float* grad_out, void* ws, ltrx_stream_t stream) {
  if (!y_pred || !y_true || !loss_out || !ws || B <= 0 || L <= 0 || !(batch_divisor > 0.f)) return LTRX_EINVAL;
  if (L > LTRX_MAX_LONG_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* per = (float*)ws;
  if (ltrx_arrays_in_lds(7, 0, L)) {
    const size_t lds = 7 * (size_t)L * sizeof(float);
    if (lds > 48 * 1024) {
      static std::atomic<uint64_t> attr_done{0};
      const int arc = ltrx_once_per_device(attr_done, []() {
        return hipFuncSetAttribute((const void*)ltrx_approxndcg_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   LTRX_LDS_ARRAY_BUDGET_BYTES) == hipSuccess ? LTRX_OK : LTRX_EHIP;
      });
      if (arc != LTRX_OK) return arc;
    }
    hipLaunchKernelGGL(ltrx_approxndcg_kernel<false>, dim3(B), dim3(1024), lds, s, y_pred, y_true, L, eps, pad_value, alpha,
                       1.0f / batch_divisor, per, per_slate_out, grad_out, (float*)nullptr, (size_t)0);
  } else {
    hipLaunchKernelGGL(ltrx_approxndcg_kernel<true>, dim3(B), dim3(1024), 0, s, y_pred, y_true, L, eps, pad_value, alpha,
                       1.0f / batch_divisor, per, per_slate_out, grad_out, per + approx_per_floats(B), ltrx_array_ws_stride(7, 0, L));
  }
  LTRX_LAUNCH_CHECK();
  return ltrx_launch_finalize_sum(per, B, -1.0f / batch_divisor, loss_out, s);
}
