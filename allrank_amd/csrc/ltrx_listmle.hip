// Fused ListMLE forward + backward.  Reference: allrank/models/losses/listMLE.py:7-38.
//
//   shuffle columns by perm (listMLE.py:17-19, the permutation is an explicit input here);
//   sort y_true descending (:21) -- STABLE in the shuffled order (tie policy);
//   x = preds in that order, padded -> -inf (:25-26); xm = x - max(x) (:28-30);
//   C_r = sum_{r' >= r} exp(xm_r') (:32);  obs_r = log(C_r + eps) - xm_r, padded -> 0 (:34-36);
//   loss = mean_b sum_r obs_r (:38).
//   gradient:  d/d xm_k = exp(xm_k) * sum_{r <= k, valid} 1/(C_r + eps) - 1, and the max-shift sends
//   -sum_k(d/d xm_k) to the arg-max element (torch.max backward); both are implemented.
//
// One workgroup per slate.  The stable sort is a counting rank out of LDS (key = label desc, shuffled
// position asc): rank_i = #{j : y_j > y_i or (y_j == y_i and pos_j < pos_i)}; the suffix/prefix sums are
// workgroup scans (per-thread serial chunk + wave shuffles).  HBM: 8 B/item in (+8 B perm, L2-resident),
// 4 B/item out (+8 B/item when order_out is requested).
#include "ltrx_device.h"

using namespace ltrx;

// GWS: the six work arrays live in a global workspace (slates too long for LDS; ltrx_device.h)
template <bool GWS>
__global__ void __launch_bounds__(1024) ltrx_listmle_kernel(const float* __restrict__ y_pred,
                                                           const float* __restrict__ y_true,
                                                           const int64_t* __restrict__ perm, int L, float eps,
                                                           float pad, float inv_div, float* __restrict__ per_ws,
                                                           float* __restrict__ per_out, float* __restrict__ grad,
                                                           int64_t* __restrict__ order_out, float* gws, size_t gws_stride) {
  extern __shared__ float lds[];
  float* base = GWS ? gws + (size_t)blockIdx.x * gws_stride : lds;
  float* ys = base;                  // [L] labels by original index
  float* xs = base + L;              // [L] preds in sorted order (-inf for padded)
  float* es = base + 2 * L;          // [L] exp(xm) -> suffix sums C
  float* qs = base + 3 * L;          // [L] 1/(C+eps) -> prefix sums
  int* pos = (int*)(base + 4 * L);   // [L] shuffled position of original item i (inverse of perm)
  int* ord = (int*)(base + 5 * L);   // [L] original item index at sorted position r
  __shared__ float red[LTRX_MAX_WAVES];
  __shared__ int redi[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  const float* sp = y_pred + (size_t)b * L;
  const float* yp = y_true + (size_t)b * L;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    ys[i] = yp[i];
    pos[(int)perm[i]] = i;          // perm[p] = original index shown at shuffled position p
  }
  __syncthreads();
  // ---- stable descending sort of the shuffled labels by counting rank ----
  float xmax = -INFINITY;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float yi = ys[i];
    const int pi = pos[i];
    int rank = 0;
    for (int j = 0; j < L; ++j) {
      const float yj = ys[j];
      rank += (yj > yi) || (yj == yi && pos[j] < pi);
    }
    const float x = (yi == pad) ? -INFINITY : sp[i];
    xs[rank] = x;
    ord[rank] = i;
    xmax = fmaxf(xmax, x);
  }
  xmax = block_max(xmax, red);      // barriers inside publish xs/ord
  // arg-max position (smallest sorted index attaining the max), for the max-shift gradient
  int amax = L;
  for (int r = threadIdx.x; r < L; r += blockDim.x)
    if (xs[r] == xmax) amax = min(amax, r);
  {
    int v = amax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if (lane_id() == 0) redi[wave_id()] = v;
    __syncthreads();
    amax = redi[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) amax = min(amax, redi[w]);
  }
  // ---- suffix sums of exp(xm): scan the reversed array ----
  for (int r = threadIdx.x; r < L; r += blockDim.x) {
    const float x = xs[L - 1 - r];
    es[r] = (x == -INFINITY) ? 0.f : expf(x - xmax);
  }
  __syncthreads();
  block_inclusive_scan(es, L, red);   // es[r] = C_{L-1-r}
  float lsum = 0.f;
  for (int r = threadIdx.x; r < L; r += blockDim.x) {
    const float x = xs[r];
    const bool valid = (x != -INFINITY);
    const float C = es[L - 1 - r];
    if (valid) lsum += logf(C + eps) - (x - xmax);
    qs[r] = valid ? 1.0f / (C + eps) : 0.f;
  }
  lsum = block_sum(lsum, red);
  if (threadIdx.x == 0) {
    per_ws[b] = lsum;
    if (per_out) per_out[b] = lsum;
  }
  if (order_out) {
    int64_t* op = order_out + (size_t)b * L;
    for (int r = threadIdx.x; r < L; r += blockDim.x) op[r] = ord[r];
  }
  if (!grad) return;
  block_inclusive_scan(qs, L, red);   // qs[r] = sum_{r' <= r, valid} 1/(C_r'+eps)   (block_sum above barrier'd qs)
  float gsum = 0.f;
  for (int r = threadIdx.x; r < L; r += blockDim.x) {
    const float x = xs[r];
    float g = 0.f;
    if (x != -INFINITY) g = expf(x - xmax) * qs[r] - 1.0f;
    es[r] = g;                        // reuse es as the per-position gradient (every thread is past reading es: block_sum barrier'd)
    gsum += g;
  }
  gsum = block_sum(gsum, red);
  float* gp = grad + (size_t)b * L;
  for (int r = threadIdx.x; r < L; r += blockDim.x) {
    float g = es[r];
    if (r == amax) g -= gsum;
    gp[ord[r]] = (xs[r] == -INFINITY) ? 0.f : g * inv_div;
  }
}

static size_t listmle_per_floats(int B) { return ((size_t)(B > 0 ? B : 0) + 3) & ~(size_t)3; }
extern "C" size_t ltrx_listmle_workspace_bytes(int B, int L) {
  return (listmle_per_floats(B) + ltrx_array_ws_floats(6, 0, B > 0 ? B : 0, L > 0 ? L : 0)) * sizeof(float);
}

extern "C" int ltrx_listmle_fwd_bwd(const float* y_pred, const float* y_true, const int64_t* perm, int B, int L,
                                    float eps, float pad_value, float batch_divisor, float* loss_out,
                                    float* per_slate_out, float* grad_out, int64_t* order_out, void* ws,
                                    ltrx_stream_t stream) {
  if (!y_pred || !y_true || !perm || !loss_out || !ws || B <= 0 || L <= 0 || !(batch_divisor > 0.f)) return LTRX_EINVAL;
  if (L > LTRX_MAX_LONG_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* per = (float*)ws;
  const dim3 block(L > 512 ? 1024 : 256);                /* long slates: 16 waves */
  if (ltrx_arrays_in_lds(6, 0, L)) {
    const size_t lds = 6 * (size_t)L * sizeof(float);
    if (lds > 48 * 1024) {
      static std::atomic<uint64_t> attr_done{0};
      const int arc = ltrx_once_per_device(attr_done, []() {
        return hipFuncSetAttribute((const void*)ltrx_listmle_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   LTRX_LDS_ARRAY_BUDGET_BYTES) == hipSuccess ? LTRX_OK : LTRX_EHIP;
      });
      if (arc != LTRX_OK) return arc;
    }
    hipLaunchKernelGGL(ltrx_listmle_kernel<false>, dim3(B), block, lds, s, y_pred, y_true, perm, L, eps, pad_value, 1.0f / batch_divisor,
                       per, per_slate_out, grad_out, order_out, (float*)nullptr, (size_t)0);
  } else {
    hipLaunchKernelGGL(ltrx_listmle_kernel<true>, dim3(B), block, 0, s, y_pred, y_true, perm, L, eps, pad_value, 1.0f / batch_divisor, per,
                       per_slate_out, grad_out, order_out, per + listmle_per_floats(B), ltrx_array_ws_stride(6, 0, L));
  }
  LTRX_LAUNCH_CHECK();
  return ltrx_launch_finalize_sum(per, B, 1.0f / batch_divisor, loss_out, s);
}
