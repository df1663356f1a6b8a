// NDCG@k / DCG@k metric + stable argsort indices.  Reference: allrank/models/metrics.py:7-77.
//
//   padded preds -> -inf, padded labels -> 0 (metrics.py:32-35); sort preds descending (:37); gather labels;
//   gain 2^y - 1 (:67) * 1/log2(pos+2) (:64-65); cumulative sum (:71) picked at ats-1 (:73-75);
//   ndcg = dcg / idcg with idcg = dcg(y_true, y_true), idcg == 0 -> filler_value (:21-24).
//
// One workgroup per slate.  Sorting = counting rank out of LDS (stable descending, padded last in index order);
// discounted gains are scattered to their sorted position in LDS and summed with a workgroup prefix scan, so
// a single pass serves every cut-off in `ats`.  The int64 order tensor (bit-exact under the tie policy of
// SURVEY.md §9.2) is optional.  HBM: 8 B/item in, 4*n_ats B/slate out (+8 B/item with order_out).
#include "ltrx_device.h"

using namespace ltrx;


__global__ void __launch_bounds__(1024) ltrx_ndcg_kernel(const float* __restrict__ y_pred,
                                                        const float* __restrict__ y_true, int L, float pad,
                                                        float filler, LtrxAts ats, float* __restrict__ ndcg_out,
                                                        float* __restrict__ dcg_out, int64_t* __restrict__ order_out,
                                                        const float* __restrict__ gains) {
  extern __shared__ float lds[];
  float* ss = lds;           // [L]
  float* ys = lds + L;       // [L]
  float* dg = lds + 2 * L;   // [L] discounted gains in predicted order -> prefix sums
  float* ig = lds + 3 * L;   // [L] discounted gains in ideal order -> prefix sums
  __shared__ float red[LTRX_MAX_WAVES];
  __shared__ int redi[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  const float* sp = y_pred + (size_t)b * L;
  const float* yp = y_true + (size_t)b * L;
  // caller-supplied gains (metrics.py:67 with gain_function != 2^x - 1): gain_function evaluated per item on the masked labels
  // (padded -> label 0, metrics.py:35), so a padded item carries gain_function(0) at its tail position in BOTH rankings
  const float* gp = gains ? gains + (size_t)b * L : nullptr;
  int nv = 0;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    ss[i] = sp[i];
    const float y = yp[i];
    ys[i] = y;
    dg[i] = 0.f;
    ig[i] = 0.f;
    nv += (y != pad);
  }
  nv = block_sum_i(nv, redi);
  int64_t* op = order_out ? order_out + (size_t)b * L : nullptr;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float yi = ys[i];
    if (yi == pad) {
      if (op || gp) {
        int before = 0;
        for (int j = 0; j < i; ++j) before += (ys[j] == pad);
        if (op) op[nv + before] = i;
        if (gp) dg[nv + before] = ig[nv + before] = gp[i] / log2f((float)(nv + before) + 2.0f);
      }
      continue;
    }
    const float si = ss[i];
    int rs = 0, ry = 0;
    for (int j = 0; j < L; ++j) {
      const float yj = ys[j];
      if (yj == pad) continue;
      const float sj = ss[j];
      rs += (sj > si) || (sj == si && j < i);
      ry += (yj > yi) || (yj == yi && j < i);
    }
    const float gain = gp ? gp[i] : exp2f(yi) - 1.0f;
    dg[rs] = gain / log2f((float)rs + 2.0f);
    ig[ry] = gain / log2f((float)ry + 2.0f);
    if (op) op[rs] = i;
  }
  __syncthreads();
  block_inclusive_scan(dg, L, red);
  block_inclusive_scan(ig, L, red);
  if (threadIdx.x < ats.n) {
    int at = ats.at[threadIdx.x];
    at = at > L ? L : at;
    const float d = dg[at - 1], id = ig[at - 1];
    ndcg_out[(size_t)b * ats.n + threadIdx.x] = (id == 0.f) ? filler : d / id;
    if (dcg_out) dcg_out[(size_t)b * ats.n + threadIdx.x] = d;
  }
}

extern "C" size_t ltrx_ndcg_workspace_bytes(int B, int L) { (void)B; (void)L; return 0; }

static int ndcg_launch(const float* y_pred, const float* y_true, const float* gains, int B, int L, const int* ats, int n_ats,
                       float pad_value, float filler_value, float* ndcg_out, float* dcg_out, int64_t* order_out,
                       void* ws, ltrx_stream_t stream) {
  (void)ws;
  if (!y_pred || !y_true || !ats || !ndcg_out || B <= 0 || L <= 0 || n_ats <= 0) return LTRX_EINVAL;
  if (n_ats > LTRX_MAX_ATS || L > LTRX_MAX_METRIC_SLATE_LEN) return LTRX_EUNSUPPORTED;
  if (L > 4096) {                                   // more than the default 64 KB of dynamic LDS
    static std::atomic<uint64_t> attr_done{0};
    const int arc = ltrx_once_per_device(attr_done, []() {
      return hipFuncSetAttribute((const void*)ltrx_ndcg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(4 * LTRX_MAX_METRIC_SLATE_LEN * sizeof(float))) == hipSuccess ? LTRX_OK : LTRX_EHIP;
    });
    if (arc != LTRX_OK) return arc;
  }
  LtrxAts a;
  a.n = n_ats;
  for (int i = 0; i < n_ats; ++i) {
    if (ats[i] <= 0) return LTRX_EINVAL;
    a.at[i] = ats[i];
  }
  hipLaunchKernelGGL(ltrx_ndcg_kernel, dim3(B), dim3(L > 512 ? 1024 : 256) /* long slates: 16 waves */, 4 * (size_t)L * sizeof(float), (hipStream_t)stream, y_pred,
                     y_true, L, pad_value, filler_value, a, ndcg_out, dcg_out, order_out, gains);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

extern "C" int ltrx_ndcg_at(const float* y_pred, const float* y_true, int B, int L, const int* ats, int n_ats,
                            float pad_value, float filler_value, float* ndcg_out, float* dcg_out, int64_t* order_out,
                            void* ws, ltrx_stream_t stream) {
  return ndcg_launch(y_pred, y_true, nullptr, B, L, ats, n_ats, pad_value, filler_value, ndcg_out, dcg_out, order_out, ws, stream);
}

extern "C" int ltrx_ndcg_at_gains(const float* y_pred, const float* y_true, const float* gains, int B, int L, const int* ats,
                                  int n_ats, float pad_value, float filler_value, float* ndcg_out, float* dcg_out,
                                  int64_t* order_out, void* ws, ltrx_stream_t stream) {
  if (!gains) return LTRX_EINVAL;
  return ndcg_launch(y_pred, y_true, gains, B, L, ats, n_ats, pad_value, filler_value, ndcg_out, dcg_out, order_out, ws, stream);
}
