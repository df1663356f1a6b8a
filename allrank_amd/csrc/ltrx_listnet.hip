// Fused ListNet forward + backward.  Reference: allrank/models/losses/listNet.py:8-30.
//
//   mask = y_true == pad; P = softmax(y_pred | mask -> -inf); T = softmax(y_true | mask -> -inf)
//   loss = mean_b( -sum_i T_i * log(P_i + eps) )
//   d loss / d s_k = (1/B) [ P_k * sum_i T_i P_i/(P_i+eps)  -  T_k P_k/(P_k+eps) ]        
//
// One workgroup (256 threads = 4 waves) per slate; scores/labels are read once from HBM (coalesced,
// lane-strided), kept in registers/LDS, all reductions are wave shuffles + a 4-entry LDS combine.
// Algorithmic HBM traffic: 8 B/item read + 4 B/item written; compute is negligible -> launch/latency bound.
#include "ltrx_device.h"

using namespace ltrx;

// GWS: the two work arrays live in a global workspace (slates too long for LDS; ltrx_device.h)
template <bool GWS>
__global__ void __launch_bounds__(256) ltrx_listnet_kernel(const float* __restrict__ y_pred,
                                                           const float* __restrict__ y_true, int L, float eps,
                                                           float pad, float inv_div, float* __restrict__ per_ws,
                                                           float* __restrict__ per_out, float* __restrict__ grad,
                                                           float* gws, size_t gws_stride) {
  extern __shared__ float lds[];
  float* base = GWS ? gws + (size_t)blockIdx.x * gws_stride : lds;
  float* ps = base;      // [L] exp(s - max)  -> P
  float* ts = base + L;  // [L] exp(y - max)  -> T
  __shared__ float red[LTRX_MAX_WAVES];
  const int b = blockIdx.x;
  const float* sp = y_pred + (size_t)b * L;
  const float* yp = y_true + (size_t)b * L;

  float smax = -INFINITY, ymax = -INFINITY;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    float y = yp[i];
    bool valid = (y != pad);
    float s = valid ? sp[i] : -INFINITY;
    float t = valid ? y : -INFINITY;
    ps[i] = s;
    ts[i] = t;
    smax = fmaxf(smax, s);
    ymax = fmaxf(ymax, t);
  }
  smax = block_max(smax, red);
  ymax = block_max(ymax, red);
  float ssum = 0.f, ysum = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    float e = (ps[i] == -INFINITY) ? 0.f : expf(ps[i] - smax);
    float f = (ts[i] == -INFINITY) ? 0.f : expf(ts[i] - ymax);
    ps[i] = e;
    ts[i] = f;
    ssum += e;
    ysum += f;
  }
  ssum = block_sum(ssum, red);
  ysum = block_sum(ysum, red);
  // a fully padded slate has ssum == 0: the reference yields NaN there; we define its contribution as 0.
  const float inv_s = ssum > 0.f ? 1.0f / ssum : 0.f;
  const float inv_y = ysum > 0.f ? 1.0f / ysum : 0.f;
  float lsum = 0.f, rsum = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    float P = ps[i] * inv_s, T = ts[i] * inv_y;
    ps[i] = P;
    ts[i] = T;
    if (T > 0.f) lsum += T * logf(P + eps);
    rsum += T * (P / (P + eps));   // P == 0 -> 0 (eps > 0); eps == 0 and P == 0 only when T == 0 as well
  }
  lsum = block_sum(lsum, red);
  rsum = block_sum(rsum, red);
  if (threadIdx.x == 0) {
    per_ws[b] = -lsum;
    if (per_out) per_out[b] = -lsum;
  }
  if (grad) {
    float* gp = grad + (size_t)b * L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
      float P = ps[i], T = ts[i];
      float r = (P > 0.f) ? P / (P + eps) : 0.f;
      gp[i] = (P * rsum - T * r) * inv_div;   // padded: P = T = 0 -> exactly 0
    }
  }
}

static size_t listnet_per_floats(int B) { return ((size_t)(B > 0 ? B : 0) + 3) & ~(size_t)3; }
extern "C" size_t ltrx_listnet_workspace_bytes(int B, int L) {
  return (listnet_per_floats(B) + ltrx_array_ws_floats(2, 0, B > 0 ? B : 0, L > 0 ? L : 0)) * sizeof(float);
}

extern "C" int ltrx_listnet_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps, float pad_value,
                                    float batch_divisor, float* loss_out, float* per_slate_out, float* grad_out,
                                    void* ws, ltrx_stream_t stream) {
  if (!y_pred || !y_true || !loss_out || !ws || B <= 0 || L <= 0 || !(batch_divisor > 0.f)) return LTRX_EINVAL;
  if (L > LTRX_MAX_LONG_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* per = (float*)ws;
  if (ltrx_arrays_in_lds(2, 0, L)) {
    const size_t lds = 2 * (size_t)L * sizeof(float);
    if (lds > 48 * 1024) {                               // more than the default dynamic-LDS allowance: once per device
      static std::atomic<uint64_t> attr_done{0};
      const int arc = ltrx_once_per_device(attr_done, []() {
        return hipFuncSetAttribute((const void*)ltrx_listnet_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   LTRX_LDS_ARRAY_BUDGET_BYTES) == hipSuccess ? LTRX_OK : LTRX_EHIP;
      });
      if (arc != LTRX_OK) return arc;
    }
    hipLaunchKernelGGL(ltrx_listnet_kernel<false>, dim3(B), dim3(256), lds, s, y_pred, y_true, L, eps, pad_value, 1.0f / batch_divisor,
                       per, per_slate_out, grad_out, (float*)nullptr, (size_t)0);
  } else {
    hipLaunchKernelGGL(ltrx_listnet_kernel<true>, dim3(B), dim3(256), 0, s, y_pred, y_true, L, eps, pad_value, 1.0f / batch_divisor, per,
                       per_slate_out, grad_out, per + listnet_per_floats(B), ltrx_array_ws_stride(2, 0, L));
  }
  LTRX_LAUNCH_CHECK();
  return ltrx_launch_finalize_sum(per, B, 1.0f / batch_divisor, loss_out, s);
}
