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

__global__ void __launch_bounds__(256) ltrx_listnet_kernel(const float* __restrict__ y_pred,
                                                           const float* __restrict__ y_true, int L, float eps,
                                                           float pad, float inv_div, float* __restrict__ per_ws,
                                                           float* __restrict__ per_out, float* __restrict__ grad) {
  extern __shared__ float lds[];
  float* ps = lds;       // [L] exp(s - max)  -> P
  float* ts = lds + L;   // [L] exp(y - max)  -> T
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

extern "C" size_t ltrx_listnet_workspace_bytes(int B, int L) { (void)L; return (size_t)(B > 0 ? B : 0) * sizeof(float); }

extern "C" int ltrx_listnet_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps, float pad_value,
                                    float batch_divisor, float* loss_out, float* per_slate_out, float* grad_out,
                                    void* ws, ltrx_stream_t stream) {
  if (!y_pred || !y_true || !loss_out || !ws || B <= 0 || L <= 0 || !(batch_divisor > 0.f)) return LTRX_EINVAL;
  if (L > LTRX_MAX_SLATE_LEN) return LTRX_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* per = (float*)ws;
  hipLaunchKernelGGL(ltrx_listnet_kernel, dim3(B), dim3(256), 2 * (size_t)L * sizeof(float), s, y_pred, y_true, L, eps,
                     pad_value, 1.0f / batch_divisor, per, per_slate_out, grad_out);
  LTRX_LAUNCH_CHECK();
  return ltrx_launch_finalize_sum(per, B, 1.0f / batch_divisor, loss_out, s);
}
