// libltrx: version + the tiny cross-slate reductions shared by every loss.
#include "ltrx_device.h"

extern "C" int ltrx_version(void) { return LTRX_VERSION; }

// out[0] = scale * sum_b per[b], summed by ONE wave-striped block in a fixed order (deterministic).
__global__ void __launch_bounds__(256) ltrx_finalize_sum_kernel(const float* __restrict__ per, int B, float scale,
                                                                float* __restrict__ out) {
  __shared__ float red[LTRX_MAX_WAVES];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) acc += per[b];
  float tot = ltrx::block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = scale * tot;
}

int ltrx_launch_finalize_sum(const float* per, int B, float scale, float* out, hipStream_t s) {
  hipLaunchKernelGGL(ltrx_finalize_sum_kernel, dim3(1), dim3(256), 0, s, per, B, scale, out);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

__global__ void ltrx_div_by_device_scalar_kernel(float* __restrict__ x, const float* __restrict__ denom) {
  const float d = denom[0];
  x[0] = (d != 0.f) ? x[0] / d : 0.f;
}

int ltrx_launch_div_by_device_scalar(float* x, const float* denom, hipStream_t s) {
  hipLaunchKernelGGL(ltrx_div_by_device_scalar_kernel, dim3(1), dim3(1), 0, s, x, denom);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
