// Custom LayerNorm of the Annotated-Transformer encoder, forward + backward, with a fused residual add.
// Reference: allrank/models/transformer.py:59-81 (LayerNorm), :98-106 (SublayerConnection: x + sublayer(norm(x))).
//
//   mean = x.mean(-1); std = x.std(-1) (UNBIASED, n-1); y = a_2 * (x - mean) / (std + eps) + b_2      (eps on std!)
//
// Forward fuses the residual sum that precedes every norm after the first:  xsum = x + res; y = LN(xsum).
// One wave per row (lane-strided over D, wave-shuffle reductions), 4 rows per 256-thread workgroup, grid-stride
// over rows.  HBM-bound: reads 4*D (8*D with residual), writes 4*D (8*D) per row -> the roofline is HBM.
// Backward:  g = dy * a;  dx = r (g - mean(g)) - r^2 <g, xc> / ((n-1) std) * xc  (+ dres),  r = 1/(std+eps);
// da = sum_rows dy * xhat, db = sum_rows dy via per-block partial rows in `ws` + a second tiny kernel (fixed
// summation order -> deterministic, no atomics).
#include "ltrx_device.h"

using namespace ltrx;

__global__ void __launch_bounds__(256) ltrx_layernorm_fwd_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ res,
                                                                 const float* __restrict__ a,
                                                                 const float* __restrict__ b, int rows, int D, float eps,
                                                                 float* __restrict__ xsum_out, float* __restrict__ y,
                                                                 float* __restrict__ mean_out,
                                                                 float* __restrict__ rstd_out) {
  const int lane = lane_id();
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + wave_id(); row < rows; row += gridDim.x * wpb) {
    const float* xr = x + (size_t)row * D;
    const float* rr = res ? res + (size_t)row * D : nullptr;
    float* xs = xsum_out ? xsum_out + (size_t)row * D : nullptr;
    float sum = 0.f;
    for (int c = lane; c < D; c += 64) {
      float v = xr[c];
      if (rr) v += rr[c];
      if (xs) xs[c] = v;
      sum += v;
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
    for (int c = lane; c < D; c += 64) {
      float v = xr[c];
      if (rr) v += rr[c];
      const float d = v - mean;
      sq += d * d;
    }
    const float stdv = sqrtf(wave_sum(sq) / (float)(D - 1));
    const float r = 1.0f / (stdv + eps);
    float* yr = y + (size_t)row * D;
    for (int c = lane; c < D; c += 64) {
      float v = xr[c];
      if (rr) v += rr[c];
      yr[c] = a[c] * ((v - mean) * r) + b[c];
    }
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = r;
    }
  }
}

// partial[blk][0][c] = sum over the block's rows of dy*xhat, partial[blk][1][c] = sum of dy
__global__ void __launch_bounds__(256) ltrx_layernorm_bwd_kernel(const float* __restrict__ dy,
                                                                 const float* __restrict__ xsum,
                                                                 const float* __restrict__ a,
                                                                 const float* __restrict__ mean_in,
                                                                 const float* __restrict__ rstd_in,
                                                                 const float* __restrict__ dres, int rows, int D,
                                                                 float eps, float* __restrict__ dx,
                                                                 float* __restrict__ partial) {
  extern __shared__ float lds[];   // [wpb][2][D] per-wave column partials
  const int lane = lane_id();
  const int w = wave_id();
  const int wpb = blockDim.x >> 6;
  float* my_da = lds + (size_t)w * 2 * D;
  float* my_db = my_da + D;
  for (int c = lane; c < D; c += 64) {
    my_da[c] = 0.f;
    my_db[c] = 0.f;
  }
  for (int row = blockIdx.x * wpb + w; row < rows; row += gridDim.x * wpb) {
    const float* dyr = dy + (size_t)row * D;
    const float* xr = xsum + (size_t)row * D;
    const float mean = mean_in[row];
    const float r = rstd_in[row];
    float gsum = 0.f, dot = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float g = dyr[c] * a[c];
      gsum += g;
      dot += g * (xr[c] - mean);
    }
    const float gm = wave_sum(gsum) / (float)D;
    dot = wave_sum(dot);
    const float stdv = 1.0f / r - eps;
    const float tcoef = (stdv > 0.f) ? r * r * dot / ((float)(D - 1) * stdv) : 0.f;
    float* dxr = dx + (size_t)row * D;
    const float* drr = dres ? dres + (size_t)row * D : nullptr;
    for (int c = lane; c < D; c += 64) {
      const float dyv = dyr[c];
      const float xc = xr[c] - mean;
      float v = r * (dyv * a[c] - gm) - tcoef * xc;
      if (drr) v += drr[c];
      dxr[c] = v;
      my_da[c] += dyv * (xc * r);   // lane-private LDS slots: no conflicts between lanes
      my_db[c] += dyv;
    }
  }
  __syncthreads();
  float* pa = partial + (size_t)blockIdx.x * 2 * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float sa = 0.f, sb = 0.f;
    for (int ww = 0; ww < wpb; ++ww) {
      sa += lds[(size_t)ww * 2 * D + c];
      sb += lds[(size_t)ww * 2 * D + D + c];
    }
    pa[c] = sa;
    pa[D + c] = sb;
  }
}

// One workgroup per 64 columns; its 4 waves split the partial rows, lanes own consecutive columns (coalesced),
// the 4 wave partials are combined through LDS in a fixed order (deterministic).
__global__ void __launch_bounds__(256) ltrx_layernorm_bwd_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                                        int D, float* __restrict__ da,
                                                                        float* __restrict__ db) {
  __shared__ float sa[4][64];
  __shared__ float sb[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a = 0.f, b = 0.f;
  if (c < D) {
    for (int k = w; k < nblk; k += 4) {
      a += partial[(size_t)k * 2 * D + c];
      b += partial[(size_t)k * 2 * D + D + c];
    }
  }
  sa[w][lane] = a;
  sb[w][lane] = b;
  __syncthreads();
  if (w == 0 && c < D) {
    da[c] = (sa[0][lane] + sa[1][lane]) + (sa[2][lane] + sa[3][lane]);
    db[c] = (sb[0][lane] + sb[1][lane]) + (sb[2][lane] + sb[3][lane]);
  }
}

static int ln_grid(int rows) {
  int g = (rows + 3) / 4;
  return g > 1024 ? 1024 : (g < 1 ? 1 : g);
}
// the backward keeps per-block column partials: fewer, fatter blocks (each wave walks many rows)
static int ln_bwd_grid(int rows) {
  int g = (rows + 15) / 16;
  return g > 256 ? 256 : (g < 1 ? 1 : g);
}

extern "C" int ltrx_layernorm_fwd(const float* x, const float* res, const float* a, const float* b, int rows, int D,
                                  float eps, float* xsum_out, float* y_out, float* mean_out, float* rstd_out,
                                  ltrx_stream_t stream) {
  if (!x || !a || !b || !y_out || !mean_out || !rstd_out || rows <= 0 || D < 2) return LTRX_EINVAL;
  if (res && !xsum_out) return LTRX_EINVAL;
  hipLaunchKernelGGL(ltrx_layernorm_fwd_kernel, dim3(ln_grid(rows)), dim3(256), 0, (hipStream_t)stream, x, res, a, b,
                     rows, D, eps, xsum_out, y_out, mean_out, rstd_out);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

extern "C" size_t ltrx_layernorm_bwd_workspace_bytes(int rows, int D) {
  if (rows <= 0 || D <= 0) return 0;
  return (size_t)ln_bwd_grid(rows) * 2 * D * sizeof(float);
}

extern "C" int ltrx_layernorm_bwd(const float* dy, const float* xsum, const float* a, const float* mean,
                                  const float* rstd, const float* dres_in, int rows, int D, float eps, float* dx_out,
                                  float* da_out, float* db_out, void* ws, ltrx_stream_t stream) {
  if (!dy || !xsum || !a || !mean || !rstd || !dx_out || !da_out || !db_out || !ws || rows <= 0 || D < 2) return LTRX_EINVAL;
  if ((size_t)4 * 2 * D * sizeof(float) > 64 * 1024) return LTRX_EUNSUPPORTED;   // D <= 2048
  hipStream_t s = (hipStream_t)stream;
  const int grid = ln_bwd_grid(rows);
  hipLaunchKernelGGL(ltrx_layernorm_bwd_kernel, dim3(grid), dim3(256), (size_t)4 * 2 * D * sizeof(float), s, dy, xsum, a,
                     mean, rstd, dres_in, rows, D, eps, dx_out, (float*)ws);
  LTRX_LAUNCH_CHECK();
  hipLaunchKernelGGL(ltrx_layernorm_bwd_reduce_kernel, dim3((D + 63) / 64), dim3(256), 0, s, (const float*)ws, grid, D,
                     da_out, db_out);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
