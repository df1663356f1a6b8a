// Model options of the shipped allRank configs that sit around the encoder on the explicit (FusedTrainer) step:
//   * FCModel.input_norm = nn.LayerNorm(n_features) (allrank/models/model.py:27,39): biased variance, eps inside the sqrt;
//   * positional encodings (allrank/models/positional.py:15-77, applied at transformer.py:51-52):
//         x <- sqrt(d_model) * x + table[row],  row = padding row for padded items and for ranks >= max_len;
//     fixed sin/cos table or learned nn.Embedding (whose padding row never receives a gradient);
//   * OutputLayer activation (model.py:106-117): Sigmoid / Tanh on the scores.
// All HBM-bound one-pass kernels (4-20 bytes per element).
#include "ltrx_device.h"

using namespace ltrx;

// ------------------------------------------------------------------------------------------------------------------
// nn.LayerNorm forward: one wave per row; saves mean and rstd = 1/sqrt(var_biased + eps).  The parameter gradients
// (dw = sum dy * xhat, db = sum dy) come from ltrx_layernorm_bwd called with these statistics (its da/db formulas are
// statistics-agnostic; the input gradient is not needed: the normalised tensor is the model INPUT).
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ltrx_layernorm_torch_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                       const float* __restrict__ b, int rows, int D, float eps,
                                                                       float* __restrict__ y, float* __restrict__ mean_out,
                                                                       float* __restrict__ rstd_out) {
  const int lane = lane_id();
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + wave_id(); row < rows; row += gridDim.x * wpb) {
    const float* xr = x + (size_t)row * D;
    float sum = 0.f;
    for (int c = lane; c < D; c += 64) sum += xr[c];
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float d = xr[c] - mean;
      sq += d * d;
    }
    const float r = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
    float* yr = y + (size_t)row * D;
    for (int c = lane; c < D; c += 64) yr[c] = (xr[c] - mean) * r * w[c] + b[c];
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = r;
    }
  }
}

extern "C" int ltrx_layernorm_torch_fwd(const float* x, const float* w, const float* b, int rows, int D, float eps, float* y,
                                        float* mean_out, float* rstd_out, ltrx_stream_t stream) {
  if (!x || !w || !b || !y || !mean_out || !rstd_out || rows <= 0 || D <= 0) return LTRX_EINVAL;
  int g = (rows + 3) / 4;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(ltrx_layernorm_torch_fwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, w, b, rows, D, eps, y, mean_out,
                     rstd_out);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// positional encoding
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pe_row(const int64_t* __restrict__ indices, const uint8_t* __restrict__ mask, int m, int pad) {
  if (mask && mask[m]) return pad;
  const long long i = indices[m];
  return (i < 0 || i > pad) ? pad : (int)i;        // positional.py:44-45 (a negative rank only occurs on padded items)
}

__global__ void __launch_bounds__(256) ltrx_posenc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                              const int64_t* __restrict__ indices, const uint8_t* __restrict__ mask,
                                                              int M, int D4, int pad, float scale, float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * D4) return;
  const int m = (int)(i / D4), c = (int)(i % D4);
  const int r = pe_row(indices, mask, m, pad);
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const float4 t = reinterpret_cast<const float4*>(table)[(size_t)r * D4 + c];
  reinterpret_cast<float4*>(y)[i] = make_float4(scale * v.x + t.x, scale * v.y + t.y, scale * v.z + t.z, scale * v.w + t.w);
}

extern "C" int ltrx_posenc_fwd(const float* x, const float* table, const int64_t* indices, const uint8_t* mask, int M, int D,
                               int padding_idx, float scale, float* y, ltrx_stream_t stream) {
  if (!x || !table || !indices || !y || M <= 0 || D <= 0 || padding_idx < 0) return LTRX_EINVAL;
  if ((D & 3) || (((uintptr_t)x | (uintptr_t)table | (uintptr_t)y) & 15)) return LTRX_EUNSUPPORTED;
  const size_t n = (size_t)M * (D / 4);
  hipLaunchKernelGGL(ltrx_posenc_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, table, indices, mask,
                     M, D / 4, padding_idx, scale, y);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// dtable[r][:] = sum over the rows m whose encoding row is r of dx[m][:]  (r < padding_idx; the padding row gets 0, like
// nn.Embedding(padding_idx=...)).  One workgroup per table row: its 4 waves scan the M row indices 64 at a time (ballot),
// add the matching rows of dx in ascending m into per-wave LDS columns, and the wave partials are combined in a fixed order
// -> deterministic, no atomics.  (The item ranks of one slate are distinct, so a table row matches at most one item per slate.)
__global__ void __launch_bounds__(256) ltrx_posenc_table_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ indices,
                                                                    const uint8_t* __restrict__ mask, int M, int D, int pad,
                                                                    float* __restrict__ dtable) {
  extern __shared__ float part[];      // [4][D]
  const int r = blockIdx.x;
  const int lane = lane_id(), w = wave_id();
  float* mine = part + (size_t)w * D;
  for (int c = lane; c < D; c += 64) mine[c] = 0.f;
  if (r < pad) {
    for (int base = w * 64; base < M; base += 4 * 64) {
      const int m = base + lane;
      const bool hit = m < M && pe_row(indices, mask, m, pad) == r;
      unsigned long long bits = __ballot(hit);
      while (bits) {
        const int k = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        const float* row = dx + (size_t)(base + k) * D;
        for (int c = lane; c < D; c += 64) mine[c] += row[c];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x)
    dtable[(size_t)r * D + c] = (part[c] + part[D + c]) + (part[2 * D + c] + part[3 * D + c]);
}

extern "C" int ltrx_posenc_table_bwd(const float* dx, const int64_t* indices, const uint8_t* mask, int M, int D, int padding_idx,
                                     float* dtable, ltrx_stream_t stream) {
  if (!dx || !indices || !dtable || M <= 0 || D <= 0 || padding_idx < 0) return LTRX_EINVAL;
  if ((size_t)4 * D * sizeof(float) > 64 * 1024) return LTRX_EUNSUPPORTED;
  hipLaunchKernelGGL(ltrx_posenc_table_bwd_kernel, dim3(padding_idx + 1), dim3(256), (size_t)4 * D * sizeof(float), (hipStream_t)stream, dx,
                     indices, mask, M, D, padding_idx, dtable);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

__global__ void __launch_bounds__(256) ltrx_scale_kernel(float* __restrict__ x, size_t n, float s) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= s;
}

extern "C" int ltrx_scale_inplace(float* x, size_t n, float s, ltrx_stream_t stream) {
  if (!x) return LTRX_EINVAL;
  if (n == 0) return LTRX_OK;
  size_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(ltrx_scale_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, s);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// OutputLayer activation: kind 1 = Sigmoid, 2 = Tanh; the backward uses the activation's OUTPUT (y (1 - y), 1 - y^2)
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ltrx_out_act_fwd_kernel(const float* __restrict__ z, size_t n, int kind, float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = z[i];
  y[i] = kind == 1 ? 1.0f / (1.0f + expf(-v)) : tanhf(v);
}
__global__ void __launch_bounds__(256) ltrx_out_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, size_t n, int kind,
                                                               float* __restrict__ dz) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = y[i];
  dz[i] = dy[i] * (kind == 1 ? v * (1.0f - v) : 1.0f - v * v);
}

extern "C" int ltrx_out_act_fwd(const float* z, size_t n, int kind, float* y, ltrx_stream_t stream) {
  if (!z || !y || (kind != 1 && kind != 2)) return LTRX_EINVAL;
  if (n == 0) return LTRX_OK;
  hipLaunchKernelGGL(ltrx_out_act_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, n, kind, y);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

extern "C" int ltrx_out_act_bwd(const float* dy, const float* y, size_t n, int kind, float* dz, ltrx_stream_t stream) {
  if (!dy || !y || !dz || (kind != 1 && kind != 2)) return LTRX_EINVAL;
  if (n == 0) return LTRX_OK;
  hipLaunchKernelGGL(ltrx_out_act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, n, kind, dz);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
