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
                                                                 float* __restrict__ rstd_out, DropSpec drop,
                                                                 const uint32_t* __restrict__ drop_step) {
  const int lane = lane_id();
  const int wpb = blockDim.x >> 6;
  if (drop_step) drop.seed ^= drop_step[0] * 0x9E3779B9u;
  for (int row = blockIdx.x * wpb + wave_id(); row < rows; row += gridDim.x * wpb) {
    const float* xr = x + (size_t)row * D;
    const float* rr = res ? res + (size_t)row * D : nullptr;
    float* xs = xsum_out ? xsum_out + (size_t)row * D : nullptr;
    float sum = 0.f;
    for (int c = lane; c < D; c += 64) {
      float v = xr[c];
      if (rr) v += (drop.thresh != 0u) ? rr[c] * drop_keep_scale(drop, (uint64_t)row * D + c) : rr[c];
      if (xs) xs[c] = v;
      sum += v;
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
    for (int c = lane; c < D; c += 64) {
      float v = xs ? xs[c] : xr[c];
      const float d = v - mean;
      sq += d * d;
    }
    const float stdv = sqrtf(wave_sum(sq) / (float)(D - 1));
    const float r = 1.0f / (stdv + eps);
    float* yr = y + (size_t)row * D;
    for (int c = lane; c < D; c += 64) {
      float v = xs ? xs[c] : xr[c];
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


// ------------------------------------------------------------------------------------------------------------------
// Fast path, D = 256 * NV (NV = 1..4): the row lives in registers (NV float4 per lane, 16-byte coalesced accesses),
// one HBM read per input, statistics from registers; the backward keeps the per-lane column partials of da/db in
// registers across all rows of the wave and spills them once at the end.
// ------------------------------------------------------------------------------------------------------------------
template <int NV>
__global__ void __launch_bounds__(256) ltrx_layernorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                     const float* __restrict__ a, const float* __restrict__ b,
                                                                     int rows, float eps, float* __restrict__ xsum_out,
                                                                     float* __restrict__ y, float* __restrict__ mean_out,
                                                                     float* __restrict__ rstd_out, DropSpec drop,
                                                                     const uint32_t* __restrict__ drop_step, int yimg) {
  constexpr int D = 256 * NV;
  const int lane = lane_id(), wpb = blockDim.x >> 6;
  if (drop_step) drop.seed ^= drop_step[0] * 0x9E3779B9u;
  float4 av[NV], bv[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    av[t] = reinterpret_cast<const float4*>(a)[lane + 64 * t];
    bv[t] = reinterpret_cast<const float4*>(b)[lane + 64 * t];
  }
  for (int row = blockIdx.x * wpb + wave_id(); row < rows; row += gridDim.x * wpb) {
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      v[t] = xr[lane + 64 * t];
      if (res) {
        float4 r = reinterpret_cast<const float4*>(res + (size_t)row * D)[lane + 64 * t];
        if (drop.thresh != 0u) {
          const uint64_t e0 = (uint64_t)row * D + 4 * (lane + 64 * t);
          r.x *= drop_keep_scale(drop, e0); r.y *= drop_keep_scale(drop, e0 + 1);
          r.z *= drop_keep_scale(drop, e0 + 2); r.w *= drop_keep_scale(drop, e0 + 3);
        }
        v[t].x += r.x; v[t].y += r.y; v[t].z += r.z; v[t].w += r.w;
      }
      sum += (v[t].x + v[t].y) + (v[t].z + v[t].w);
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      const float dx = v[t].x - mean, dy = v[t].y - mean, dz = v[t].z - mean, dw = v[t].w - mean;
      sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    const float stdv = sqrtf(wave_sum(sq) / (float)(D - 1));
    const float r = 1.0f / (stdv + eps);
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      if (xsum_out) reinterpret_cast<float4*>(xsum_out + (size_t)row * D)[lane + 64 * t] = v[t];
      float4 o;
      o.x = av[t].x * ((v[t].x - mean) * r) + bv[t].x;
      o.y = av[t].y * ((v[t].y - mean) * r) + bv[t].y;
      o.z = av[t].z * ((v[t].z - mean) * r) + bv[t].z;
      o.w = av[t].w * ((v[t].w - mean) * r) + bv[t].w;
      // yimg: the normalised row only ever feeds GEMMs (the q/k/v or feed-forward projection and their weight gradients): written
      // as their pre-split operand image -- same bytes, same address, no split left in the consumers' loops
      reinterpret_cast<float4*>(y + (size_t)row * D)[lane + 64 * t] = yimg ? ltrx_split_image4(o) : o;
    }
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = r;
    }
  }
}

template <int NV, int WPB>
__global__ void __launch_bounds__(64 * WPB) ltrx_layernorm_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ xsum,
                                                                     const float* __restrict__ a, const float* __restrict__ mean_in,
                                                                     const float* __restrict__ rstd_in, const float* __restrict__ dres,
                                                                     int rows, float eps, float* __restrict__ dx,
                                                                     float* __restrict__ partial) {
  constexpr int D = 256 * NV;
  __shared__ __attribute__((aligned(16))) float lds[WPB * 2 * D];
  const int lane = lane_id(), w = wave_id(), wpb = WPB;
  float4 av[NV], da[NV], db[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    av[t] = reinterpret_cast<const float4*>(a)[lane + 64 * t];
    da[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = blockIdx.x * wpb + w; row < rows; row += gridDim.x * wpb) {
    const float mean = mean_in[row], r = rstd_in[row];
    float4 g[NV], xc[NV];
    float gsum = 0.f, dot = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      g[t] = reinterpret_cast<const float4*>(dy + (size_t)row * D)[lane + 64 * t];
      xc[t] = reinterpret_cast<const float4*>(xsum + (size_t)row * D)[lane + 64 * t];
      xc[t].x -= mean; xc[t].y -= mean; xc[t].z -= mean; xc[t].w -= mean;
      // da/db use the raw dy; then g becomes dy * a
      da[t].x += g[t].x * (xc[t].x * r); da[t].y += g[t].y * (xc[t].y * r);
      da[t].z += g[t].z * (xc[t].z * r); da[t].w += g[t].w * (xc[t].w * r);
      db[t].x += g[t].x; db[t].y += g[t].y; db[t].z += g[t].z; db[t].w += g[t].w;
      g[t].x *= av[t].x; g[t].y *= av[t].y; g[t].z *= av[t].z; g[t].w *= av[t].w;
      gsum += (g[t].x + g[t].y) + (g[t].z + g[t].w);
      dot += (g[t].x * xc[t].x + g[t].y * xc[t].y) + (g[t].z * xc[t].z + g[t].w * xc[t].w);
    }
    const float gm = wave_sum(gsum) / (float)D;
    dot = wave_sum(dot);
    const float stdv = 1.0f / r - eps;
    const float tc = (stdv > 0.f) ? r * r * dot / ((float)(D - 1) * stdv) : 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      float4 o;
      o.x = r * (g[t].x - gm) - tc * xc[t].x;
      o.y = r * (g[t].y - gm) - tc * xc[t].y;
      o.z = r * (g[t].z - gm) - tc * xc[t].z;
      o.w = r * (g[t].w - gm) - tc * xc[t].w;
      if (dres) {
        const float4 d = reinterpret_cast<const float4*>(dres + (size_t)row * D)[lane + 64 * t];
        o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w;
      }
      reinterpret_cast<float4*>(dx + (size_t)row * D)[lane + 64 * t] = o;
    }
  }
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    reinterpret_cast<float4*>(lds + (size_t)w * 2 * D)[lane + 64 * t] = da[t];
    reinterpret_cast<float4*>(lds + (size_t)w * 2 * D + D)[lane + 64 * t] = db[t];
  }
  __syncthreads();
  float* pa = partial + (size_t)blockIdx.x * 2 * D;
  for (int c = threadIdx.x; c < 2 * D; c += blockDim.x) {
    float sacc = 0.f;
    for (int ww = 0; ww < wpb; ++ww) sacc += lds[(size_t)ww * 2 * D + c];
    pa[c] = sacc;
  }
}

// One workgroup per 64 columns; its 16 waves split the partial rows, lanes own consecutive columns (coalesced),
// the wave partials are combined through LDS in a fixed order (deterministic).
__global__ void __launch_bounds__(1024) ltrx_layernorm_bwd_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                                         int D, float* __restrict__ da,
                                                                         float* __restrict__ db) {
  __shared__ float sa[16][64];
  __shared__ float sb[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a = 0.f, b = 0.f;
  if (c < D) {
    for (int k = w; k < nblk; k += 16) {
      a += partial[(size_t)k * 2 * D + c];
      b += partial[(size_t)k * 2 * D + D + c];
    }
  }
  sa[w][lane] = a;
  sb[w][lane] = b;
  __syncthreads();
  if (w == 0 && c < D) {
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      ta += sa[k][lane];
      tb += sb[k][lane];
    }
    da[c] = ta;
    db[c] = tb;
  }
}

static int ln_grid(int rows) {
  int g = (rows + 3) / 4;
  return g > 1024 ? 1024 : (g < 1 ? 1 : g);
}
// the backward keeps per-block column partials: fewer, fatter blocks (each wave walks many rows)
// (round 3: 1024 four-wave workgroups run the kernel 20 % faster than 320 (120 -> 96 us at 61440 x 512: 16 instead of 5 waves per CU
//  in flight) but triple the partial rows of the reduce kernel (8 -> 22 us); 256 sixteen-wave workgroups -- ln_bwd_wide below -- keep the
//  waves in flight AND cut the partial rows: 118 + 8.5 -> 94 + 6.9 us, same-box A/B gpurun_out/r3_ln_wide_ab.txt, profiles/NOTES.md)
static int ln_bwd_grid(int rows) {
  int g = (rows + 15) / 16;
  return g > 320 ? 320 : (g < 1 ? 1 : g);
}
// D <= 512 and enough rows: 16-wave workgroups, one per CU -- the same 16 waves per CU in flight as 1024 four-wave workgroups
// (which stream 20 % faster than 320 of them) but 256 partial rows for the reduce kernel instead of 1024
#ifndef LTRX_LN_BWD_G16
#define LTRX_LN_BWD_G16 256
#endif
#ifndef LTRX_LN_BWD_WIDE_ROWS
#define LTRX_LN_BWD_WIDE_ROWS (16 * 4 * LTRX_LN_BWD_G16)
#endif
static bool ln_bwd_wide(int rows, int D) { return D <= 512 && rows >= LTRX_LN_BWD_WIDE_ROWS; }
static int ln_fwd_vec_grid(int rows) {
  int g = (rows + 7) / 8;
  return g > 1024 ? 1024 : (g < 1 ? 1 : g);
}
static bool ln_vec_ok(int D, const void* p0, const void* p1, const void* p2) {
  return D % 256 == 0 && D <= 1024 && (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) == 0;
}

static int ln_fwd_launch(const float* x, const float* res, const float* a, const float* b, int rows, int D,
                         float eps, float* xsum_out, float* y_out, float* mean_out, float* rstd_out,
                         float res_drop_p, uint32_t drop_seed, const uint32_t* drop_step, int y_as_image, ltrx_stream_t stream) {
  if (!x || !a || !b || !y_out || !mean_out || !rstd_out || rows <= 0 || D < 2) return LTRX_EINVAL;
  if (res && !xsum_out) return LTRX_EINVAL;
  if (!(res_drop_p >= 0.f) || res_drop_p >= 1.f) return LTRX_EINVAL;
  const DropSpec drop = ltrx_make_drop(res ? res_drop_p : 0.f, drop_seed);
  hipStream_t s = (hipStream_t)stream;
  if (ln_vec_ok(D, x, y_out, res) && ln_vec_ok(D, a, b, xsum_out)) {
    const dim3 g(ln_fwd_vec_grid(rows));
#define LTRX_LN_FWD(NV) hipLaunchKernelGGL(ltrx_layernorm_fwd_vec_kernel<NV>, g, dim3(256), 0, s, x, res, a, b, rows, eps, xsum_out, y_out, mean_out, rstd_out, drop, drop_step, y_as_image)
    switch (D / 256) {
      case 1: LTRX_LN_FWD(1); break;
      case 2: LTRX_LN_FWD(2); break;
      case 3: LTRX_LN_FWD(3); break;
      default: LTRX_LN_FWD(4); break;
    }
#undef LTRX_LN_FWD
  } else {
    if (y_as_image) return LTRX_EUNSUPPORTED;       // (the image form exists in the row-in-registers kernel: D = 256, 512, 768, 1024)
    hipLaunchKernelGGL(ltrx_layernorm_fwd_kernel, dim3(ln_grid(rows)), dim3(256), 0, s, x, res, a, b, rows, D, eps, xsum_out,
                       y_out, mean_out, rstd_out, drop, drop_step);
  }
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

extern "C" int ltrx_layernorm_fwd(const float* x, const float* res, const float* a, const float* b, int rows, int D,
                                  float eps, float* xsum_out, float* y_out, float* mean_out, float* rstd_out,
                                  float res_drop_p, uint32_t drop_seed, const uint32_t* drop_step, ltrx_stream_t stream) {
  return ln_fwd_launch(x, res, a, b, rows, D, eps, xsum_out, y_out, mean_out, rstd_out, res_drop_p, drop_seed, drop_step, 0, stream);
}

// the same with y written as a pre-split operand IMAGE (include/ltrx.h: ltrx_split_image's layout) -- for a normalised activation
// that only feeds GEMMs (ltrx_gemm_nt_img with LTRX_GEMM_A_IS_IMAGE, ltrx_gemm_tn_group_img with b_is_image)
extern "C" int ltrx_layernorm_fwd_image(const float* x, const float* res, const float* a, const float* b, int rows, int D,
                                        float eps, float* xsum_out, void* y_image_out, float* mean_out, float* rstd_out,
                                        float res_drop_p, uint32_t drop_seed, const uint32_t* drop_step, ltrx_stream_t stream) {
  return ln_fwd_launch(x, res, a, b, rows, D, eps, xsum_out, reinterpret_cast<float*>(y_image_out), mean_out, rstd_out, res_drop_p,
                       drop_seed, drop_step, 1, stream);
}

extern "C" size_t ltrx_layernorm_bwd_workspace_bytes(int rows, int D) {
  if (rows <= 0 || D <= 0) return 0;
  return (size_t)(LTRX_LN_BWD_G16 > 320 ? LTRX_LN_BWD_G16 : 320) * 2 * D * sizeof(float);
}

// the streaming kernel of the backward: dx, and per-workgroup column partials [grid][2][D] of (da, db) in ws; returns the partial
// row count through *grid_out
static int ln_bwd_main(const float* dy, const float* xsum, const float* a, const float* mean, const float* rstd, const float* dres_in,
                       int rows, int D, float eps, float* dx_out, void* ws, int* grid_out, hipStream_t s) {
  if (!dy || !xsum || !a || !mean || !rstd || !dx_out || !ws || rows <= 0 || D < 2) return LTRX_EINVAL;
  if ((size_t)4 * 2 * D * sizeof(float) > 64 * 1024) return LTRX_EUNSUPPORTED;   // D <= 2048
  int grid = ln_bwd_grid(rows);
  if (ln_vec_ok(D, dy, xsum, dx_out) && ln_vec_ok(D, a, dres_in, ws)) {
#define LTRX_LN_BWD(NV, WPB) \
  hipLaunchKernelGGL((ltrx_layernorm_bwd_vec_kernel<NV, WPB>), dim3(grid), dim3(64 * WPB), 0, s, dy, xsum, a, mean, rstd, dres_in, rows, eps, dx_out, (float*)ws)
    if (ln_bwd_wide(rows, D)) {
      grid = LTRX_LN_BWD_G16;
      if (D / 256 == 1) LTRX_LN_BWD(1, 16); else LTRX_LN_BWD(2, 16);
    } else {
      switch (D / 256) {
        case 1: LTRX_LN_BWD(1, 4); break;
        case 2: LTRX_LN_BWD(2, 4); break;
        case 3: LTRX_LN_BWD(3, 4); break;
        default: LTRX_LN_BWD(4, 4); break;
      }
    }
#undef LTRX_LN_BWD
  } else {
    hipLaunchKernelGGL(ltrx_layernorm_bwd_kernel, dim3(grid), dim3(256), (size_t)4 * 2 * D * sizeof(float), s, dy, xsum, a,
                       mean, rstd, dres_in, rows, D, eps, dx_out, (float*)ws);
  }
  LTRX_LAUNCH_CHECK();
  *grid_out = grid;
  return LTRX_OK;
}

extern "C" int ltrx_layernorm_bwd(const float* dy, const float* xsum, const float* a, const float* mean,
                                  const float* rstd, const float* dres_in, int rows, int D, float eps, float* dx_out,
                                  float* da_out, float* db_out, void* ws, ltrx_stream_t stream) {
  if (!da_out || !db_out) return LTRX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int grid = 0;
  const int rc = ln_bwd_main(dy, xsum, a, mean, rstd, dres_in, rows, D, eps, dx_out, ws, &grid, s);
  if (rc != LTRX_OK) return rc;
  hipLaunchKernelGGL(ltrx_layernorm_bwd_reduce_kernel, dim3((D + 63) / 64), dim3(1024), 0, s, (const float*)ws, grid, D,
                     da_out, db_out);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// the same backward WITHOUT the parameter-gradient reduction: dx is final, ws holds *partial_rows_out rows of [da(D) | db(D)] partials
// for the caller to sum -- ltrx_reduce_group sums the partials of several LayerNorms and the weight-gradient slabs of
// ltrx_gemm_tn_group in one launch (da = column sums of ws[:, 0:D], db = of ws[:, D:2D]; row stride 2 D).
extern "C" int ltrx_layernorm_bwd_partial(const float* dy, const float* xsum, const float* a, const float* mean, const float* rstd,
                                          const float* dres_in, int rows, int D, float eps, float* dx_out, void* ws,
                                          int* partial_rows_out, ltrx_stream_t stream) {
  if (!partial_rows_out) return LTRX_EINVAL;
  return ln_bwd_main(dy, xsum, a, mean, rstd, dres_in, rows, D, eps, dx_out, ws, partial_rows_out, (hipStream_t)stream);
}
