// Device-side helpers shared by the libltrx kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <atomic>

#include "../../include/ltrx.h"

#define LTRX_WAVE 64
#define LTRX_MAX_WAVES 16  // 1024-thread workgroup

#define LTRX_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return LTRX_EHIP - (int)e__;       \
  } while (0)

// One-time per-DEVICE setup (hipFuncSetAttribute applies to the current device's copy of a kernel): thread-safe, and the only
// process state the library keeps -- idempotent facts about loaded code, never a mode that changes results (ltrx.h: re-entrant).
template <typename F>
static inline int ltrx_once_per_device(std::atomic<uint64_t>& done, F&& setup) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return LTRX_EHIP;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return LTRX_OK;
  const int rc = setup();            // two threads racing here both set the same attribute: harmless
  if (rc == LTRX_OK) done.fetch_or(bit, std::memory_order_release);
  return rc;
}

// cut-off ranks of a metric call, passed to the kernel by value (ltrx_ndcg_at, ltrx_mrr_at)
#define LTRX_MAX_ATS 16

struct LtrxAts {
  int n;
  int at[LTRX_MAX_ATS];
};

namespace ltrx {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- wave (64-lane) reductions on the DPP cross-lane network: four in-row steps (quad_perm x2, row_half_mirror,
// row_mirror: every lane of a 16-lane row holds the row total), two row broadcasts (row_bcast15 into rows 1 and 3,
// row_bcast31 into rows 2 and 3: lane 63 holds the wave total) and one v_readlane -- 7 VALU-rate instructions instead of
// six dependent ds_bpermute round trips through the LDS crossbar.  Every lane gets the result; the order is fixed.
#define LTRX_DPP_I(old, src, ctrl, rowmask, bound) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rowmask), 0xF, (bound))
#define LTRX_DPP_F(old, src, ctrl, rowmask, bound) \
  __builtin_bit_cast(float, LTRX_DPP_I(__builtin_bit_cast(int, (old)), __builtin_bit_cast(int, (src)), (ctrl), (rowmask), (bound)))
__device__ __forceinline__ float wave_sum(float v) {
  v += LTRX_DPP_F(0.f, v, 0xB1, 0xF, true);     // quad_perm [1,0,3,2]
  v += LTRX_DPP_F(0.f, v, 0x4E, 0xF, true);     // quad_perm [2,3,0,1]
  v += LTRX_DPP_F(0.f, v, 0x141, 0xF, true);    // row_half_mirror
  v += LTRX_DPP_F(0.f, v, 0x140, 0xF, true);    // row_mirror
  v += LTRX_DPP_F(0.f, v, 0x142, 0xA, false);   // row_bcast15 -> rows 1, 3
  v += LTRX_DPP_F(0.f, v, 0x143, 0xC, false);   // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, LTRX_DPP_F(v, v, 0xB1, 0xF, false));
  v = fmaxf(v, LTRX_DPP_F(v, v, 0x4E, 0xF, false));
  v = fmaxf(v, LTRX_DPP_F(v, v, 0x141, 0xF, false));
  v = fmaxf(v, LTRX_DPP_F(v, v, 0x140, 0xF, false));
  v = fmaxf(v, LTRX_DPP_F(v, v, 0x142, 0xA, false));
  v = fmaxf(v, LTRX_DPP_F(v, v, 0x143, 0xC, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_sum_i(int v) {
  v += LTRX_DPP_I(0, v, 0xB1, 0xF, true);
  v += LTRX_DPP_I(0, v, 0x4E, 0xF, true);
  v += LTRX_DPP_I(0, v, 0x141, 0xF, true);
  v += LTRX_DPP_I(0, v, 0x140, 0xF, true);
  v += LTRX_DPP_I(0, v, 0x142, 0xA, false);
  v += LTRX_DPP_I(0, v, 0x143, 0xC, false);
  return __builtin_amdgcn_readlane(v, 63);
}

// ---- workgroup reductions: wave partials through LDS, summed in a fixed order (deterministic). ----
// `red` must hold LTRX_MAX_WAVES floats.  All threads of the block must call; all get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();  // protect `red` from a previous use
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  const int nw = blockDim.x >> 6;
  float t = 0.f;
  for (int w = 0; w < nw; ++w) t += red[w];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  const int nw = blockDim.x >> 6;
  float t = red[0];
  for (int w = 1; w < nw; ++w) t = fmaxf(t, red[w]);
  return t;
}
__device__ __forceinline__ int block_sum_i(int v, int* red) {
  v = wave_sum_i(v);
  __syncthreads();
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  const int nw = blockDim.x >> 6;
  int t = 0;
  for (int w = 0; w < nw; ++w) t += red[w];
  return t;
}

// ---- in-place inclusive prefix sum of an LDS array a[0..n) by the whole workgroup. ----
// Each thread scans a contiguous chunk serially, the per-thread totals are scanned with wave shuffles and
// a fixed-order combine across waves.  `red` holds LTRX_MAX_WAVES floats.  Contains the needed barriers;
// a[] must be fully written (and barrier'd) by the caller before the call; it is valid for all threads after.
__device__ __forceinline__ void block_inclusive_scan(float* a, int n, float* red) {
  const int T = blockDim.x;
  const int chunk = (n + T - 1) / T;
  const int lo = threadIdx.x * chunk;
  const int hi = min(lo + chunk, n);
  float tot = 0.f;
  for (int i = lo; i < hi; ++i) {
    tot += a[i];
    a[i] = tot;
  }
  // inclusive scan of `tot` across the wave
  float inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float up = __shfl_up(inc, o, 64);
    if (lane_id() >= o) inc += up;
  }
  __syncthreads();
  if (lane_id() == 63) red[wave_id()] = inc;
  __syncthreads();
  float base = inc - tot;  // exclusive prefix inside the wave
  for (int w = 0; w < wave_id(); ++w) base += red[w];
  for (int i = lo; i < hi; ++i) a[i] += base;
  __syncthreads();
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// Counter-based dropout: the keep decision of element `idx` of a tensor is a pure function of (seed, idx), so forward and
// backward kernels regenerate the same mask without storing it (murmur3 finaliser over the folded 64-bit index).
// `seed` = per-site constant XOR a per-step word read from device memory (so a captured hipGraph draws a fresh mask at
// every replay).  Returns 1/(1-p) for kept elements, 0 for dropped ones.
struct DropSpec {
  uint32_t seed;
  uint32_t thresh;     // drop iff (hash >> 8) < thresh,  thresh = p * 2^24  (0 = dropout off)
  float inv_keep;
};
__device__ __forceinline__ float drop_keep_scale(const DropSpec& d, uint64_t idx) {
  uint32_t x = (uint32_t)idx ^ ((uint32_t)(idx >> 32) * 0x9E3779B9u) ^ d.seed;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return ((x >> 8) >= d.thresh) ? d.inv_keep : 0.f;
}

}  // namespace ltrx
inline ltrx::DropSpec ltrx_make_drop(float p, uint32_t seed) {
  ltrx::DropSpec d;
  d.seed = seed;
  d.thresh = (p > 0.f) ? (uint32_t)(p * 16777216.0f) : 0u;
  d.inv_keep = (p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}
namespace ltrx {
}  // namespace ltrx

// 4 floats -> the 16 bytes {hi0..hi3, lo0..lo3} (bf16) of a pre-split operand image: hi = bf16(x), lo = bf16(x - hi), the same
// expressions the GEMM kernels apply while staging (ltrx_gemm.hip split4), so an image written anywhere is bit-identical to the
// on-the-fly split
typedef __bf16 ltrx_bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ltrx_split_image4(const float4 v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  ltrx_bf16x4_t hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
  float4 o;
  *reinterpret_cast<ltrx_bf16x4_t*>(&o.x) = hi;
  *reinterpret_cast<ltrx_bf16x4_t*>(&o.z) = lo;
  return o;
}

// Per-slate work arrays of the listwise loss kernels (`narr` floats per item): in LDS (extern __shared__) while they fit the CU's
// 160 KB, otherwise in a global workspace that the SAME kernel body addresses through a generic pointer (template flag GWS): the
// arrays of a slate stay in its CU's L1 / the XCD's L2, and __syncthreads() orders global accesses inside a workgroup exactly as it
// orders LDS.  Slates up to LTRX_MAX_SLATE_LEN take the LDS form (the tuned path); up to LTRX_MAX_LONG_SLATE_LEN the global form --
// the reference pads a validation set to its longest slate with no bound (allrank/data/dataset_loading.py:185-194).
#define LTRX_LDS_ARRAY_BUDGET_BYTES (160 * 1024 - 1024)
static inline bool ltrx_arrays_in_lds(int narr, int extra_floats, int L) {
  return ((size_t)narr * (size_t)L + (size_t)extra_floats) * sizeof(float) <= (size_t)LTRX_LDS_ARRAY_BUDGET_BYTES;
}
// bytes of the global work arrays (0 when they fit in LDS), rounded so that the per-slate block keeps 16-byte alignment
static inline size_t ltrx_array_ws_floats(int narr, int extra_floats, int B, int L) {
  if (ltrx_arrays_in_lds(narr, extra_floats, L)) return 0;
  const size_t per = (((size_t)narr * (size_t)L + (size_t)extra_floats) + 3) & ~(size_t)3;
  return per * (size_t)B;
}
static inline size_t ltrx_array_ws_stride(int narr, int extra_floats, int L) {
  return (((size_t)narr * (size_t)L + (size_t)extra_floats) + 3) & ~(size_t)3;
}

// Final cross-slate reduction: out[0] = scale * sum_b per[b]  (fixed order -> deterministic).  One block.
// (host launcher lives in ltrx_common.hip; kernels are never launched across translation units)
int ltrx_launch_finalize_sum(const float* per, int B, float scale, float* out, hipStream_t s);
// x[0] = (denom[0] != 0) ? x[0] / denom[0] : 0   (one thread; used for batch-global normalisers kept on the device)
int ltrx_launch_div_by_device_scalar(float* x, const float* denom, hipStream_t s);
