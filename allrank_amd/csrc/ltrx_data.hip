// On-device batch assembly for a training set that lives in HBM in CSR form: the FixLength
// transform of allrank/data/dataset_loading.py:32-93 and the stacking of a batch (ToTensor + default collate, :19-29).
//
//   ltrx_fixlength_positions : per slate of the batch, the positions (inside the slate) that fill its L slots:
//       len <  L : 0 .. len-1, then -1                       (FixLength._pad, :81-93)
//       len >= L : L distinct positions, uniformly at random, in random order   (np.random.choice(len, L, replace=False), :70)
//                  if the sample holds no relevant item (label sum 0) and the slate's label sum is exactly 1, the last slot
//                  is replaced by argmax(labels) (:72-74); if the slate has other relevance, the draw is repeated (:75-76)
//     One workgroup per slate: a counter-based key per item (hash of seed, slate ID IN THE DATASET, attempt, position -- round 6:
//     not the row inside the batch, so a rank that assembles only its block of a global batch draws what a one-rank run draws
//     for the same slate), the rank of every key by
//     counting in LDS (ties by position) -- the L largest keys in descending order are a uniform random L-subset in uniform
//     random order -- label sums by workgroup reductions.  No host round trip, no data-dependent launch sizes.
//   ltrx_assemble_batch : xb[b][l][:] = x_items[offsets[slate_b] + pos] (zeros for pos = -1), yb = label or -1,
//       indices = pos (the original rank, what positional encodings consume) -- float4-coalesced reads and writes.
#include "ltrx_device.h"

using namespace ltrx;

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}

constexpr int FIX_MAX_LEN = 12288;        // items of one slate held in LDS (48 KB of keys)

}  // namespace

__global__ void __launch_bounds__(256) ltrx_fixlength_positions_kernel(const int64_t* __restrict__ offsets,
                                                                       const float* __restrict__ y_items,
                                                                       const int64_t* __restrict__ slates, int L, uint32_t seed_lo,
                                                                       uint32_t seed_hi, int64_t* __restrict__ positions) {
  __shared__ uint32_t keys[FIX_MAX_LEN];
  __shared__ float redf[LTRX_MAX_WAVES];
  __shared__ int redi[LTRX_MAX_WAVES];
  __shared__ int sh_arg;
  const int b = blockIdx.x;
  const int64_t s = slates[b];
  const int64_t base = offsets[s];
  const int len = (int)(offsets[s + 1] - base);
  int64_t* out = positions + (size_t)b * L;
  if (len < L) {
    for (int l = threadIdx.x; l < L; l += blockDim.x) out[l] = l < len ? l : -1;
    return;
  }
  const float* y = y_items + base;
  // slate-level label statistics: total and the first position of the maximum (np.argmax)
  float tot = 0.f, best = -INFINITY;
  int arg = 0x7fffffff;
  for (int i = threadIdx.x; i < len; i += blockDim.x) {
    const float v = y[i];
    tot += v;
    if (v > best) {
      best = v;
      arg = i;
    }
  }
  tot = block_sum(tot, redf);
  const float gbest = block_max(best, redf);
  {
    int cand = (best == gbest) ? arg : 0x7fffffff;       // smallest position among the threads that saw the maximum
    int v = cand;                                        // workgroup minimum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if (lane_id() == 0) redi[wave_id()] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      int m = redi[0];
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = min(m, redi[w]);
      sh_arg = m;
    }
    __syncthreads();
  }
  const int argmax_pos = sh_arg;
  for (int attempt = 0; attempt < 64; ++attempt) {
    const uint32_t sd = mix32(seed_lo ^ mix32(seed_hi + 0x9E3779B9u * (uint32_t)(s + 1) + 0x7F4A7C15u * (uint32_t)((uint64_t)s >> 32)) ^ (0x85EBCA6Bu * (uint32_t)attempt));
    for (int i = threadIdx.x; i < len; i += blockDim.x) keys[i] = mix32(sd ^ (0xC2B2AE35u * (uint32_t)(i + 1)));
    __syncthreads();
    float ysel = 0.f;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
      const uint32_t k = keys[i];
      int rank = 0;
      for (int j = 0; j < len; ++j) {
        const uint32_t kj = keys[j];
        rank += (kj > k) || (kj == k && j < i);
      }
      if (rank < L) {
        out[rank] = i;
        ysel += y[i];
      }
    }
    ysel = block_sum(ysel, redf);                       // (also orders the writes of `out` before the fix-up below)
    if (ysel == 0.f) {
      if (tot == 1.f) {
        if (threadIdx.x == 0) out[L - 1] = argmax_pos;   // dataset_loading.py:72-74
        return;
      }
      if (tot > 0.f) {
        __syncthreads();
        continue;                                        // :75-76 draw again
      }
    }
    return;
  }
}

extern "C" int ltrx_fixlength_positions(const int64_t* offsets, const float* y_items, const int64_t* slates, int B, int L,
                                        int max_slate_len, uint64_t seed, int64_t* positions, ltrx_stream_t stream) {
  if (!offsets || !y_items || !slates || !positions || B <= 0 || L <= 0) return LTRX_EINVAL;
  if (max_slate_len > FIX_MAX_LEN) return LTRX_EUNSUPPORTED;
  hipLaunchKernelGGL(ltrx_fixlength_positions_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, offsets, y_items, slates, L,
                     (uint32_t)seed, (uint32_t)(seed >> 32), positions);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

template <bool VEC>
__global__ void __launch_bounds__(256) ltrx_assemble_batch_kernel(const float* __restrict__ x_items, const float* __restrict__ y_items,
                                                                  const int64_t* __restrict__ offsets, const int64_t* __restrict__ slates,
                                                                  const int64_t* __restrict__ positions, int L, int F,
                                                                  size_t total, float* __restrict__ xb, float* __restrict__ yb,
                                                                  int64_t* __restrict__ idx) {
  const int per_row = VEC ? F / 4 : F;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t row = i / per_row;                   // b * L + l
  const int c = (int)(i % per_row);
  const int64_t pos = positions[row];
  const int64_t item = pos >= 0 ? offsets[slates[row / L]] + pos : 0;
  if (VEC) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pos >= 0) v = reinterpret_cast<const float4*>(x_items + (size_t)item * F)[c];
    reinterpret_cast<float4*>(xb + row * F)[c] = v;
  } else {
    xb[row * F + c] = pos >= 0 ? x_items[(size_t)item * F + c] : 0.f;
  }
  if (c == 0) {
    yb[row] = pos >= 0 ? y_items[item] : -1.0f;      // PADDED_Y_VALUE
    idx[row] = pos;                                   // PADDED_INDEX_VALUE = -1
  }
}

extern "C" int ltrx_assemble_batch(const float* x_items, const float* y_items, const int64_t* offsets, const int64_t* slates,
                                   const int64_t* positions, int B, int L, int F, float* xb, float* yb, int64_t* indices,
                                   ltrx_stream_t stream) {
  if (!x_items || !y_items || !offsets || !slates || !positions || !xb || !yb || !indices || B <= 0 || L <= 0 || F <= 0) return LTRX_EINVAL;
  const bool vec = (F % 4 == 0) && ((((uintptr_t)x_items | (uintptr_t)xb) & 15) == 0);
  const size_t total = (size_t)B * L * (vec ? F / 4 : F);
  const dim3 grid((unsigned)((total + 255) / 256));
  if (vec)
    hipLaunchKernelGGL(ltrx_assemble_batch_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x_items, y_items, offsets, slates, positions,
                       L, F, total, xb, yb, indices);
  else
    hipLaunchKernelGGL(ltrx_assemble_batch_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x_items, y_items, offsets, slates, positions,
                       L, F, total, xb, yb, indices);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// libsvm / SVMlight text -> CSR-ready arrays, on the device (the reference parses with scikit-learn's load_svmlight_file on
// the host, dataset_loading.py:130: ~100 MB/s against 3.7 GB of text for one WEB30K fold).  The file's bytes are uploaded
// once; one thread per line parses   <label> qid:<id> <index>:<value> ... [# comment]
//   pass 1 (X == NULL): label, qid, smallest and largest feature index (for sklearn's zero_based="auto" rule and n_features)
//   pass 2: the values into the dense row X[line][index - index_base]
// Numbers: up to 19 significant digits accumulated exactly in 64 bits, scaled by an exact power of ten (|exp10| <= 22) with ONE
// double-precision operation, i.e. correctly rounded like strtod for the inputs LTR files contain, then narrowed to fp32 --
// the same double -> float path as the reference (np.float64 from sklearn, torch.float32 in ToTensor).
// ------------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\r'; }

// parses a decimal floating-point literal at p (< end); returns the position after it; *ok = false if no digits were found
__device__ const unsigned char* parse_number(const unsigned char* p, const unsigned char* end, double* out, bool* ok) {
  const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19,
                          1e20, 1e21, 1e22};
  bool neg = false;
  if (p < end && (*p == '-' || *p == '+')) {
    neg = *p == '-';
    ++p;
  }
  unsigned long long mant = 0;
  int nd = 0, e10 = 0;
  bool any = false, dot = false;
  for (; p < end; ++p) {
    const unsigned char c = *p;
    if (c >= '0' && c <= '9') {
      any = true;
      if (nd < 19) {
        mant = mant * 10ull + (unsigned long long)(c - '0');
        if (mant != 0ull) ++nd;
        if (dot) --e10;
      } else if (!dot) {
        ++e10;                                  // digits beyond the 19th only move the decimal point
      }
    } else if (c == '.' && !dot) {
      dot = true;
    } else {
      break;
    }
  }
  if (any && p < end && (*p == 'e' || *p == 'E')) {
    const unsigned char* q = p + 1;
    bool eneg = false;
    if (q < end && (*q == '-' || *q == '+')) {
      eneg = *q == '-';
      ++q;
    }
    int ev = 0;
    bool ed = false;
    for (; q < end && *q >= '0' && *q <= '9'; ++q) {
      ev = ev < 10000 ? ev * 10 + (*q - '0') : ev;
      ed = true;
    }
    if (ed) {
      e10 += eneg ? -ev : ev;
      p = q;
    }
  }
  double v = (double)mant;
  if (e10 > 0) v = e10 <= 22 ? v * P10[e10] : v * pow(10.0, (double)e10);
  else if (e10 < 0) v = -e10 <= 22 ? v / P10[-e10] : v / pow(10.0, (double)(-e10));
  *out = neg ? -v : v;
  *ok = any;
  return p;
}

}  // namespace

__global__ void __launch_bounds__(256) ltrx_libsvm_parse_kernel(const unsigned char* __restrict__ text, const int64_t* __restrict__ line_start,
                                                                int64_t n_lines, int64_t n_bytes, float* __restrict__ y,
                                                                int64_t* __restrict__ qid, float* __restrict__ X, int n_features,
                                                                int index_base, int* __restrict__ minmax_index, int* __restrict__ bad) {
  const int64_t line = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= n_lines) return;
  const unsigned char* p = text + line_start[line];
  const unsigned char* limit = text + (line + 1 < n_lines ? line_start[line + 1] : n_bytes);
  const unsigned char* end = p;                     // (the caller may have dropped blank / comment lines from line_start)
  while (end < limit && *end != '\n') ++end;
  while (end > p && end[-1] == '\r') --end;
  while (p < end && is_space(*p)) ++p;
  double v;
  bool ok;
  p = parse_number(p, end, &v, &ok);
  if (!ok) {
    atomicAdd(bad, 1);
    return;
  }
  if (y) y[line] = (float)v;
  int64_t q = 0;
  int lo = 0x7fffffff, hi = -1;
  while (p < end) {
    while (p < end && is_space(*p)) ++p;
    if (p >= end || *p == '#') break;
    if (end - p > 4 && p[0] == 'q' && p[1] == 'i' && p[2] == 'd' && p[3] == ':') {
      p += 4;
      bool qneg = false;
      if (p < end && *p == '-') {
        qneg = true;
        ++p;
      }
      int64_t t = 0;
      for (; p < end && *p >= '0' && *p <= '9'; ++p) t = t * 10 + (*p - '0');
      q = qneg ? -t : t;
      continue;
    }
    int idx = 0;
    bool idig = false;
    for (; p < end && *p >= '0' && *p <= '9'; ++p) {
      idx = idx * 10 + (*p - '0');
      idig = true;
    }
    if (!idig || p >= end || *p != ':') {
      atomicAdd(bad, 1);
      break;
    }
    ++p;
    p = parse_number(p, end, &v, &ok);
    if (!ok) {
      atomicAdd(bad, 1);
      break;
    }
    lo = min(lo, idx);
    hi = max(hi, idx);
    if (X) {
      const int c = idx - index_base;
      if (c >= 0 && c < n_features) X[(size_t)line * n_features + c] = (float)v;
    }
  }
  if (qid) qid[line] = q;
  if (minmax_index && hi >= 0) {
    atomicMin(&minmax_index[0], lo);
    atomicMax(&minmax_index[1], hi);
  }
}

extern "C" int ltrx_libsvm_parse(const uint8_t* text, const int64_t* line_start, int64_t n_lines, int64_t n_bytes, float* y,
                                 int64_t* qid, float* X, int n_features, int index_base, int* minmax_index, int* bad_lines,
                                 ltrx_stream_t stream) {
  if (!text || !line_start || !bad_lines || n_lines <= 0 || n_bytes <= 0) return LTRX_EINVAL;
  if (X && n_features <= 0) return LTRX_EINVAL;
  hipLaunchKernelGGL(ltrx_libsvm_parse_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, (hipStream_t)stream, text, line_start,
                     n_lines, n_bytes, y, qid, X, n_features, index_base, minmax_index, bad_lines);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
