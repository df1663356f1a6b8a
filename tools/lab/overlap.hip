// Lab: do global loads and MFMAs of the SAME waves overlap on a CU?  One 512-thread workgroup per CU (the large-tile GEMM's
// occupancy: 2 waves per SIMD), per iteration the K-step's instruction mix: 8 global_load_dwordx4 per thread (64 KB per
// workgroup, L2-resident source) and 48 v_mfma_f32_32x32x16_bf16 per wave on 8 independent accumulators.  No LDS, no barrier.
//   mode 1: loads only      mode 2: MFMAs only      mode 3: 8 loads, then 48 MFMAs (data consumed one iteration later)
//   mode 4: one load every 6 MFMAs (same totals)
//   mode 5: mode 3 with the MFMA operands fetched from LDS (24 ds_read_b128 per wave per iteration, conflict-free swizzle)
//   mode 6: mode 5 + one LDS-only barrier per iteration
//   mode 7: mode 6 + the staging work: split of the loaded data into bf16 hi/lo and 16 ds_write_b64 per thread (double buffer)
//   mode 10: mode 7 with the fragment reads software-pipelined one (ks, i) step ahead     mode 11: + staging spread over the steps
//   mode 8: mode 7 without the global loads (registers re-used)        mode 9: mode 7 without the MFMAs
// build: hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ long long g_clk[2];
__device__ __forceinline__ int swz_off(int row, int k) {
  const int c = (k >> 3) ^ ((row >> 2) & 3);
  return row * 32 + c * 8 + (k & 7);
}
struct Smem {
  __bf16 a[2][256 * 32];
  __bf16 b[2][256 * 32];
};
__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}

// modes 5..9: the K-loop of the 256 x 256 x 32 kernel rebuilt piece by piece
template <int MODE>
__global__ void __launch_bounds__(512) kl(const float4* __restrict__ src, float* __restrict__ out, int iters, size_t span_f4) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem* s = reinterpret_cast<Smem*>(smem_raw);
  const float4* p = src + threadIdx.x;
  const int wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int srow = threadIdx.x >> 3, sc4 = (threadIdx.x & 7) * 4;
  for (int i = threadIdx.x; i < (int)(2 * sizeof(Smem) / 4); i += 512) reinterpret_cast<float*>(smem_raw)[i] = 0.f;
  __syncthreads();
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = make_float4(1.f + u, 2.f, 3.f, 4.f);
  size_t off = 0;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const Smem& t = s[it & 1];
    Smem& d = s[(it + 1) & 1];
    if (MODE >= 10) {
      // fragment software pipeline: step = (ks, i): 6 MFMAs on A[i] x B[0..1]; the A fragments of the NEXT step (and, at i == 3,
      // the B fragments of the next ks) are requested before this step's MFMAs -> every ds_read has 6 MFMAs of cover
      bf16x8 bq[2][2][2], aq[2][2];          // [buffer][term][j], [buffer][term]
      auto readB = [&](int buf, int ks) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bq[buf][tt][j] = *reinterpret_cast<const bf16x8*>(&t.b[tt][swz_off(wc * 64 + j * 32 + l31, ks * 16 + 8 * half)]);
      };
      auto readA = [&](int buf, int ks, int i) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          aq[buf][tt] = *reinterpret_cast<const bf16x8*>(&t.a[tt][swz_off(wr * 128 + i * 32 + l31, ks * 16 + 8 * half)]);
      };
      readB(0, 0);
      readA(0, 0, 0);
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int ks = st >> 2, i = st & 3, ab = st & 1, bb = ks;
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < 8) {
          if (i == 3) readB(bb ^ 1, ks + 1);
          readA(ab ^ 1, (st + 1) >> 2, (st + 1) & 3);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ab][0], bq[bb][1][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ab][1], bq[bb][0][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ab][0], bq[bb][0][j], acc[i][j], 0, 0, 0);
        if (MODE == 13 && (st & 1)) {
          const int q = st >> 1;
          if (v[q].x + v[4 + q].y == 12345.678f) acc[0][0][1] += 1.f;
          v[q] = p[(off + (size_t)q * 512) % span_f4];
          v[4 + q] = p[(off + (size_t)(4 + q) * 512) % span_f4];
        }
        if (MODE == 11 && (st & 1)) {            // staging interleaved: one (A, B) float4 pair after every second step
          const int q = st >> 1;
          bf16x4 h, l;
          const int o = swz_off(srow + 64 * q, sc4);
          split4(v[q], h, l);
          *reinterpret_cast<bf16x4*>(&d.a[0][o]) = h;
          *reinterpret_cast<bf16x4*>(&d.a[1][o]) = l;
          split4(v[4 + q], h, l);
          *reinterpret_cast<bf16x4*>(&d.b[0][o]) = h;
          *reinterpret_cast<bf16x4*>(&d.b[1][o]) = l;
          v[q] = p[(off + (size_t)q * 512) % span_f4];
          v[4 + q] = p[(off + (size_t)(4 + q) * 512) % span_f4];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE != 9) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[2][4], bfr[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bfr[tt][j] = *reinterpret_cast<const bf16x8*>(&t.b[tt][swz_off(wc * 64 + j * 32 + l31, ks * 16 + 8 * half)]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            af[tt][i] = *reinterpret_cast<const bf16x8*>(&t.a[tt][swz_off(wr * 128 + i * 32 + l31, ks * 16 + 8 * half)]);
        }
#define MMA(TA, TB)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =       \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA][i], bfr[TB][j], acc[i][j], 0, 0, 0);
        MMA(0, 1) MMA(1, 0) MMA(0, 0)
#undef MMA
      }
    }
    if (MODE == 11 || MODE == 13) {
    } else if (MODE >= 7 && MODE != 12) {                    // staging: split + LDS store of the data loaded one iteration ago
      bf16x4 h, l;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = swz_off(srow + 64 * q, sc4);
        split4(v[q], h, l);
        *reinterpret_cast<bf16x4*>(&d.a[0][o]) = h;
        *reinterpret_cast<bf16x4*>(&d.a[1][o]) = l;
        split4(v[4 + q], h, l);
        *reinterpret_cast<bf16x4*>(&d.b[0][o]) = h;
        *reinterpret_cast<bf16x4*>(&d.b[1][o]) = l;
      }
    } else {
      float4 z = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) { z.x += v[u].x; z.y += v[u].y; z.z += v[u].z; z.w += v[u].w; }
      if (z.x == 12345.678f) acc[0][0][0] += z.y + z.z + z.w;
    }
    if (MODE != 8 && MODE != 11 && MODE != 13) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(off + (size_t)u * 512) % span_f4];
    }
    if (MODE >= 6 && MODE != 12 && MODE != 13) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    off += 8 * 512;
  }
  if (blockIdx.x == 7 && threadIdx.x == 0) {
    g_clk[0] = clock64() - c0;
    g_clk[1] = wall_clock64() - w0;
  }
  float tsum = v[0].x;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tsum += acc[i][j][r];
  if (tsum == 12345.678f) out[0] = tsum;
}

template <int MODE>
__global__ void __launch_bounds__(512) k(const float4* __restrict__ src, float* __restrict__ out, int iters, size_t span_f4) {
  const float4* p = src + threadIdx.x;
  f32x16 acc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 fa, fb;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    fa[e] = (__bf16)(float)(threadIdx.x + e);
    fb[e] = (__bf16)(float)(threadIdx.x * 3 + e);
  }
  float4 v[8], s = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = make_float4(0, 0, 0, 0);
  size_t off = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE != 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {            // consume the previous iteration's data (forces the wait here, after the MFMAs)
        s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
      }
    }
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(off + (size_t)u * 512) % span_f4];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 2 || MODE == 3) {
#pragma unroll
      for (int m = 0; m < 48; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 7], 0, 0, 0);
    }
    if (MODE == 4) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = p[(off + (size_t)u * 512) % span_f4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 6; ++m) acc[(6 * u + m) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[(6 * u + m) & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    off += 8 * 512;
  }
  float t = s.x + s.y + s.z + s.w;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[a][r];
  if (t == 12345.678f) out[0] = t;
}

template <int MODE>
static void run(const float4* src, float* out, const char* name) {
  const int iters = 2000;
  const size_t span = (size_t)(2 << 20) / 16;    // every workgroup re-reads the same 2 MB: L2 hits
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, src, out, iters, span);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, src, out, iters, span);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.3f us per iteration (K-step equivalent)\n", name, ms * 1e3 / 3 / iters);
}

template <int MODE>
static void runl(const float4* src, float* out, const char* name) {
  const int iters = 2000;
  const size_t span = (size_t)(2 << 20) / 16;
  hipFuncSetAttribute((const void*)kl<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * sizeof(Smem)));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kl<MODE>, dim3(256), dim3(512), 2 * sizeof(Smem), 0, src, out, iters, span);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kl<MODE>, dim3(256), dim3(512), 2 * sizeof(Smem), 0, src, out, iters, span);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[2];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h));
  printf("%-44s %8.3f us per iteration   %6.0f shader cycles per iteration, %5.0f MHz effective (100 MHz wall counter)\n", name,
         ms * 1e3 / 3 / iters, (double)h[0] / iters, (double)h[0] / ((double)h[1] / 100.0));
}

// mode 20/21: the same K-step on FOUR waves (one per SIMD, up to 512 registers each): wave tile 128 x 128 (256 accumulator
// registers), 96 MFMAs, 16 global loads, 32 ds_write_b64 and 32 ds_read_b128 per wave per iteration; everything but the MFMAs
// has to hide in the issue gaps of the wave's own MFMA stream.  20: source order = (ks, i) steps of 12 MFMAs with the next
// step's fragments requested first and 1/8 of the staging after each step.  21: the staging work only after the MFMAs (bulk).
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
kl4(const float4* __restrict__ src, float* __restrict__ out, int iters, size_t span_f4) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem* s = reinterpret_cast<Smem*>(smem_raw);
  const float4* p = src + threadIdx.x;
  const int wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int srow = threadIdx.x >> 3, sc4 = (threadIdx.x & 7) * 4;       // 32 rows x 8 float4 per pass, 8 passes per operand
  for (int i = threadIdx.x; i < (int)(2 * sizeof(Smem) / 4); i += 256) reinterpret_cast<float*>(smem_raw)[i] = 0.f;
  __syncthreads();
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = make_float4(1.f + u, 2.f, 3.f, 4.f);
  size_t off = 0;
  for (int it = 0; it < iters; ++it) {
    const Smem& t = s[it & 1];
    Smem& d = s[(it + 1) & 1];
    bf16x8 bq[2][2][4], aq[2][2];          // [buffer][term][j], [buffer][term]
    auto readB = [&](int buf, int ks) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bq[buf][tt][j] = *reinterpret_cast<const bf16x8*>(&t.b[tt][swz_off(wc * 128 + j * 32 + l31, ks * 16 + 8 * half)]);
    };
    auto readA = [&](int buf, int ks, int i) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        aq[buf][tt] = *reinterpret_cast<const bf16x8*>(&t.a[tt][swz_off(wr * 128 + i * 32 + l31, ks * 16 + 8 * half)]);
    };
    auto stage = [&](int q) {              // one A and one B float4 of this thread: split, store, reload
      bf16x4 h, l;
      const int o = swz_off(srow + 32 * q, sc4);
      split4(v[q], h, l);
      *reinterpret_cast<bf16x4*>(&d.a[0][o]) = h;
      *reinterpret_cast<bf16x4*>(&d.a[1][o]) = l;
      split4(v[8 + q], h, l);
      *reinterpret_cast<bf16x4*>(&d.b[0][o]) = h;
      *reinterpret_cast<bf16x4*>(&d.b[1][o]) = l;
      v[q] = p[(off + (size_t)q * 256) % span_f4];
      v[8 + q] = p[(off + (size_t)(8 + q) * 256) % span_f4];
    };
    readB(0, 0);
    readA(0, 0, 0);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int ks = st >> 2, i = st & 3, ab = st & 1, bb = ks;
      __builtin_amdgcn_sched_barrier(0);
      if (st + 1 < 8) {
        if (i == 3) readB(bb ^ 1, ks + 1);
        readA(ab ^ 1, (st + 1) >> 2, (st + 1) & 3);
      }
      if (MODE == 22) stage(st);
      if (MODE != 22) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ab][0], bq[bb][1][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ab][1], bq[bb][0][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ab][0], bq[bb][0][j], acc[i][j], 0, 0, 0);
      if (MODE == 20) stage(st);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 21) {
#pragma unroll
      for (int q = 0; q < 8; ++q) stage(q);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    off += 16 * 256;
  }
  float tsum = v[0].x;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tsum += acc[i][j][r];
  if (tsum == 12345.678f) out[0] = tsum;
}

template <int MODE>
static void runl4(const float4* src, float* out, const char* name) {
  const int iters = 2000;
  const size_t span = (size_t)(2 << 20) / 16;
  hipFuncSetAttribute((const void*)kl4<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * sizeof(Smem)));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kl4<MODE>, dim3(256), dim3(256), 2 * sizeof(Smem), 0, src, out, iters, span);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kl4<MODE>, dim3(256), dim3(256), 2 * sizeof(Smem), 0, src, out, iters, span);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.3f us per iteration (K-step equivalent)\n", name, ms * 1e3 / 3 / iters);
}

// modes 30-34: do MFMAs and VALU work overlap on a SIMD?  NW waves per SIMD (workgroup of 256 * NW threads), per iteration
// 12 v_mfma_f32_32x32x16_bf16 (384 pipe cycles) and NV independent v_fma_f32.
//   30: MFMAs only   31: VALU only   32: 12 MFMAs then NV VALU (phases)   33: one MFMA, NV/12 VALU, ... (interleaved in program order)
template <int MODE, int NV>
__global__ void __launch_bounds__(512) kv(float* __restrict__ out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 fa, fb;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    fa[e] = (__bf16)(float)(threadIdx.x + e);
    fb[e] = (__bf16)(float)(threadIdx.x * 3 + e);
  }
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = 1.0f + i + threadIdx.x;
  const float c1 = 1.0001f, c2 = 0.5f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 30 || MODE == 32) {
#pragma unroll
      for (int m = 0; m < 12; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 31 || MODE == 32) {
#pragma unroll
      for (int v = 0; v < NV; ++v) x[v & 15] = __builtin_fmaf(x[v & 15], c1, c2);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 33) {
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < NV / 12; ++v) x[(m * (NV / 12) + v) & 15] = __builtin_fmaf(x[(m * (NV / 12) + v) & 15], c1, c2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += x[i];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[a][r];
  if (t == 12345.678f) out[0] = t;
}
template <int MODE, int NV>
static void runv(float* out, int threads, const char* name) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((kv<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kv<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s waves/SIMD %d  NV %3d: %7.1f ns per iteration\n", name, threads / 256, NV, ms * 1e6 / 3 / iters);
}

int main() {
  float4* src;
  float* out;
  hipMalloc(&src, (size_t)64 << 20);
  hipMalloc(&out, 64);
  hipMemset(src, 0, (size_t)64 << 20);
  if (getenv("VALU_ONLY")) {
    for (int th = 256; th <= 512; th += 256) {
      runv<30, 144>(out, th, "30: 12 MFMAs");
      runv<31, 144>(out, th, "31: 144 v_fma");
      runv<32, 144>(out, th, "32: 12 MFMAs, then 144 v_fma");
      runv<33, 144>(out, th, "33: (1 MFMA, 12 v_fma) x 12");
      runv<31, 48>(out, th, "31: 48 v_fma");
      runv<32, 48>(out, th, "32: 12 MFMAs, then 48 v_fma");
      runv<33, 48>(out, th, "33: (1 MFMA, 4 v_fma) x 12");
    }
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep) {
    run<1>(src, out, "loads only (8 x dwordx4 per thread)");
    run<2>(src, out, "MFMAs only (48 per wave)");
    run<3>(src, out, "8 loads then 48 MFMAs");
    run<4>(src, out, "1 load per 6 MFMAs");
    runl<5>(src, out, "5: loads + MFMAs fed by 24 ds_read_b128");
    runl<6>(src, out, "6: + barrier per iteration");
    runl<7>(src, out, "7: + split and 16 ds_write_b64 (full loop)");
    runl<8>(src, out, "8: full loop without global loads");
    runl<9>(src, out, "9: full loop without MFMAs/ds_reads");
    runl<10>(src, out, "10: full loop, fragment reads pipelined");
    runl<11>(src, out, "11: 10 + staging interleaved per 2 steps");
    runl<12>(src, out, "12: mode 5 with pipelined fragment reads");
    runl<13>(src, out, "13: 12 with the loads spread (2 per 2 steps)");
    runl4<20>(src, out, "20: 4 waves x 512 regs, staging after each step");
    runl4<21>(src, out, "21: 4 waves x 512 regs, staging in bulk");
    runl4<22>(src, out, "22: 4 waves, staging before each step's MFMAs (free order)");
  }
  return 0;
}
