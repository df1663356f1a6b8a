// Fused masked self-attention on the bf16 matrix cores with fp32-class accuracy, for slates that FIT IN LDS
// (allrank/models/transformer.py:137-156 attention, :178-203 MultiHeadedAttention.forward; slate length <= 256, 32 < d_k <= 64).
//
// Why another attention path: the exact-fp32 kernels of ltrx_mha.hip run on v_mfma_f32_32x32x2_f32, 1/16 of the bf16 MFMA
// rate -- 29 % of the training step at config (3).  Here every contraction is three v_mfma_f32_32x32x16_bf16 products
// X Y ~= Xhi Yhi + Xhi Ylo + Xlo Yhi (x = hi + lo, both bf16; fp32 accumulate; error <= 3 * 2^-18 per product, the same
// arithmetic as the dense projections of ltrx_gemm.hip): 5.3x fewer matrix-pipe cycles.  What made the earlier split-bf16
// attempt (round 1, ltrx_mha_bf16.hip: no faster than fp32) slow was not the MFMAs but everything around them: every
// 128-query workgroup re-staged and re-split every 32-key tile of K and V behind two barriers, with a strided transposition
// pass for the operand that is contracted over keys.  This version removes all of that:
//   * ONE workgroup (8 waves) per (slate, head).  The two streamed operands of a kernel (K and V; Q and dO in the dK/dV
//     kernel) are split ONCE into bf16 hi/lo images of the whole slate -- 2 x 64 KB of the CU's 160 KB LDS -- then every wave
//     runs its whole loop out of LDS: one barrier per kernel instead of two per tile, no re-staging, no re-splitting.
//   * no transposed copies: an operand contracted over its ROWS (V in P V, K in dS K, dO and Q in the dK/dV kernel) is read
//     from the same row-major image with ds_read_b64_tr_b16 (each 16-lane group fetches a [4 rows][16 cols] block transposed:
//     lane = column, 4 consecutive rows), whose row groups {16u + 4 half + 0..3, 16u + 8 + 4 half + 0..3} are exactly the
//     rows a lane's P / dS registers 8u..8u+7 hold in the MFMA D layout -- P never moves between lanes.
//   * image layout: [row][64] bf16 per plane, 16-byte chunk c of row r at position c ^ s(r), s(r) = bit2(r) | bit3(r) << 1 |
//     bit1(r) << 2: conflict-free for the 16-lane groups of the row-wise ds_read_b128 AND for the four rows x two column
//     blocks of a transposed read (row bit 1 moves the 32-byte region, row bit 0 the 128-byte half of the bank row).
//   * the backward computes S, P, dP and delta ONCE (round 3; five tile products instead of seven): the dK/dV kernel, which needs
//     dS with the queries inside a lane, also writes it to an HBM workspace, and dQ = dS K -- which needs the keys inside a lane --
//     is a second, memory-bound kernel that reads it back: the transposition happens in memory, deterministically (no atomics,
//     no cross-wave reduction; everything in one kernel does not fit the LDS: Q, dO, K, V images alone are 256 KB).
// (Round 6: the tile loops were put on an instruction diet -- see the block above the forward kernel.)
// Softmax in the log2 domain, natural-log LSE saved, dropout on P regenerated from (seed, query row, key) -- identical
// conventions to ltrx_mha.hip, so forward / backward kernels of the two paths are interchangeable.  Variable-length
// (cu_seqlens) batches: slate b is rows cu[b] .. cu[b+1]-1; waves beyond the slate's length exit after the staging barrier.
#include "ltrx_device.h"

using namespace ltrx;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

namespace {

constexpr int RMAX = 256;                 // rows (items of a slate) held in LDS
constexpr int DK = 64;                    // padded head dimension
constexpr int PLANE = RMAX * DK * 2;      // bytes of one bf16 plane
constexpr size_t RES_STATS = 4 * (size_t)PLANE + 3 * RMAX * sizeof(float);   // LDS bytes of the four planes + per-row statistics (the forward: mask bias [RMAX] + 8 tile flags)
constexpr int XROW = 64;                  // the dS workspace's row stride is a multiple of this (64 floats = 256 bytes)
constexpr size_t dq_smem(int nw) { return 2 * (size_t)PLANE + (size_t)nw * 32 * 32 * sizeof(float); }     // dQ kernel: 80 KB with 4 waves (two workgroups per CU)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

typedef DropSpec DropCfg;
__device__ __forceinline__ uint32_t drop_row_seed(const DropCfg& d, uint32_t bh, int L, int qrow) {   // == ltrx_mha.hip
  uint32_t x = d.seed ^ ((bh * (uint32_t)L + (uint32_t)qrow) * 0x9E3779B9u);
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float drop_scale_rk(const DropCfg& d, uint32_t row_seed, int key) {
  uint32_t x = (row_seed ^ (uint32_t)key) * 0x9E3779B1u;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  return ((x >> 8) >= d.thresh) ? d.inv_keep : 0.f;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// byte offset of 16-byte chunk `chunk` (8 columns) of row `row` inside a plane
__device__ __forceinline__ int img_off(int row, int chunk) {
  const int s = ((row >> 2) & 1) | (((row >> 3) & 1) << 1) | (((row >> 1) & 1) << 2);
  return row * (DK * 2) + ((chunk ^ s) << 4);
}

// Staging is pipelined per 32-row tile: the 512 threads fetch tile t+1 of the kernel's TWO streamed tensors into registers
// (threads 0-255: tensor A, 256-511: tensor B; one (row, 8-column chunk) = 2 float4 each) while the waves compute on tile t;
// the split into bf16 hi / lo and the LDS store happen right before the barrier that opens tile t+1.  Tiles live at their
// own rows of the whole-slate image, so ONE barrier per tile orders everything (no WAR: a tile is written once).
struct TileRegs {
  float4 a, b;
};
__device__ __forceinline__ void tile_gload(TileRegs& r, const float* __restrict__ base, int tile, int nrows, int dk, size_t rs) {
  const int idx = threadIdx.x & 255;
  const int row = tile * 32 + (idx >> 3), c = (idx & 7) * 8;
  r.a = make_float4(0.f, 0.f, 0.f, 0.f);
  r.b = r.a;
  if (row < nrows && c < dk) r.a = *reinterpret_cast<const float4*>(base + (size_t)row * rs + c);
  if (row < nrows && c + 4 < dk) r.b = *reinterpret_cast<const float4*>(base + (size_t)row * rs + c + 4);
}
// PL ("plain"): the one-product bf16 throughput mode (mode 2 of ltrx_mha_fwd / ltrx_mha_bwd): no lo planes, no lo products -- NOT parity arithmetic
template <bool PL>
__device__ __forceinline__ void tile_sstore(unsigned char* img, int tile, const TileRegs& r) {
  const int idx = threadIdx.x & 255;
  const int row = (tile & 7) * 32 + (idx >> 3), chunk = idx & 7;      // ring slot of the tile
  const float x[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
  bf16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (__bf16)x[e];
    l[e] = (__bf16)(x[e] - (float)h[e]);
  }
  const int o = img_off(row, chunk);
  *reinterpret_cast<bf16x8*>(img + o) = h;
  if (!PL) *reinterpret_cast<bf16x8*>(img + PLANE + o) = l;
}

// the wave's fixed operand FIXED[row0 + l31][16 ks + 8 half + (0..7)], pre-split (registers).  Loaded 16 lanes per row (4 rows x
// 256 contiguous bytes per instruction) and turned into one-row-per-lane fragments through a wave-private 8-KB LDS scratch
// (16-byte chunk c of row r at position c ^ (r & 15)).  Read straight from memory in the fragment layout -- every lane its own row,
// 2 x 4 sixteen-byte pieces -- each instruction costs the texture-address unit one line lookup per LANE, all eight waves of a
// workgroup at once, before the first tile can start (round 3: forward 193 -> 185 us, dK/dV prologue 17.9 k -> 12.9 k cycles).
struct FixedRegs {
  f32x4 x[8];
};
__device__ __forceinline__ void fixed_gload(FixedRegs& f, const float* __restrict__ base, int row0, int nrows, int dk, size_t rs) {
  const int lane = threadIdx.x & 63, c16 = lane & 15;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = (lane >> 4) + 4 * j;
    const bool ok = (row0 + r < nrows) && (4 * c16 < dk);
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)(ok ? row0 + r : 0) * rs + (ok ? 4 * c16 : 0));    // (row 0 exists)
    f.x[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
__device__ __forceinline__ void fixed_finish(bf16x8 (&fh)[4], bf16x8 (&fl)[4], const FixedRegs& f, f32x4* scratch, float pre = 1.0f) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5, c16 = lane & 15;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = (lane >> 4) + 4 * j;
    scratch[r * 16 + (c16 ^ (r & 15))] = f.x[j];
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const f32x4 a = scratch[l31 * 16 + ((4 * ks + 2 * half) ^ (l31 & 15))];
    const f32x4 b = scratch[l31 * 16 + ((4 * ks + 2 * half + 1) ^ (l31 & 15))];
    const float v[8] = {a.x * pre, a.y * pre, a.z * pre, a.w * pre, b.x * pre, b.y * pre, b.z * pre, b.w * pre};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      fh[ks][e] = (__bf16)v[e];
      fl[ks][e] = (__bf16)(v[e] - (float)fh[ks][e]);
    }
  }
}

#define LTRX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
// barrier that orders LDS only: __syncthreads() also drains vmcnt(0), i.e. it would wait at every tile for the global loads of
// the NEXT tile that were issued just before it (and for the touch_line requests)
__device__ __forceinline__ void lds_only_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#ifdef LTRX_MHA_STAMP        // lab builds only (tools/lab/lib_variant.sh): cycle stamps of one workgroup of the forward kernel
__device__ unsigned long long g_mha_stamps[8][40][8];       // (or of the dK/dV kernel with -DLTRX_MHA_STAMP_DKDV)
#define STAMP_(kt, ph)                                                                 \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if (blockIdx.x == LTRX_MHA_STAMP && (threadIdx.x & 63) == 0) g_mha_stamps[threadIdx.x >> 6][kt][ph] = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#ifdef LTRX_MHA_STAMP_DKDV
#define STAMP(kt, ph)
#define DSTAMP(kt, ph) STAMP_(kt, ph)
#else
#define STAMP(kt, ph) STAMP_(kt, ph)
#define DSTAMP(kt, ph)
#endif
#else
#define STAMP(kt, ph)
#define DSTAMP(kt, ph)
#endif
#ifndef LTRX_MHA_TOUCH
#define LTRX_MHA_TOUCH 1
#endif
#ifndef LTRX_MHA_PRIO_HALF        // static priority for the second-dispatched half of an eight-wave workgroup (the arbitration loser of every segment)
#define LTRX_MHA_PRIO_HALF 0
#endif

// Tile products.  Row-wise: acc[r] = sum_c IMG[tile_row0 + rowmap(r, half)][c] * FIXED[l31][c] (rows_mma below).  Column-wise:
// out[ct][r'] += sum_row IMG[tile_row0 + row][32 ct + l31] * p[row]  (p in D layout: register r <-> tile row rowmap(r, half); cols_mma
// below): the A fragment (column 32 ct + l31, rows {16u + 4 half + 0..3, 16u + 8 + 4 half + 0..3}) comes from two transposed reads.
struct ColFrags {
  bf16x8 h[2][2], l[2][2];          // [u][ct]
};
// lane owns output row row0 + l31; register 4g + e of out[ct] is column 32 ct + 8 g + 4 half + e.  Written through a wave-private
// 4-KB LDS scratch, one [32 rows][32 columns] half at a time, so that a store instruction covers 8 rows x 128 contiguous bytes
// instead of 64 lanes x 16 bytes in 32 different rows (the texture-address unit pays per line: 64 instead of 512 line writes per
// wave and tensor; round 3).  Scratch layout = the dS exchange of the dQ kernel: chunk c of row r at c ^ ((r >> 1) & 7).
__device__ __forceinline__ void store_rows(float* __restrict__ base, int row0, int nrows, int dk, size_t rs, const f32x16 (&out)[2],
                                           float scale, f32x4* scratch) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      scratch[l31 * 8 + ((2 * g + half) ^ ((l31 >> 1) & 7))] =
          f32x4{out[ct][4 * g + 0] * scale, out[ct][4 * g + 1] * scale, out[ct][4 * g + 2] * scale, out[ct][4 * g + 3] * scale};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (lane >> 3) + 8 * j, c = 32 * ct + 4 * (lane & 7);
      const f32x4 v = scratch[r * 8 + ((lane & 7) ^ ((r >> 1) & 7))];
      if (row0 + r < nrows && c < dk) *reinterpret_cast<f32x4*>(base + (size_t)(row0 + r) * rs + c) = v;
    }
  }
}
// a wave-private 4-KB scratch inside the image planes once the wave has left its tile loop: only the LAST tile's ring slot can
// still be read by a slower wave (every earlier tile lies behind a barrier this wave has passed), so the half of planes 0 / 1
// (16 KB each = 8 waves x 4 KB) that does not hold that slot is free
__device__ __forceinline__ f32x4* end_scratch(unsigned char* smem, int last_tile, int wave) {
  const size_t half_off = (((last_tile & 7) >= 4) ? 0 : PLANE / 2);
  return reinterpret_cast<f32x4*>(smem + (size_t)(wave >> 2) * PLANE + half_off + (size_t)(wave & 3) * 4096);
}

__device__ __forceinline__ void zero2(f32x16 (&o)[2]) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
}

struct Slate {
  int b, head, bh, Lmax, len;
  size_t row0;
};
__device__ __forceinline__ Slate which_slate(int L, int h, const int* __restrict__ cu, const int* __restrict__ order) {
  Slate s;
  s.head = blockIdx.x % h;
  s.b = order ? order[blockIdx.x / h] : (int)(blockIdx.x / h);
  s.bh = s.b * h + s.head;
  s.Lmax = L;
  s.row0 = cu ? (size_t)cu[s.b] : (size_t)s.b * L;
  s.len = cu ? cu[s.b + 1] - cu[s.b] : L;
  return s;
}

// The workgroup that will take this CU's place is (about) blockIdx.x + 256 -- one workgroup per CU, 256 CUs, workgroup ids dealt
// to the 8 XCDs round-robin, so it runs on THIS XCD.  Every workgroup's prologue is a burst of 77-300 KB of compulsory loads
// (its fixed operands, the first streamed tile) with all CUs in the same phase: ~10 k of a forward workgroup's 57 k cycles
// (cycle stamps, tools/lab/mha_stamps.py).  The forward kernel therefore requests one dword per 128-byte line of the successor's
// Q rows and first K / V tile during its LAST tile -- any earlier and the next streamed tile's vmcnt wait, which is in-order, waits
// for these requests too (measured +14 %); as LDS-DMA into a scratch area the release fence of the LDS barrier waits for them
// (same +14 %) -- into registers that stay reserved until the end of the kernel, and the successor's prologue finds the lines in
// (or on their way into) this XCD's L2: forward 198 -> 190 us at config 3 (prologue 10.1 k -> 4.9 k cycles).  The backward
// kernels measured no gain from the same trick and do not use it.
struct Touch {
  float t[2];
};
// one dword of line `i` of the rows [0, nrows) x head slice of `base`; the destination register stays reserved (and unread) until
// touch_join() at the very end of the kernel -- the compiler does not know this is a load, so it never waits for it.
// (The value lives in ONE register between the two asm statements only as long as the allocator has no reason to copy it: the
//  forward kernel uses 188 of its 256 VGPRs.  The build's resource remarks and the bit-exact attention tests are the check that
//  this still holds after a change; LTRX_MHA_TOUCH=0 compiles the mechanism out.)
__device__ __forceinline__ float touch_line(const float* __restrict__ base, int i, int nrows, int dk, size_t rs) {
  const int lpr = (dk * 4 + 127) >> 7;                       // 128-byte lines per row of a head slice
  float t = 0.f;
  if (i < nrows * lpr) {
    const float* src = base + (size_t)(i / lpr) * rs + (i % lpr) * 32;
    asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(src) : "memory");
  }
  return t;
}
__device__ __forceinline__ void touch_join(const Touch& x) {
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(x.t[0]), "v"(x.t[1]) : "memory");
}
// the (slate, head) of workgroup blockIdx.x + 256 of the same row block, if there is one
__device__ __forceinline__ bool next_slate(int L, int h, const int* __restrict__ cu, const int* __restrict__ order, Slate& s) {
  const unsigned id = blockIdx.x + 256u;
  if (id >= gridDim.x) return false;
  s.head = id % h;
  s.b = order ? order[id / h] : (int)(id / h);
  s.bh = s.b * h + s.head;
  s.Lmax = L;
  s.row0 = cu ? (size_t)cu[s.b] : (size_t)s.b * L;
  s.len = cu ? cu[s.b + 1] - cu[s.b] : L;
  return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Round 6: the tile loops of all three kernels with FEWER INSTRUCTIONS (the list below is the forward's; the backward's is at its kernels).  profiles/r06_attention_forward_experiments.md: about two thirds of a
// SIMD's vector-instruction time is ADDED to its matrix time and the kernel's parts add up, so the lever is the instruction count of a tile
// (round 5: 24 MFMA + 214 VALU + 17 exp + 29 LDS instructions per wave in the compute half).  What went:
//   * LDS addresses: the per-lane byte offsets of the 4 K-fragment reads, the 8 transposed V reads and the mask-bias read do not depend
//     on the tile -- computed once (13 registers), one v_add per address and tile instead of the swizzle arithmetic (~47 -> 13);
//   * S in ONE accumulator (a dependent MFMA chain costs nothing: r06_mfma_issue_probe.txt) instead of four partial sums added up
//     on the VALU (24 v_pk_add);
//   * the softmax scale folded into Q before its split (prologue), the mask bias only on tiles that contain a masked or
//     out-of-range key (a per-tile flag written by the staging threads): 16 fma + 4 LDS reads on the other tiles;
//   * lazy rescaling: a query's reference maximum only moves when a tile exceeds it by 2^8 (P <= 256: no overflow anywhere in fp32,
//     the bf16 hi / lo split keeps its relative precision; O / l and m + log2 l do not depend on the reference), so the 32
//     accumulator multiplies by alpha run on the first tile and then almost never (wave-uniform branch);
//   * the cross-half maximum with v_permlane32_swap instead of ds_bpermute (an LDS round trip);
//   * packed fp32 arithmetic (v_pk_add_f32) for S - m, the row sum and the lo part of the P split.
// The result differs from round 5's kernel in rounding only (summation order of S, reference of the softmax).
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kLazyTau = 8.0f;
__device__ __forceinline__ float max_across_halves(float x) {     // lane i <-> lane i ^ 32: {lo, lo} and {hi, hi} after the swap
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_across_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// two fp32 values -> one register of two bf16 hi parts and one of two bf16 lo parts (x = hi + lo to 2^-17 relative): ONE conversion
// per pair and part, the hi parts widened back with a shift / a mask, the subtraction packed -- 5 instructions per pair
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <bool PL>
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& l) {
  const bf16x2 hp = {(__bf16)x0, (__bf16)x1};
  h = __builtin_bit_cast(uint32_t, hp);
  if (!PL) {
    const f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    const f32x2 d = f32x2{x0, x1} - hf;
    const bf16x2 lp = {(__bf16)d[0], (__bf16)d[1]};
    l = __builtin_bit_cast(uint32_t, lp);
  }
}
struct TileOffsets {
  int k[4];       // K fragments: byte offset of chunk 2 ks + half of row l31 inside a 32-row tile of the K hi plane (lo plane: + PLANE)
  int v[2][2][2]; // V transposed reads [u][ct][0 / 1]: byte offsets inside a 32-row tile, relative to the V hi plane
  int kb;         // mask bias: byte offset of this lane half's first float inside a tile's 32 floats
};
__device__ __forceinline__ TileOffsets tile_offsets() {     // (the same for every [32 rows][64] bf16 tile of every image plane)
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5, i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int rsub = i16 >> 2, c8 = (i16 & 3) >> 1, b8 = (i16 & 1) * 8;
  TileOffsets f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f.k[ks] = img_off(l31, 2 * ks + half);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int r0 = 16 * u + 4 * half + rsub, chunk = (2 * ct + g16) * 2 + c8;
      f.v[u][ct][0] = img_off(r0, chunk) + b8;
      f.v[u][ct][1] = img_off(r0 + 8, chunk) + b8;
    }
  f.kb = 16 * half;
  return f;
}
// S^T[key = rowmap(r, half)][query = l31] of one tile in one accumulator; `tile` = LDS byte address of the tile's first row in the K hi plane
template <bool PL>
__device__ __forceinline__ f32x16 rows_mma(const unsigned char* tile, const TileOffsets& fo, const bf16x8 (&fh)[4], const bf16x8 (&fl)[4]) {
  bf16x8 xh[4], xl[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    xh[ks] = *reinterpret_cast<const bf16x8*>(tile + fo.k[ks]);
    if (!PL) xl[ks] = *reinterpret_cast<const bf16x8*>(tile + fo.k[ks] + PLANE);
  }
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
  if (!PL) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a = LTRX_MFMA(xl[ks], fh[ks], a);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a = LTRX_MFMA(xh[ks], fl[ks], a);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) a = LTRX_MFMA(xh[ks], fh[ks], a);
  return a;
}
// O^T += V^T P^T of one tile; `tile` = LDS byte address of the tile's first row in the V hi plane
template <bool PL>
__device__ __forceinline__ void cols_mma(const unsigned char* tile, const TileOffsets& fo, const f32x16& p, f32x16 (&out)[2]) {
  ColFrags f;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + fo.v[u][ct][0]));
      const bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + fo.v[u][ct][1]));
      f.h[u][ct] = bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      if (!PL) {
        const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + fo.v[u][ct][0] + PLANE));
        const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + fo.v[u][ct][1] + PLANE));
        f.l[u][ct] = bf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      }
    }
  bf16x8 ph[2], pl[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    u32x4 hw, lw = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t hj, lj = 0u;
      split_pair<PL>(p[8 * u + 2 * j], p[8 * u + 2 * j + 1], hj, lj);
      hw[j] = hj;
      lw[j] = lj;
    }
    ph[u] = __builtin_bit_cast(bf16x8, hw);
    pl[u] = __builtin_bit_cast(bf16x8, lw);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (!PL) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) out[ct] = LTRX_MFMA(f.l[u][ct], ph[u], out[ct]);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) out[ct] = LTRX_MFMA(f.h[u][ct], pl[u], out[ct]);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) out[ct] = LTRX_MFMA(f.h[u][ct], ph[u], out[ct]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward: wave w owns queries 32 w .. 32 w + 31 of the slate
// ------------------------------------------------------------------------------------------------------------------
template <bool DROP, bool PL>
__global__ void __launch_bounds__(512) ltrx_mha_fwd_res_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const uint8_t* __restrict__ kpm, int L,
                                                               int h, int dk, int rs, float* __restrict__ o, int ors,
                                                               float* __restrict__ lse, float scale, DropCfg drop,
                                                               const uint32_t* __restrict__ drop_step, const int* __restrict__ cu,
                                                               const int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kimg = smem;
  unsigned char* vimg = smem + 2 * PLANE;
  float* kbias = reinterpret_cast<float*>(smem + 4 * PLANE);
  int* kflag = reinterpret_cast<int*>(kbias + RMAX);           // per ring slot: does the tile hold a masked / out-of-range key?
  Touch tch = {{0.f, 0.f}};
  if (DROP && drop_step) drop.seed ^= drop_step[0] * 0x9E3779B9u;
  const Slate sl = which_slate(L, h, cu, order);
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int len = sl.len;
  const float* qb = q + sl.row0 * rs + (size_t)sl.head * dk;
  const float* src = (threadIdx.x < 256 ? k : v) + sl.row0 * rs + (size_t)sl.head * dk;     // this thread's streamed tensor
  unsigned char* dst = threadIdx.x < 256 ? kimg : vimg;
  TileRegs tr;
  STAMP(32, 0);
  tile_gload(tr, src, 0, len, dk, rs);
  if ((int)(blockIdx.y * RMAX) >= len) return;        // whole workgroup beyond this slate (uniform: before any barrier)
  uint8_t km_n = 0;                                   // threads 0-31: padding-mask byte of key 32 kt + threadIdx.x of the tile being staged
  if (threadIdx.x < 32 && kpm && (int)threadIdx.x < len) km_n = kpm[sl.row0 + threadIdx.x];
  const int q0 = blockIdx.y * RMAX + wave * 32;
  const bool active = q0 < len;                 // (inactive waves still help staging and take every barrier)
  bf16x8 qh[4], ql[4];
  {   // (scratch: this wave's 8 KB of ring slots 4-7 of the image planes, first written by tile 4 -- four barriers from here)
    FixedRegs fq;
    fixed_gload(fq, qb, q0, len, dk, rs);
    // (Q is scaled by scale * log2 e before its split: S comes out of the MFMAs in the softmax's log2 domain)
    fixed_finish(qh, ql, fq, reinterpret_cast<f32x4*>(smem + (size_t)(wave >> 1) * PLANE + PLANE / 2 + (size_t)(wave & 1) * 8192),
                 scale * kLog2e);
  }
  const TileOffsets fo = tile_offsets();
  if (LTRX_MHA_PRIO_HALF && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1);
  f32x16 oacc[2];
  zero2(oacc);
  float m = -INFINITY, l = 0.f;
  const uint32_t drow = DROP ? drop_row_seed(drop, sl.bh, sl.Lmax, q0 + (lane & 31)) : 0u;
  const int nkt = (len + 31) / 32;
  for (int kt = 0; kt < nkt; ++kt) {
    STAMP(kt, 0);
    tile_sstore<PL>(dst, kt, tr);
    if (threadIdx.x < 32) {
      const bool off = kt * 32 + (int)threadIdx.x >= len || km_n;
      kbias[(kt & 7) * 32 + threadIdx.x] = off ? -INFINITY : 0.f;
      const bool any_off = __builtin_amdgcn_ballot_w64(off) != 0;
      if (threadIdx.x == 0) kflag[kt & 7] = any_off ? 1 : 0;
    }
    if (kt + 1 < nkt) {
      tile_gload(tr, src, kt + 1, len, dk, rs);       // in flight during this tile's MFMAs
      // the mask bytes of the next tile travel with it (loaded at the top of their own tile, the round trip sat between wave 0 and
      // the barrier every other wave was already waiting at)
      if (threadIdx.x < 32 && kpm && (kt + 1) * 32 + (int)threadIdx.x < len) km_n = kpm[sl.row0 + (kt + 1) * 32 + threadIdx.x];
    }
    if (LTRX_MHA_TOUCH && kt == nkt - 1 && blockIdx.y == 0) {     // (after the last wait on a streamed tile: nothing waits for these)
      Slate nx;
      if (next_slate(L, h, cu, order, nx)) {
        const size_t o = nx.row0 * rs + (size_t)nx.head * dk;
        tch.t[0] = touch_line(q + o, threadIdx.x, nx.len, dk, rs);
        tch.t[1] = touch_line(threadIdx.x < 256 ? k + o : v + o, threadIdx.x & 255, min(nx.len, 32), dk, rs);
      }
    }
    STAMP(kt, 1);
    lds_only_barrier();
    STAMP(kt, 2);
    if (!active) continue;
    const int slot = (kt & 7) * 32;                            // ring slot (rows of the LDS images) of this tile
    const unsigned char* ktile = kimg + slot * (DK * 2);
    f32x16 s = rows_mma<PL>(ktile, fo, qh, ql);              // S^T[key = rowmap(r, half)][query = l31], log2 domain
    STAMP(kt, 3);
    if (__builtin_amdgcn_readfirstlane(kflag[kt & 7])) {       // (wave-uniform) a masked or out-of-range key in this tile
      const float* kb = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(kbias + slot) + fo.kb);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + 8 * g);       // keys 8 g + 4 half + 0..3 = rowmap(4 g + e, half)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * g + e] += b4[e];
      }
    }
    float mt = fmaxf(s[0], s[1]);
#pragma unroll
    for (int r = 2; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = max_across_halves(mt);
    const bool grow = mt > m + kLazyTau;                       // (first tile: m = -inf; nothing but masked keys so far: mt = -inf -> no)
    const float mn = grow ? mt : m;
    const float mref = (mn == -INFINITY) ? 0.f : mn;
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {              // some query's reference moves: rescale what was accumulated under the old one
      const float alpha = grow ? fast_exp2(m - mref) : 1.0f;
      l *= alpha;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[ct][r] *= alpha;
    }
    m = mn;
    f32x16 p;
    f32x2 ps2 = {0.f, 0.f};
    const f32x2 mref2 = {mref, mref};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 t = f32x2{s[r], s[r + 1]} - mref2;
      const f32x2 e2 = {fast_exp2(t[0]), fast_exp2(t[1])};
      ps2 += e2;
      p[r] = e2[0];
      p[r + 1] = e2[1];
    }
    l += ps2[0] + ps2[1];
    if (DROP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r] *= drop_scale_rk(drop, drow, kt * 32 + rowmap(r, half));
    }
    STAMP(kt, 4);
    cols_mma<PL>(vimg + slot * (DK * 2), fo, p, oacc);      // O^T[d][query] += V^T[d][key] P^T[key][query]
    STAMP(kt, 5);
  }
  STAMP(33, 0);
  if (!active) {
    if (LTRX_MHA_TOUCH) touch_join(tch);
    return;
  }
  const float lt = sum_across_halves(l);
  const float inv = (lt > 0.f) ? 1.0f / lt : 0.f;
  store_rows(o + sl.row0 * ors + (size_t)sl.head * dk, q0, len, dk, ors, oacc, inv, end_scratch(smem, nkt - 1, wave));
  const int qrow = q0 + (lane & 31);
  if (half == 0 && qrow < len) lse[((size_t)sl.b * h + sl.head) * sl.Lmax + qrow] = (lt > 0.f) ? (m + log2f(lt)) * kLn2 : 0.f;
  STAMP(34, 0);
  if (LTRX_MHA_TOUCH) touch_join(tch);
}

// ------------------------------------------------------------------------------------------------------------------
// backward, second kernel: dQ = dS K.  dS comes from the dK/dV kernel's workspace (row-major [query][key] per (slate, head),
// row stride LK) -- S, P and dP are computed ONCE per backward, in the kernel below.  A wave owns 32 queries and turns its
// [32 queries][32 keys] fp32 tile of dS into the MFMA's D layout (lane = query, registers = keys) through a wave-private 4-KB
// LDS area: coalesced 128-byte rows in, one row per lane out (16-byte chunk c of row r at position c ^ ((r >> 1) & 7):
// conflict-free both ways; no barrier is involved in that exchange).  K lives in LDS in chunks of 8 tiles (256 keys, the ring of
// the other kernels), staged by the workgroup between two barriers per chunk.
// What bounds this kernel is memory latency and phase, not the 12 MFMAs of a tile.  Slates of one chunk (<= 256): FOUR-wave
// workgroups (NW = 4, 128 queries) with 80 KB of LDS, so that two of them share a CU and one streams while the other is in its
// prologue or stores its result (one eight-wave workgroup per CU: 171 -> 157 us at the bench shape), the dS tiles of D = 6 steps in
// flight in registers.  Longer slates: eight-wave workgroups (every workgroup stages ALL the slate's keys, so fewer, larger ones:
// 456 vs 500 us at 64 x 1024), D = 4 and the next chunk of K prefetched.
// ------------------------------------------------------------------------------------------------------------------
struct DsRegs {
  f32x4 x[4];
};
__device__ __forceinline__ void ds_gload(DsRegs& d, const float* __restrict__ rows, int tile, int LK) {
  const int lane = threadIdx.x & 63;
  const float* p = rows + (size_t)(lane >> 3) * LK + tile * 32 + (lane & 7) * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) d.x[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)(8 * j) * LK));
}
template <bool PL, int NW, int D, bool KPF>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) ltrx_mha_bwd_dq_res_kernel(const float* __restrict__ k, const float* __restrict__ dsw, int LK, int L,
                                                                  int h, int dk, int rs, float* __restrict__ dq, int drs,
                                                                  const int* __restrict__ cu, const int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kimg = smem;
  const Slate sl = which_slate(L, h, cu, order);
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31;
  f32x4* xs = reinterpret_cast<f32x4*>(smem + 2 * PLANE) + wave * (32 * 32 / 4);
  const int len = sl.len;
  constexpr int TPT = 32 / NW;                         // tiles of a chunk staged per thread (2048 (row, 8-column) positions per chunk)
  if ((int)(blockIdx.y * (32 * NW)) >= len) return;
  const float* src = k + sl.row0 * rs + (size_t)sl.head * dk;
  const int q0 = blockIdx.y * (32 * NW) + wave * 32;
  const int sub = (NW == 8) ? (int)(threadIdx.x >> 8) : 0;     // NW = 8: threads 0-255 stage the even tiles, 256-511 the odd ones
  const bool active = q0 < len;
  const float* rows = dsw + ((size_t)sl.b * h + sl.head) * LK * LK + (size_t)q0 * LK;
  const int nkt = (len + 31) / 32, nchunk = KPF ? (nkt + 7) / 8 : 1;      // (!KPF: launched for L <= 256 only)
  static_assert(KPF ? (8 % D == 0) : true, "the register slot of tile 8 c + s must not depend on c");
  TileRegs tr[TPT];                                    // thread = one (row, 8 columns) position of TPT of the chunk's 8 tiles
#pragma unroll
  for (int j = 0; j < TPT; ++j) tile_gload(tr[j], src, (8 / TPT) * j + sub, len, dk, rs);       // (tiles beyond the slate are zeros)
  DsRegs dr[D];
  if (active) {
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < nkt) ds_gload(dr[s], rows, s, LK);
  }
  f32x16 dqacc[2];
  zero2(dqacc);
  const TileOffsets fo = tile_offsets();
  for (int c = 0; c < nchunk; ++c) {
    if (c > 0) lds_only_barrier();                     // every wave is done with the previous chunk of K
#pragma unroll
    for (int j = 0; j < TPT; ++j) tile_sstore<PL>(kimg, (8 / TPT) * j + sub, tr[j]);
    if (KPF && c + 1 < nchunk) {
#pragma unroll
      for (int j = 0; j < TPT; ++j) tile_gload(tr[j], src, 8 * (c + 1) + (8 / TPT) * j + sub, len, dk, rs);
    }
    lds_only_barrier();
    if (!active) continue;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int i = 8 * c + s;
      if (i >= nkt) break;
      DsRegs& cur = dr[s % D];                         // = (8 c + s) % D
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = (lane >> 3) + 8 * j;
        xs[r * 8 + ((lane & 7) ^ ((r >> 1) & 7))] = cur.x[j];
      }
      if (i + D < nkt) ds_gload(cur, rows, i + D, LK);
      f32x16 ds;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 x = xs[l31 * 8 + ((2 * g + half) ^ ((l31 >> 1) & 7))];     // keys 8 g + 4 half + 0..3 of the tile
        ds[4 * g + 0] = x.x;
        ds[4 * g + 1] = x.y;
        ds[4 * g + 2] = x.z;
        ds[4 * g + 3] = x.w;
      }
      cols_mma<PL>(kimg + s * 32 * (DK * 2), fo, ds, dqacc);      // dQ^T[d][query] += K^T[d][key] dS^T[key][query]
    }
  }
  if (!active) return;
  store_rows(dq + sl.row0 * drs + (size_t)sl.head * dk, q0, len, dk, drs, dqacc, 1.0f, xs);      // (xs: this wave's dS exchange area)
}

// ------------------------------------------------------------------------------------------------------------------
// backward, first kernel: dK, dV, delta, and dS for the dQ kernel above.  Q and dO resident; wave owns 32 keys (K and V fragments
// in registers).  delta_q = <dO_q, O_q> in fp32 is computed by the threads that stage the dO tiles: the matching O values travel
// with every dO tile (8 floats per thread, 8 threads per row), so delta costs no extra pass, no extra kernel and -- unlike a
// prologue that reads all of O and dO up front, measured +60 us -- adds nothing to the burst of compulsory loads every workgroup
// starts with.  The dS tile a wave has in its registers after the softmax backward -- [query = rowmap(r, half)][key = l31] -- is
// written to the workspace row-major (one 128-byte segment per half-wave and register, nontemporal: it is read once, by another
// kernel, after 0.5 GB of other traffic).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tile_delta(const TileRegs& g, const TileRegs& a) {       // this thread's 8 columns, then the row's 8 threads
  float d = g.a.x * a.a.x + g.a.y * a.a.y + g.a.z * a.a.z + g.a.w * a.a.w;
  d += g.b.x * a.b.x + g.b.y * a.b.y + g.b.z * a.b.z + g.b.w * a.b.w;
  d += __shfl_xor(d, 4, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 1, 64);
  return d;
}
template <bool DROP, bool PL>
__global__ void __launch_bounds__(512) ltrx_mha_bwd_dkdv_res_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const uint8_t* __restrict__ kpm,
    const float* __restrict__ o, const float* __restrict__ dout, const float* __restrict__ lse, int L, int h, int dk, int rs, int ors,
    float* __restrict__ dkout, float* __restrict__ dvout, int drs, float* __restrict__ dsw, int LK, float scale, DropCfg drop,
    const uint32_t* __restrict__ drop_step, const int* __restrict__ cu, const int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* qimg = smem;
  unsigned char* doimg = smem + 2 * PLANE;
  float* lse_t = reinterpret_cast<float*>(smem + 4 * PLANE);
  float* del_t = lse_t + RMAX;
  uint32_t* drow_t = reinterpret_cast<uint32_t*>(del_t + RMAX);
  if (DROP && drop_step) drop.seed ^= drop_step[0] * 0x9E3779B9u;
  const Slate sl = which_slate(L, h, cu, order);
  // (wave index in a scalar register: `active` below must be a SCALAR branch -- as a divergent one both sides run under exec
  //  masks and the tile loads of the inactive side make the active side wait vmcnt(0) before its own)
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int len = sl.len;
  const bool first = wave < 4;                         // waves 0-3 stage Q, waves 4-7 stage dO (+ O for delta)
  const float* src = first ? q + sl.row0 * rs + (size_t)sl.head * dk : dout + sl.row0 * ors + (size_t)sl.head * dk;
  const float* osrc = o + sl.row0 * ors + (size_t)sl.head * dk;
  const size_t srs = first ? (size_t)rs : (size_t)ors;
  unsigned char* dst = first ? qimg : doimg;
  TileRegs tr, to;
  DSTAMP(32, 0);
  tile_gload(tr, src, 0, len, dk, srs);
  if (!first) tile_gload(to, osrc, 0, len, dk, ors);
  const size_t statb = ((size_t)sl.b * h + sl.head) * sl.Lmax;
  float lse_n = 0.f;                                   // threads 0-31: LSE of query 32 qt + threadIdx.x of the tile being staged
  if (threadIdx.x < 32 && (int)threadIdx.x < len) lse_n = lse[statb + threadIdx.x];
  if ((int)(blockIdx.y * RMAX) >= len) return;
  const int k0 = blockIdx.y * RMAX + wave * 32;
  bf16x8 kh[4], kl[4], vh[4], vl[4];
  const bool active = k0 < len;
  const int key = k0 + (lane & 31);
  bool key_masked;
  {   // every load of the prologue is issued before the first wait (K, V, the mask byte: one memory round trip, not three)
    FixedRegs fk, fv;
    fixed_gload(fk, k + sl.row0 * rs + (size_t)sl.head * dk, k0, len, dk, rs);
    fixed_gload(fv, v + sl.row0 * rs + (size_t)sl.head * dk, k0, len, dk, rs);
    const uint8_t km = kpm ? kpm[sl.row0 + (key < len ? key : 0)] : (uint8_t)0;
    // scratch: this wave's 8 KB of ring slots 4-7 of the image planes (first written by tile 4, four barriers from here)
    f32x4* scratch = reinterpret_cast<f32x4*>(smem + (size_t)(wave >> 1) * PLANE + PLANE / 2 + (size_t)(wave & 1) * 8192);
    // (round 6: the two factors every element of a tile used to be multiplied by are folded into the fixed operands before their split --
    //  K carries scale * log2 e, so S comes out of the MFMAs in the softmax's log2 domain; V carries scale, so dO V^T is scale * dP, and
    //  the staging threads store scale * delta: dS = P (scale dP - scale delta))
    fixed_finish(kh, kl, fk, scratch, scale * kLog2e);
    fixed_finish(vh, vl, fv, scratch, scale);
    key_masked = (key >= len) || km != 0;
  }
  const float kbias = key_masked ? -INFINITY : 0.f;
  const f32x2 kb2 = {kbias, kbias};
  const TileOffsets fo = tile_offsets();
  f32x16 dkacc[2], dvacc[2];
  zero2(dkacc);
  zero2(dvacc);
  const int nqt = (len + 31) / 32;
  if (LTRX_MHA_PRIO_HALF && wave >= 4) __builtin_amdgcn_s_setprio(1);
  // dS tile addressing: a wave-uniform base (scalar registers) + ONE 32-bit per-lane offset, so that the 16 stores of a tile cost
  // no vector registers beyond the data (the kernel sits at the 256-register limit)
  float* const dsp = dsw + ((size_t)sl.b * h + sl.head) * LK * LK + (blockIdx.y * RMAX + wave * 32);
  const int dso = (4 * half * LK + (lane & 31)) * 4;      // bytes
  DSTAMP(32, 1);
  for (int qt = 0; qt < nqt; ++qt) {
    DSTAMP(qt, 0);
    if (!first) {                                                // (rows beyond the slate were loaded as zeros: delta 0)
      const float d = tile_delta(tr, to);
      if ((threadIdx.x & 7) == 0) del_t[(qt & 7) * 32 + ((threadIdx.x & 255) >> 3)] = d * scale;
    }
    tile_sstore<PL>(dst, qt, tr);
    if (threadIdx.x < 32) {                                      // per-query statistics of this tile (ring slot)
      const int qr = qt * 32 + threadIdx.x, sl_ = (qt & 7) * 32 + threadIdx.x;
      lse_t[sl_] = (qr < len) ? lse_n * kLog2e : INFINITY;       // +inf -> P = exp2(-inf) = 0 for rows >= len
      if (DROP) drow_t[sl_] = drop_row_seed(drop, sl.bh, sl.Lmax, qr);
    }
    DSTAMP(qt, 1);
    lds_only_barrier();
    DSTAMP(qt, 2);
    if (!active) {
      if (qt + 1 < nqt) {
        tile_gload(tr, src, qt + 1, len, dk, srs);
        if (!first) tile_gload(to, osrc, qt + 1, len, dk, ors);
      }
      continue;
    }
    const int slot = (qt & 7) * 32;
    // (order chosen for register pressure: both 16-register products first, then the two accumulations; the fences keep the
    //  scheduler from hoisting the second accumulation's transposed reads above the first)
    const unsigned char* qtile = qimg + slot * (DK * 2);
    const unsigned char* dotile = doimg + slot * (DK * 2);
    f32x16 p = rows_mma<PL>(qtile, fo, kh, kl);                       // S[query = rowmap(r, half)][key = l31], log2 domain
    f32x16 ds = rows_mma<PL>(dotile, fo, vh, vl);                     // scale * dP[query][key] = scale * dO V^T
    DSTAMP(qt, 3);
    {
      const float* lse_p = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(lse_t + slot) + fo.kb);
      const float* del_p = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(del_t + slot) + fo.kb);
#pragma unroll
      for (int g = 0; g < 4; ++g) {                                   // rows 8 g + 4 half + 0..3 = rowmap(4 g + e, half)
        const f32x4 ls4 = *reinterpret_cast<const f32x4*>(lse_p + 8 * g);
        const f32x4 dl4 = *reinterpret_cast<const f32x4*>(del_p + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const int r = 4 * g + e;
          const f32x2 t = (f32x2{p[r], p[r + 1]} - f32x2{ls4[e], ls4[e + 1]}) + kb2;
          const f32x2 pr = {fast_exp2(t[0]), fast_exp2(t[1])};
          f32x2 dm = {1.0f, 1.0f};
          if (DROP) {
            const int qr = slot + rowmap(r, half);
            dm = f32x2{drop_scale_rk(drop, drow_t[qr], key), drop_scale_rk(drop, drow_t[qr + 1], key)};
          }
          const f32x2 dpv = {ds[r], ds[r + 1]};
          const f32x2 x = (DROP ? dpv * dm : dpv) - f32x2{dl4[e], dl4[e + 1]};
          const f32x2 dsv = pr * x;                                   // dS
          const f32x2 pm = DROP ? pr * dm : pr;                       // P M
          ds[r] = dsv[0];
          ds[r + 1] = dsv[1];
          p[r] = pm[0];
          p[r + 1] = pm[1];
        }
      }
    }
    DSTAMP(qt, 4);
    {
      const float* const t = dsp + (size_t)(qt * 32) * LK;
#pragma unroll
      for (int r = 0; r < 16; ++r)         // (asm: hipcc otherwise keeps 16 loop-invariant 64-bit vector addresses = 32 registers)
        asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(dso), "v"(ds[r]), "s"(t + (size_t)((r & 3) + 8 * (r >> 2)) * LK));
      // (round 6 probe: the same tile as four 16-byte stores per lane in a [key][query] layout -- 32-byte pieces in 32 rows -- made the
      //  tile 9-13 k cycles instead of 7.1 k: sixteen 128-byte row segments per instruction pair beat four scattered wide stores)
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile's loads go out AFTER the dS stores: the wait that precedes the next tile_sstore is vmcnt(0) and the memory
    // counter retires in order -- issued before the stores (at the barrier, as in the other kernels) every tile waited for the write
    // acknowledgements of its 16 stores; now those hide behind the loads' own latency and the two products below
    if (qt + 1 < nqt) {
      tile_gload(tr, src, qt + 1, len, dk, srs);
      if (!first) tile_gload(to, osrc, qt + 1, len, dk, ors);
      // (wave 0 is always active.)  The statistics of the NEXT tile travel with its operands: loaded at the top of the tile that
      // uses them, the round trip sat between every workgroup-wide barrier and wave 0's arrival at it
      if (threadIdx.x < 32 && (qt + 1) * 32 + (int)threadIdx.x < len) lse_n = lse[statb + (qt + 1) * 32 + threadIdx.x];
    }
    DSTAMP(qt, 5);
    __builtin_amdgcn_sched_barrier(0);
    cols_mma<PL>(dotile, fo, p, dvacc);                              // dV^T[d][key] += dO^T[d][query] (P M)[query][key]
    DSTAMP(qt, 6);
    __builtin_amdgcn_sched_barrier(0);
    cols_mma<PL>(qtile, fo, ds, dkacc);                              // dK^T[d][key] += Q^T[d][query] dS[query][key]
    DSTAMP(qt, 7);
    __builtin_amdgcn_sched_barrier(0);
  }
  DSTAMP(33, 0);
  if (!active) return;
  f32x4* const es = end_scratch(smem, nqt - 1, wave);
  store_rows(dkout + sl.row0 * drs + (size_t)sl.head * dk, k0, len, dk, drs, dkacc, 1.0f, es);
  store_rows(dvout + sl.row0 * drs + (size_t)sl.head * dk, k0, len, dk, drs, dvacc, 1.0f, es);
  DSTAMP(34, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// host launchers (called from ltrx_mha.hip's C entry points when the shape fits: L <= 256, 32 < d_k <= 64)
// ------------------------------------------------------------------------------------------------------------------
static constexpr size_t RES_SMEM = RES_STATS;

bool ltrx_mha_res_fits(int L, int dk) { return L > 0 && (L + RMAX - 1) / RMAX <= 65535 && dk > 32 && dk <= DK; }
// (the dS exchange is B h LK^2 floats: bounded by the longest slate the losses take, LTRX_MAX_SLATE_LEN)
bool ltrx_mha_res_bwd_fits(int L, int dk) { return ltrx_mha_res_fits(L, dk) && L <= LTRX_MAX_SLATE_LEN; }
static int ds_stride(int L) { return (L + XROW - 1) / XROW * XROW; }
// workspace of the backward: dS[B, h, LK, LK], LK = L rounded up to 64
size_t ltrx_mha_res_bwd_ws_bytes(int B, int L, int h) {
  const size_t lk = (size_t)ds_stride(L);
  return (size_t)B * h * lk * lk * sizeof(float);
}

template <typename K>
static int res_attr(K kernel, size_t bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? LTRX_OK : LTRX_EHIP;
}

int ltrx_mha_fwd_res_launch(const float* q, const float* k, const float* v, const uint8_t* kpm, int B, int L, int h, int dk, int rs,
                            float* o, int ors, float* lse, float p_drop, uint32_t seed, const uint32_t* seed_step, const int* cu,
                            const int* order, bool plain, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  const int arc = ltrx_once_per_device(attr_done, []() {
    if (res_attr(ltrx_mha_fwd_res_kernel<false, false>, RES_SMEM) != LTRX_OK || res_attr(ltrx_mha_fwd_res_kernel<true, false>, RES_SMEM) != LTRX_OK ||
        res_attr(ltrx_mha_fwd_res_kernel<false, true>, RES_SMEM) != LTRX_OK || res_attr(ltrx_mha_fwd_res_kernel<true, true>, RES_SMEM) != LTRX_OK)
      return LTRX_EHIP;
    return LTRX_OK;
  });
  if (arc != LTRX_OK) return arc;
  const DropCfg drop = ltrx_make_drop(p_drop, seed);
  const float scale = 1.0f / sqrtf((float)dk);
  const dim3 grid(B * h, (L + RMAX - 1) / RMAX);
#define LTRX_FWD(D_, P_)                                                                                                          \
  hipLaunchKernelGGL((ltrx_mha_fwd_res_kernel<D_, P_>), grid, dim3(512), RES_SMEM, s, q, k, v, kpm, L, h, dk, rs, o, ors, lse, scale, \
                     drop, seed_step, cu, order)
  if (drop.thresh != 0u) {
    if (plain) LTRX_FWD(true, true); else LTRX_FWD(true, false);
  } else {
    if (plain) LTRX_FWD(false, true); else LTRX_FWD(false, false);
  }
#undef LTRX_FWD
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

int ltrx_mha_bwd_res_launch(const float* q, const float* k, const float* v, const uint8_t* kpm, const float* o, const float* dout,
                            const float* lse, int B, int L, int h, int dk, int rs, int ors, float* dq, float* dkk, float* dv, int drs,
                            void* ws, float p_drop, uint32_t seed, const uint32_t* seed_step, const int* cu, const int* order,
                            bool plain, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  const int arc = ltrx_once_per_device(attr_done, []() {
    if (res_attr(ltrx_mha_bwd_dq_res_kernel<false, 4, 6, false>, dq_smem(4)) != LTRX_OK || res_attr(ltrx_mha_bwd_dq_res_kernel<true, 4, 6, false>, dq_smem(4)) != LTRX_OK ||
        res_attr(ltrx_mha_bwd_dq_res_kernel<false, 8, 4, true>, dq_smem(8)) != LTRX_OK || res_attr(ltrx_mha_bwd_dq_res_kernel<true, 8, 4, true>, dq_smem(8)) != LTRX_OK ||
        res_attr(ltrx_mha_bwd_dkdv_res_kernel<false, false>, RES_SMEM) != LTRX_OK ||
        res_attr(ltrx_mha_bwd_dkdv_res_kernel<true, false>, RES_SMEM) != LTRX_OK ||
        res_attr(ltrx_mha_bwd_dkdv_res_kernel<false, true>, RES_SMEM) != LTRX_OK ||
        res_attr(ltrx_mha_bwd_dkdv_res_kernel<true, true>, RES_SMEM) != LTRX_OK)
      return LTRX_EHIP;
    return LTRX_OK;
  });
  if (arc != LTRX_OK) return arc;
  const DropCfg drop = ltrx_make_drop(p_drop, seed);
  const float scale = 1.0f / sqrtf((float)dk);
  const dim3 grid(B * h, (L + RMAX - 1) / RMAX);
  const int LK = ds_stride(L);
  float* dsw = (float*)ws;
#define LTRX_DKDV(D_, P_)                                                                                                          \
  hipLaunchKernelGGL((ltrx_mha_bwd_dkdv_res_kernel<D_, P_>), grid, dim3(512), RES_SMEM, s, q, k, v, kpm, o, dout, lse, L, h, dk,  \
                     rs, ors, dkk, dv, drs, dsw, LK, scale, drop, seed_step, cu, order)
  if (drop.thresh != 0u) {
    if (plain) LTRX_DKDV(true, true); else LTRX_DKDV(true, false);
  } else {
    if (plain) LTRX_DKDV(false, true); else LTRX_DKDV(false, false);
  }
#undef LTRX_DKDV
  LTRX_LAUNCH_CHECK();
#define LTRX_DQ(P_, NW_, D_, K_)                                                                                                   \
  hipLaunchKernelGGL((ltrx_mha_bwd_dq_res_kernel<P_, NW_, D_, K_>), dim3(B * h, (L + 32 * NW_ - 1) / (32 * NW_)), dim3(64 * NW_),      \
                     dq_smem(NW_), s, k, dsw, LK, L, h, dk, rs, dq, drs, cu, order)
  if (grid.y == 1) {
    if (plain) LTRX_DQ(true, 4, 6, false); else LTRX_DQ(false, 4, 6, false);
  } else {
    if (plain) LTRX_DQ(true, 8, 4, true); else LTRX_DQ(false, 8, 4, true);
  }
#undef LTRX_DQ
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

#ifdef LTRX_MHA_STAMP
extern "C" int ltrx_debug_mha_stamps(unsigned long long* host_dst) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_mha_stamps), sizeof(unsigned long long) * 8 * 40 * 8) == hipSuccess ? 0 : 1;
}
#endif
