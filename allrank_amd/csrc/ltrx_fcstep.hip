// Slate-resident training step of BASELINE configs[1]: FCModel([H]) -> OutputLayer(H, 1) -> ListNet, forward AND backward, in ONE
// kernel that reads the feature tensor from HBM exactly once.
//
//   reference path:  allrank/models/model.py:35-44 (FCModel.forward: act(Linear(x))), :111-128 (OutputLayer: Linear(H, 1).squeeze),
//                    allrank/models/losses/listNet.py:8-30, autograd backward of the three, allrank/training/train_utils.py:18-29.
//
// Why: 53 kFLOP per item against 552 B per item -- this is the one BASELINE config the HBM roofline binds.  As a launch sequence
// (FC GEMM, score head, ListNet, head backward, weight-gradient GEMM, slab reduce, Adam: ~15 launches) the step read x three
// times and sat at 2.4 % of the HBM roof (profiles/r03_bench_fc_listnet_kernel_stats.md).  Here one workgroup owns one slate at
// a time:
//   * x[L, F] (130 KB at L = 240, F = 136) is streamed into LDS once, split into bf16 hi / lo images on the way (the same
//     three-product arithmetic as ltrx_gemm.hip: x w ~= xh wh + xh wl + xl wh, fp32 accumulate), and is used twice from there:
//     row-wise (ds_read_b128) as the A operand of h = x W1^T, column-wise (ds_read_b64_tr_b16) as the A operand of dW1^T = x^T dh;
//   * 2 NHB waves (NHB = ceil(H / 16) <= 6): wave (hb, rh) owns hidden units 16 hb .. +15 and slate rows 128 rh .. +127.  Its W1
//     rows stay in registers as pre-split MFMA B fragments for the whole kernel (v_mfma_f32_16x16x32_bf16), its h tiles stay in
//     the MFMA D layout and feed the weight-gradient MFMAs as B fragments without moving between lanes (the k index of a step
//     is permuted to the D layout's row order: rows {4 kg + i} of two 16-row tiles), its dW1 tile (16 x F) stays in accumulators
//     ACROSS the slates the workgroup processes -- a workgroup writes ONE partial gradient, however many slates it owns;
//   * scores: in-lane products with w_out, a 16-lane DPP row reduction, six partials per row through LDS; ListNet (value and
//     d loss / d score) by one wave with wave reductions; scores (and optionally d loss / d score) are the only per-item writes.
//   * a second kernel sums the per-workgroup partials in a fixed order (deterministic) and, optionally, applies torch.optim.Adam
//     / AdamW in the same pass (ltrx_train.hip's update rule), so a step is two launches.
// HBM traffic per item: 4 F (features, once) + 4 (label) + 4 (score) [+ 4 d loss / d score when asked for] -- SURVEY.md 8(d)'s
// algorithmic bytes; plus one partial gradient (4 (H F + 2 H + 8) B) per WORKGROUP.
//
// Register budget (168 per thread at 12 waves): dW1 tile 36 + h tiles 32 + the forward's W1 fragments 40 (re-read per slate) or the
// backward's working set ~45, + 16 for the prefetch of a third of the next slate (FC_NPRE).  Requesting the WHOLE next slate a slate ahead (48 registers through
// the backward; forward reordered K-step-outer so that the W1 fragments stream) was built in round 4 and spills 120-330 registers --
// the second half of a slate's loads stays exposed at the top of its iteration.
//
// LDS image of one plane (hi or lo): two panels [256 rows][64 cols] bf16 (128-byte rows; 16-byte chunk c of row r at position
// c ^ s(r), s(r) = 2 ((r >> 1) & 3): conflict-free for the row-wise b128 reads of a 16x16x32 A fragment AND for the [8 rows][16 cols]
// footprint of a half-wave's transposed read) and a tail panel [256][16] for columns 128..143 (32-byte rows, no swizzle needed).
#include "ltrx_device.h"

using namespace ltrx;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

namespace {

constexpr int FC_ROWS = 256;                          // slate rows held in LDS
constexpr int FC_PANEL = FC_ROWS * 128;               // bytes of a [256][64] bf16 panel
constexpr int FC_TAIL = FC_ROWS * 32;                 // bytes of the [256][16] tail panel
constexpr int FC_PLANE = 2 * FC_PANEL + FC_TAIL;      // 73728
constexpr int FC_MAXF = 144;
constexpr int FC_MAXH = 96;
constexpr int FC_MAXHB = FC_MAXH / 16;
constexpr int FC_NKS = 5;                             // K-steps of 32 features (forward)
constexpr int FC_NFB = 9;                             // 16-feature blocks (weight gradient)
constexpr int FC_NLD = 12;                            // float4 loads per thread and slate at the largest shape (768 threads)
#ifndef LTRX_FC_NPRE
#define LTRX_FC_NPRE 4
#endif
// ... of which this many are requested one slate ahead.  Round 5 sweep (tools/lab/lib_variant.sh -DLTRX_FC_NPRE=n, tools/fcstep_check.py
// --timing-only, two interleaved rounds on one box; us per step at 256 / 2048 slates): n = 1: 35.4 / 169, 2: 34.8 / 167, 3: 34.3 / 167,
// 4: 35.0 / 164, 5: 36.6 / 171, 6 (rounds 4): 38.1 / 174, 8: 38.6 / 178, 9: 39.8 / 181 -- every float4 held across the slate costs four of the
// 168 registers, and beyond a third of the slate the spills it causes (13-54 VGPRs at n = 8) cost more than the latency it hides.
constexpr int FC_NPRE = LTRX_FC_NPRE;
constexpr size_t FC_SMEM = 2 * (size_t)FC_PLANE + (size_t)(FC_MAXHB + 3) * FC_ROWS * sizeof(float);

#define FC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#ifdef LTRX_FC_STAMP        // lab builds only (tools/lab/lib_variant.sh): cycle stamps of workgroup LTRX_FC_STAMP, [wave][slate][phase]
__device__ unsigned long long g_fc_stamps[12][10][10];
#define FC_STAMP(it, ph)                                                                                              \
  do {                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    if (blockIdx.x == LTRX_FC_STAMP && (threadIdx.x & 63) == 0 && (it) < 10)                                           \
      g_fc_stamps[threadIdx.x >> 6][it][ph] = __builtin_readcyclecounter();                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  } while (0)
#else
#define FC_STAMP(it, ph)
#endif

// byte offset of 16-byte chunk `col8` (features 8 col8 .. +7) of row `row` inside a plane
__device__ __forceinline__ int plane_off(int row, int col8) {
  if (col8 < 16) {
    const int s = ((row >> 1) & 3) << 1;
    return (col8 >> 3) * FC_PANEL + row * 128 + (((col8 & 7) ^ s) << 4);
  }
  return 2 * FC_PANEL + row * 32 + ((col8 - 16) << 4);
}

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (__bf16)x[e];
    l[e] = (__bf16)(x[e] - (float)h[e]);
  }
}

// barrier that orders LDS only: __syncthreads() also drains vmcnt(0), i.e. it would wait for the next slate's prefetch
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// sum over the 16 lanes of a DPP row (every lane of the row gets the total; fixed order)
__device__ __forceinline__ float row16_sum(float v) {
  v += LTRX_DPP_F(0.f, v, 0xB1, 0xF, true);      // quad_perm [1,0,3,2]
  v += LTRX_DPP_F(0.f, v, 0x4E, 0xF, true);      // quad_perm [2,3,0,1]
  v += LTRX_DPP_F(0.f, v, 0x141, 0xF, true);     // row_half_mirror
  v += LTRX_DPP_F(0.f, v, 0x140, 0xF, true);     // row_mirror
  return v;
}

// ListNet (allrank/models/losses/listNet.py:8-30) of ONE slate held one row per lane by waves 0-3 (row r = threadIdx.x < 256): value
// (added to loss_acc on wave 0) and d loss / d score of this lane's row.  Every wave of the workgroup must call (two LDS barriers
// inside).  Softmax statistics are combined across the four waves as (max, sum of exp(. - max)) pairs: S = sum_w S_w exp(m_w - M) --
// the factor is exactly 1 for the wave that holds the maximum; P = exp(s - M) / S as the reference computes it.  red: 24 floats of LDS.
__device__ __forceinline__ float listnet_rows(float s, float yy, bool in_range, int wave, int lane, float* red, float pad, float eps,
                                              float inv_div, float& loss_acc) {
  float sv = -INFINITY, yv = -INFINITY, P = 0.f, T = 0.f, rr = 0.f;
  bool val = false;
  if (wave < 4) {
    val = in_range && (yy != pad);
    sv = val ? s : -INFINITY;
    yv = val ? yy : -INFINITY;
    const float mw = wave_max(sv), myw = wave_max(yv);
    const float e = val ? expf(sv - mw) : 0.f, f = val ? expf(yv - myw) : 0.f;
    const float sw = wave_sum(e), syw = wave_sum(f);
    if (lane == 0) *reinterpret_cast<f32x4*>(&red[4 * wave]) = f32x4{mw, sw, myw, syw};
  }
  lds_barrier();
  if (wave < 4) {
    f32x4 st[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) st[w] = *reinterpret_cast<const f32x4*>(&red[4 * w]);
    const float M = fmaxf(fmaxf(st[0][0], st[1][0]), fmaxf(st[2][0], st[3][0]));
    const float MY = fmaxf(fmaxf(st[0][2], st[1][2]), fmaxf(st[2][2], st[3][2]));
    float S = 0.f, SY = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (st[w][1] > 0.f) S += st[w][1] * expf(st[w][0] - M);
      if (st[w][3] > 0.f) SY += st[w][3] * expf(st[w][2] - MY);
    }
    const float inv_s = S > 0.f ? 1.0f / S : 0.f;            // fully padded slate: contributes 0 (reference: NaN)
    const float inv_y = SY > 0.f ? 1.0f / SY : 0.f;
    P = val ? expf(sv - M) * inv_s : 0.f;
    T = val ? expf(yv - MY) * inv_y : 0.f;
    rr = (P > 0.f) ? P / (P + eps) : 0.f;
    const float lt = (T > 0.f) ? T * logf(P + eps) : 0.f;
    const float lw = wave_sum(lt), rw = wave_sum(T * rr);
    if (lane == 0) {
      red[16 + 2 * wave] = lw;
      red[17 + 2 * wave] = rw;
    }
  }
  lds_barrier();
  float g = 0.f;
  if (wave < 4) {
    const float R = (red[17] + red[19]) + (red[21] + red[23]);
    g = (P * R - T * rr) * inv_div;                          // padded / beyond L: P = T = 0 -> exactly 0
    if (wave == 0) loss_acc -= (red[16] + red[18]) + (red[20] + red[22]);
  }
  return g;
}

struct FcArgs {
  const float* x;         // [B, L, F]
  const float* y;         // [B, L]
  const float* w1;        // [H, F]
  const float* b1;        // [H]
  const float* wout;      // [H]
  const float* bout;      // [1]
  float* scores;          // [B, L]
  float* dscores;         // [B, L] or null
  float* hidden;          // [B, L, H] or null (tests: the activations whose sign pattern the backward used)
  float* slab;            // [gridDim.x][stride]
  float* step_count;      // Adam's device-side step counter (bumped by workgroup 0) or null
  int B, L, F, H;
  int off_b1, off_wout, off_bout, nflat, stride;
  float eps, pad, inv_div;
};

// FULL: F in (128, 144] -- every K-step / feature block exists, the loops carry no bounds checks (straight-line code the
// compiler can software-pipeline); otherwise the same loops skip the steps beyond F at run time.
template <bool RELU, bool FULL>
__global__ void __launch_bounds__(768) ltrx_fc_listnet_kernel(const FcArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* hi = smem;
  unsigned char* lo = smem + FC_PLANE;
  float* s_part = reinterpret_cast<float*>(smem + 2 * FC_PLANE);      // [FC_MAXHB][256]
  float* ybuf = s_part + FC_MAXHB * FC_ROWS;
  float* dbuf = ybuf + FC_ROWS;                                        // d loss / d score of the slate (0 beyond L / padded)
  float* misc = dbuf + FC_ROWS;                                        // cross-wave softmax statistics
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n16 = lane & 15, kg = lane >> 4;
  const int nthr = blockDim.x, nhb = nthr >> 7;
  const int hb = wave >> 1, rh = wave & 1;
  const int L = a.L, F = a.F, H = a.H;
  const int nks = (F + 31) >> 5, nfb = (F + 15) >> 4;

  if (a.step_count && blockIdx.x == 0 && tid == 0) a.step_count[0] += 1.0f;   // (the reduce + Adam launch reads it)
  // the first slate's first half is requested before anything else (it travels while the image is being zeroed)
  float4 va[FC_NPRE];
  {
    const float4* x4 = reinterpret_cast<const float4*>(a.x + (size_t)blockIdx.x * L * F);
    const int t4 = L * (F >> 2);
#pragma unroll
    for (int j = 0; j < FC_NPRE; ++j) {
      const int q = tid + nthr * j;
      va[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < t4) va[j] = x4[q];
    }
  }

  // ---- once per workgroup: zero the image (rows >= L and columns >= F stay zero for the whole kernel) ----
  {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int o = tid * 16; o < 2 * FC_PLANE; o += nthr * 16) *reinterpret_cast<f32x4*>(smem + o) = z;
    for (int i = tid; i < (FC_MAXHB + 3) * FC_ROWS; i += nthr) s_part[i] = 0.f;
  }
  // ---- this wave's rows of W1 (lane (n16, kg): hidden unit 16 hb + n16, features 32 ks + 8 kg .. +7), its bias and output weight.
  //      The fragments are re-read (L2) and re-split for every slate instead of living in 40 registers for the whole kernel: with
  //      them resident the weight-gradient accumulators spilled (3 of 9 tiles through scratch per K-step) ----
  const int hrow = hb * 16 + n16;
  const bool hok = hrow < H;
  const float* wrow = a.w1 + (size_t)(hok ? hrow : 0) * F;
  const float b1n = hok ? a.b1[hrow] : 0.f;
  const float won = hok ? a.wout[hrow] : 0.f;
  const float bout = a.bout[0];

  f32x4 out[FC_NFB];                       // dW1^T tiles: out[fb][i] = d W1[hrow][16 fb + 4 kg + i], summed over this wave's rows
#pragma unroll
  for (int fb = 0; fb < FC_NFB; ++fb) out[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dwo = 0.f, db1 = 0.f;              // d w_out[hrow], d b1[hrow] (this lane's rows)
  float dbo = 0.f, loss_acc = 0.f;         // waves 0-3: d b_out over their rows; wave 0: sum of the per-slate losses

  const int nf4 = F >> 2, total4 = L * nf4;
  const float inv_nf4 = 1.0f / (float)nf4;
  // Lane-constant parts of the image addresses: the swizzle s(r) = 2 ((r >> 1) & 3) only depends on row bits 1..2, which are lane
  // bits for every access below (tiles start at multiples of 16 rows), so an address is ONE of a few lane bases + an immediate.
  //   forward A fragment, row 128 rh + 32 tp + n16 (+16), chunk 4 ks + kg:  fa[ks & 1] + tp * 4096 (+2048) + (ks >> 1) * FC_PANEL
  //   (tail chunk 16 + kg, kg < 2):                                          ft + tp * 1024 (+512)
  const int fs = ((n16 >> 1) & 3) << 1;
  const int frow = 128 * rh + n16;
  int fa[2];
  fa[0] = frow * 128 + ((kg ^ fs) << 4);
  fa[1] = frow * 128 + (((4 + kg) ^ fs) << 4);
  const int ft = 2 * FC_PANEL + frow * 32 + ((kg & 1) << 4);
  //   transposed read, row 128 rh + 32 kp + 4 kg + rsub (+16), chunk 2 fb + (piece >> 1), 8-byte piece (piece & 1):
  //                                                                          ba[fb & 3] + kp * 4096 (+2048) + (fb >> 2) * FC_PANEL
  //   (tail, fb = 8):                                                        bt + kp * 1024 (+512)
  const int rsub = n16 >> 2, piece = n16 & 3;
  const int brow = 128 * rh + 4 * kg + rsub;
  const int bsv = 2 * (kg & 1) + (rsub >> 1);                 // s(row) >> 1
  int ba[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ba[q] = brow * 128 + ((2 * (q ^ bsv) + (piece >> 1)) << 4) + (piece & 1) * 8;
  const int bt = 2 * FC_PANEL + brow * 32 + ((piece >> 1) << 4) + (piece & 1) * 8;

  // staging of one float4 (4 consecutive features of one row): split, two 8-byte LDS stores
  auto stage_at = [&](int o, const float4& p) {
    bf16x4 h_, l_;
    const float xv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h_[e] = (__bf16)xv[e];
      l_[e] = (__bf16)(xv[e] - (float)h_[e]);
    }
    *reinterpret_cast<bf16x4*>(hi + o) = h_;
    *reinterpret_cast<bf16x4*>(lo + o) = l_;
  };
  auto stage_off = [&](int q) {
    const int row = (int)(((float)q + 0.5f) * inv_nf4);        // == q / nf4 (exact for q < 2^16; checked offline)
    const int c4 = q - row * nf4;
    return plane_off(row, c4 >> 1) + (c4 & 1) * 8;
  };
  // a thread stages the same image positions for every slate: the FC_NLD offsets (/ 8, 16 bits each) live packed in 6 registers
  // instead of being re-derived (a float multiply, a swizzle, a panel select: ~18 VALU per float4, ~2 k cycles per SIMD and slate)
  unsigned so[FC_NLD / 2];
#pragma unroll
  for (int j = 0; j < FC_NLD; j += 2) {
    const int q0 = tid + nthr * j, q1 = tid + nthr * (j + 1);
    const unsigned o0 = q0 < total4 ? (unsigned)stage_off(q0) >> 3 : 0u, o1 = q1 < total4 ? (unsigned)stage_off(q1) >> 3 : 0u;
    so[j >> 1] = o0 | (o1 << 16);
  }
  auto stage4 = [&](int q, const float4& p) {
    const int row = (int)(((float)q + 0.5f) * inv_nf4);
    const int c4 = q - row * nf4;
    bf16x4 h_, l_;
    const float xv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h_[e] = (__bf16)xv[e];
      l_[e] = (__bf16)(xv[e] - (float)h_[e]);
    }
    const int o = plane_off(row, c4 >> 1) + (c4 & 1) * 8;
    *reinterpret_cast<bf16x4*>(hi + o) = h_;
    *reinterpret_cast<bf16x4*>(lo + o) = l_;
  };
  // The first FC_NPRE float4s per thread of a slate are requested one slate AHEAD -- after the forward MFMAs of the slate before, so
  // that they travel during its ListNet and weight-gradient phases (the barriers in between order LDS only, they do not drain the
  // vector-memory counter); the rest is requested at the top of the slate's own iteration.
  auto prefetch = [&](int bb, int tid_, int total4_) {
    const float4* x4 = reinterpret_cast<const float4*>(a.x + (size_t)bb * total4_ * 4);
#pragma unroll
    for (int j = 0; j < FC_NPRE; ++j) {
      const int q = tid_ + nthr * j;
      va[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bb < a.B && q < total4_) va[j] = x4[q];
    }
  };
  __syncthreads();

  int it_ = 0;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x, ++it_) {
    FC_STAMP(it_, 0);
    // hipcc hoists every address / predicate that does not depend on the slate out of this loop and then spills what it hoisted
    // (150 VGPRs, 126 SGPRs): opaque re-definitions keep those one-instruction values inside the loop
    int L_ = L, total4_ = total4;
    asm volatile("" : "+s"(L_), "+s"(total4_));
    asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(ba[0]), "+v"(ba[1]), "+v"(ba[2]), "+v"(ba[3]));
    int ft_ = ft, bt_ = bt, tid_ = tid;
    asm volatile("" : "+v"(ft_), "+v"(bt_), "+v"(tid_));
    bf16x8 wh[FC_NKS], wl[FC_NKS];
    {
      // ---- requests: the second half of the slate, its labels; while they travel the first half (requested a slate ago) is split and
      //      stored; the W1 fragments are requested (L2) into the registers that half has left; then the second half is stored ----
      const float4* x4 = reinterpret_cast<const float4*>(a.x + (size_t)b * L_ * F);
      float4 vb[FC_NLD - FC_NPRE];
#pragma unroll
      for (int j = FC_NPRE; j < FC_NLD; ++j) {
        const int q = tid_ + nthr * j;
        vb[j - FC_NPRE] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < total4_) vb[j - FC_NPRE] = x4[q];
      }
      for (int i = tid_; i < FC_ROWS; i += nthr) ybuf[i] = (i < L_) ? a.y[(size_t)b * L_ + i] : a.pad;
#pragma unroll
      for (int j = 0; j < FC_NPRE; ++j) {
        const int q = tid_ + nthr * j;
        if (q < total4_) stage_at((int)(((so[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) << 3), va[j]);
      }
      float4 wr[2 * FC_NKS];
#pragma unroll
      for (int ks = 0; ks < FC_NKS; ++ks) {
        const int c0 = 32 * ks + 8 * kg;
        wr[2 * ks] = make_float4(0.f, 0.f, 0.f, 0.f);
        wr[2 * ks + 1] = wr[2 * ks];
        if (hok && c0 < F) wr[2 * ks] = *reinterpret_cast<const float4*>(wrow + c0);
        if (hok && c0 + 4 < F) wr[2 * ks + 1] = *reinterpret_cast<const float4*>(wrow + c0 + 4);
      }
#pragma unroll
      for (int j = FC_NPRE; j < FC_NLD; ++j) {
        const int q = tid_ + nthr * j;
        if (q < total4_) stage_at((int)(((so[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) << 3), vb[j - FC_NPRE]);
      }
#pragma unroll
      for (int ks = 0; ks < FC_NKS; ++ks) {
        const float v[8] = {wr[2 * ks].x, wr[2 * ks].y, wr[2 * ks].z, wr[2 * ks].w,
                            wr[2 * ks + 1].x, wr[2 * ks + 1].y, wr[2 * ks + 1].z, wr[2 * ks + 1].w};
        split8(v, wh[ks], wl[ks]);
      }
      // slates longer than FC_NLD * blockDim float4s (small workgroups): the rest in a plain loop
      for (int q = tid_ + nthr * FC_NLD; q < total4_; q += nthr) stage4(q, x4[q]);
    }
    FC_STAMP(it_, 1);
    lds_barrier();
    FC_STAMP(it_, 2);

    // ---- forward: h[t] = act(x[rows of tile t] W1[hrow]^T + b1)   (D layout: lane (n16, kg) holds rows 4 kg + i of the tile) ----
    f32x4 h[8];
    // A fragments of K-step ks of tile pair tp (rows 32 tp + n16 and + 16 of this wave's half): hi / lo of both tiles
    auto load_frag = [&](int tp, int ks, bf16x8 (&fr)[4]) {
      if (ks < 4) {
        const int o = fa[ks & 1] + tp * 4096 + (ks >> 1) * FC_PANEL;
        fr[0] = *reinterpret_cast<const bf16x8*>(hi + o);
        fr[1] = *reinterpret_cast<const bf16x8*>(lo + o);
        fr[2] = *reinterpret_cast<const bf16x8*>(hi + o + 2048);
        fr[3] = *reinterpret_cast<const bf16x8*>(lo + o + 2048);
      } else {                                              // features 128 .. 143 live in the tail panel; 144 .. 159 do not exist
        const int o = ft_ + tp * 1024;
        fr[0] = *reinterpret_cast<const bf16x8*>(hi + o);
        fr[1] = *reinterpret_cast<const bf16x8*>(lo + o);
        fr[2] = *reinterpret_cast<const bf16x8*>(hi + o + 512);
        fr[3] = *reinterpret_cast<const bf16x8*>(lo + o + 512);
        if (kg >= 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) fr[q][e] = (__bf16)0.f;
        }
      }
    };
    {
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (FULL || 128 * rh + 32 * tp < L_) {             // (wave-uniform) tiles entirely beyond the slate: h = act(b1), unused
#pragma unroll
          for (int ks = 0; ks < FC_NKS; ++ks) {
            if (FULL || ks < nks) {
              bf16x8 fr[4];
              load_frag(tp, ks, fr);
              acc0 = FC_MFMA(fr[1], wh[ks], acc0);
              acc1 = FC_MFMA(fr[3], wh[ks], acc1);
              acc0 = FC_MFMA(fr[0], wl[ks], acc0);
              acc1 = FC_MFMA(fr[2], wl[ks], acc1);
              acc0 = FC_MFMA(fr[0], wh[ks], acc0);
              acc1 = FC_MFMA(fr[2], wh[ks], acc1);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float u = acc0[i] + b1n, w = acc1[i] + b1n;
          if (RELU) {
            u = fmaxf(u, 0.f);
            w = fmaxf(w, 0.f);
          }
          h[2 * tp][i] = u;
          h[2 * tp + 1][i] = w;
        }
      }
    }
    prefetch(b + gridDim.x, tid_, total4_);                 // (in flight until the top of the next iteration)
    // score partials: sum over this wave's 16 hidden units of h * w_out, per row
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x4 p;
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = row16_sum(h[t][i] * won);
      if (n16 == 0) *reinterpret_cast<f32x4*>(&s_part[hb * FC_ROWS + 128 * rh + 16 * t + 4 * kg]) = p;
    }
    if (a.hidden) {                                               // (tests only)
      int rb_ = 128 * rh + 4 * kg;
      asm volatile("" : "+v"(rb_));
      float* hp = a.hidden + (size_t)b * L_ * H + hrow;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = rb_ + 16 * t + i;
          if (r < L_ && hok) hp[r * H] = h[t][i];
        }
    }
    FC_STAMP(it_, 3);
    lds_barrier();
    FC_STAMP(it_, 4);

    // ---- scores + ListNet (listNet.py:8-30) by waves 0-3, one row per lane ----
    {
      float sc_ = 0.f, yy_ = a.pad;
      const int r = tid_;
      if (wave < 4) {
        sc_ = bout;
        for (int q = 0; q < nhb; ++q) sc_ += s_part[q * FC_ROWS + r];
        yy_ = ybuf[r];
        if (r < L_) a.scores[(size_t)b * L_ + r] = sc_;
      }
      const float g = listnet_rows(sc_, yy_, r < L_, wave, lane, misc, a.pad, a.eps, a.inv_div, loss_acc);
      if (wave < 4) {
        dbuf[r] = g;
        if (a.dscores && r < L_) a.dscores[(size_t)b * L_ + r] = g;
        dbo += g;
      }
    }
    FC_STAMP(it_, 5);
    lds_barrier();
    FC_STAMP(it_, 6);

    // ---- backward: dh = dscore (x) w_out (* relu'), d w_out += dscore h, d b1 += dh, dW1^T += x^T dh on the matrix cores ----
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
      const int R0 = 128 * rh + 32 * kp;
      if (R0 < L_) {                                              // (wave-uniform)
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(&dbuf[R0 + 4 * kg]);
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(&dbuf[R0 + 16 + 4 * kg]);
        float dh[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float h0 = h[2 * kp][i], h1 = h[2 * kp + 1][i];
          float g0 = d0[i] * won, g1 = d1[i] * won;
          if (RELU) {
            g0 = h0 > 0.f ? g0 : 0.f;
            g1 = h1 > 0.f ? g1 : 0.f;
          }
          dwo += d0[i] * h0;
          dwo += d1[i] * h1;
          db1 += g0;
          db1 += g1;
          dh[i] = g0;
          dh[4 + i] = g1;
        }
        bf16x8 dhh, dhl;
        split8(dh, dhh, dhl);
        // lane i16 of a 16-lane group addresses row (i16 >> 2), 8-byte piece (i16 & 3) of a [4 rows][16 cols] block and receives
        // column i16 of it: rows R + 4 kg + 0..3 of feature 16 fb + n16
#pragma unroll
        for (int fb = 0; fb < FC_NFB; ++fb) {
          if (FULL || fb < nfb) {
            const int o0 = (fb < 8) ? ba[fb & 3] + kp * 4096 + (fb >> 2) * FC_PANEL : bt_ + kp * 1024;
            const int o1 = o0 + ((fb < 8) ? 2048 : 512);
            const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(hi + o0));
            const bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(hi + o1));
            const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(lo + o0));
            const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(lo + o1));
            const bf16x8 xh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            const bf16x8 xl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            out[fb] = FC_MFMA(xl, dhh, out[fb]);
            out[fb] = FC_MFMA(xh, dhl, out[fb]);
            out[fb] = FC_MFMA(xh, dhh, out[fb]);
          }
        }
      }
    }
    FC_STAMP(it_, 7);
    lds_barrier();                                                // the image is free for the next slate
    FC_STAMP(it_, 8);
  }

  // ---- one partial gradient per workgroup: the two row halves are added through LDS (the image is dead), fixed order ----
  float* slab = a.slab + (size_t)blockIdx.x * a.stride;
  dwo += __shfl_xor(dwo, 16, 64);
  dwo += __shfl_xor(dwo, 32, 64);
  db1 += __shfl_xor(db1, 16, 64);
  db1 += __shfl_xor(db1, 32, 64);
  f32x4* sc = reinterpret_cast<f32x4*>(smem);                     // [hb][fb][lane]
  float* sc2 = reinterpret_cast<float*>(smem + (size_t)FC_MAXHB * FC_NFB * 64 * sizeof(f32x4));   // [hb][2][16]
  if (wave < 4) {
    const float t = wave_sum(dbo);
    if (lane == 0) misc[32 + wave] = t;
  }
  if (rh == 1) {
#pragma unroll
    for (int fb = 0; fb < FC_NFB; ++fb) sc[(hb * FC_NFB + fb) * 64 + lane] = out[fb];
    if (kg == 0) {
      sc2[(hb * 2 + 0) * 16 + n16] = dwo;
      sc2[(hb * 2 + 1) * 16 + n16] = db1;
    }
  }
  __syncthreads();
  if (rh == 0) {
    const int H4 = (H + 3) & ~3;
#pragma unroll
    for (int fb = 0; fb < FC_NFB; ++fb) {
      const f32x4 o = out[fb] + sc[(hb * FC_NFB + fb) * 64 + lane];
      const int c = 16 * fb + 4 * kg;
      if (hok && c < F) *reinterpret_cast<f32x4*>(slab + (size_t)hrow * F + c) = o;
    }
    if (kg == 0 && hrow < H4) {
      slab[a.off_wout + hrow] = hok ? dwo + sc2[(hb * 2 + 0) * 16 + n16] : 0.f;
      slab[a.off_b1 + hrow] = hok ? db1 + sc2[(hb * 2 + 1) * 16 + n16] : 0.f;
    }
  }
  if (wave == 0 && lane < 4) slab[a.off_bout + lane] = (lane == 0) ? (misc[32] + misc[33]) + (misc[34] + misc[35]) : 0.f;
  if (wave == 0 && lane >= 4 && lane < 8) slab[a.nflat + lane - 4] = (lane == 4) ? loss_acc : 0.f;
}

// Fixed-order sum of the workgroup partials, float4 column c4 of the flat layout: 16 groups of partials in flight per column,
// combined through LDS in one order.  Writes the gradient, the loss (slot nflat), and optionally applies Adam / AdamW
// (ltrx_adam_kernel's update, same expression order) in the same pass.
struct FcAdam {
  float* p;
  float* m;
  float* v;
  const float* step;
  float lr, b1, b2, eps, wd;
  int decoupled;
};
__global__ void __launch_bounds__(1024) ltrx_fc_reduce_kernel(const float* __restrict__ slab, int nwg, int stride, int nflat,
                                                               float* __restrict__ grads, float* __restrict__ loss_out,
                                                               float inv_div, const FcAdam ad) {
  // 16 float4 columns x 64 groups of partials per workgroup (~200 workgroups at H x F = 96 x 136): every thread has its
  // nwg / 64 loads in flight at once; the 64 group sums are combined in two fixed-order levels through LDS
  __shared__ f32x4 part[64][16];
  __shared__ f32x4 part2[8][16];
  const int col = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int c4 = blockIdx.x * 16 + col, n4 = (nflat >> 2) + 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c4 < n4)
    for (int g = gq; g < nwg; g += 64) acc += *reinterpret_cast<const f32x4*>(slab + (size_t)g * stride + 4 * c4);
  part[gq][col] = acc;
  __syncthreads();
  if (gq < 8) {
    f32x4 u = part[8 * gq][col];
#pragma unroll
    for (int q = 1; q < 8; ++q) u += part[8 * gq + q][col];
    part2[gq][col] = u;
  }
  __syncthreads();
  if (gq != 0 || c4 >= n4) return;
  f32x4 t = part2[0][col];
#pragma unroll
  for (int q = 1; q < 8; ++q) t += part2[q][col];
  if (4 * c4 >= nflat) {
    loss_out[0] = t[0] * inv_div;
    return;
  }
  *reinterpret_cast<f32x4*>(grads + 4 * c4) = t;
  if (!ad.p) return;
  const float l2 = ad.decoupled ? 0.f : ad.wd, shrink = ad.decoupled ? 1.0f - ad.lr * ad.wd : 1.0f;
  const float ts = ad.step[0];
  const float bc1 = 1.0f - powf(ad.b1, ts);
  const float bc2s = sqrtf(1.0f - powf(ad.b2, ts));
  const float step_size = ad.lr / bc1;
  f32x4 pp = *reinterpret_cast<f32x4*>(ad.p + 4 * c4);
  f32x4 mm = *reinterpret_cast<f32x4*>(ad.m + 4 * c4);
  f32x4 vv = *reinterpret_cast<f32x4*>(ad.v + 4 * c4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gr = t[e] * 1.0f + l2 * pp[e];
    mm[e] = ad.b1 * mm[e] + (1.0f - ad.b1) * gr;
    vv[e] = ad.b2 * vv[e] + (1.0f - ad.b2) * gr * gr;
    pp[e] = pp[e] * shrink - step_size * (mm[e] / (sqrtf(vv[e]) / bc2s + ad.eps));
  }
  *reinterpret_cast<f32x4*>(ad.p + 4 * c4) = pp;
  *reinterpret_cast<f32x4*>(ad.m + 4 * c4) = mm;
  *reinterpret_cast<f32x4*>(ad.v + 4 * c4) = vv;
}

// ------------------------------------------------------------------------------------------------------------------
// Linear scorer (FC activation None): score = w_out . (W1 x + b1) + b_out = x . v + c with v = W1^T w_out, c = w_out . b1 + b_out --
// two consecutive linear maps are one, exactly.  The gradients keep that structure:
//   u = sum_{slates, l} dscore_l x_l,  D = sum dscore_l   =>   dW1 = w_out (x) u,  db1 = D w_out,  dw_out = W1 u + D b1,  db_out = D,
// so a slate needs two matrix-VECTOR products (fp32 FMAs, no split, no matrix cores) and a workgroup's partial gradient is F + 2
// floats.  The slate never touches LDS: thread (c4, r0) keeps the float4 column group c4 of rows r0, r0 + RPT, ... (RPT = 768 / (F/4)
// rows per pass) in registers -- 44 VGPRs at 240 x 136 -- next to the SAME registers of the next slate, whose loads are issued before
// the current slate is processed: the kernel streams x at the rate one CU can pull from HBM.  Row sums go through a small LDS table
// in a fixed order (deterministic); ListNet is the four-wave routine of the MFMA kernel.
// This is an opt-in specialisation (FusedTrainer(fc_step="collapse")): the default for an FCModel + listNet job is the MFMA kernel
// above, which evaluates the two layers as the reference does and also covers ReLU.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FL_KMAX = 13;                  // rows per thread at the largest shape (256 rows, 36 column groups, 21 rows per pass)
constexpr int FL_PSTRIDE = 37;               // row stride of the partial-dot table (floats): odd -> conflict-free row sums
constexpr size_t FL_SMEM = (size_t)(FC_ROWS * FL_PSTRIDE + 3 * FC_ROWS + 64 + FC_MAXF) * sizeof(float) + 48 * 36 * sizeof(f32x4);

struct FlArgs {
  const float* x;
  const float* y;
  const float* w1;
  const float* b1;
  const float* wout;
  const float* bout;
  float* scores;
  float* dscores;
  float* slab;          // [gridDim.x][F + 4 + H4]: u[F], D, loss, 0, 0, (W1 u + D b1)[H4]
  float* wout_snapshot; // [H]: w_out as this step read it
  float* step_count;
  int B, L, F, H;
  float eps, pad, inv_div;
};

// KMAX: rows per thread (ceil(L / RPT)): 11 covers the WEB30K shape (240 rows, 34 column groups), FL_KMAX every supported one
template <int KMAX>
__global__ void __launch_bounds__(768) ltrx_fc_linear_listnet_kernel(const FlArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* part = reinterpret_cast<float*>(smem);                 // [256][37] partial dots
  float* ybuf = part + FC_ROWS * FL_PSTRIDE;
  float* dbuf = ybuf + FC_ROWS;
  float* sbuf = dbuf + FC_ROWS;
  float* misc = sbuf + FC_ROWS;                                 // 64 floats: softmax statistics, final combines
  float* vbuf = misc + 64;                                      // [F] v = W1^T w_out
  f32x4* ured = reinterpret_cast<f32x4*>(vbuf + FC_MAXF);       // [rpt][nc4] at the end
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int L = a.L, F = a.F, H = a.H, nc4 = F >> 2;
  const int rpt = 768 / nc4;                                    // rows per pass
  const int c4 = tid % nc4, r0 = tid / nc4;
  const bool active = r0 < rpt;
  if (a.step_count && blockIdx.x == 0 && tid == 0) a.step_count[0] += 1.0f;

  f32x4 xa[KMAX], xb[KMAX];
  float ya = a.pad, yb = a.pad;                                 // label of row `tid` of the slate in xa / xb (requested with it: a load
                                                                // issued later would make its wait cover the whole prefetch, vmcnt is in order)
  auto load = [&](f32x4 (&xr)[KMAX], float& yr, int bb) {
    int r0_ = r0, L_ = L;                                       // (opaque copies: keep the per-row predicates / addresses out of scratch)
    asm volatile("" : "+v"(r0_), "+s"(L_));
    const float* xs = a.x + (size_t)bb * L * F + 4 * c4;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int r = r0_ + rpt * k;
      xr[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (active && bb < a.B && r < L_) xr[k] = *reinterpret_cast<const f32x4*>(xs + (size_t)r * F);
    }
    yr = a.pad;
    if (bb < a.B && tid < L_) yr = a.y[(size_t)bb * L + tid];
  };
  load(xa, ya, blockIdx.x);
  // v = W1^T w_out, c = w_out . b1 + b_out: 768 / F groups of F threads, each group a slice of the hidden units with all its loads in
  // flight at once (one thread per feature looping over H dependent loads cost 40 us per launch), combined in a fixed order
  {
    float* wo_s = part;                                         // [H] w_out, [H] b1 (the partial-dot table is free before the first slate)
    float* b1_s = part + FC_MAXH;
    float* vpart = part + 2 * FC_MAXH;                          // [ngrp][F]
    for (int n = tid; n < H; n += 768) {
      wo_s[n] = a.wout[n];
      b1_s[n] = a.b1[n];
      if (blockIdx.x == 0) a.wout_snapshot[n] = wo_s[n];         // (the finishing launch updates w_out while other workgroups still need it)
    }
    __syncthreads();
    const int ngrp = 768 / F, grp = tid / F, f = tid - grp * F;
    const int chunk = (H + ngrp - 1) / ngrp;
    if (grp < ngrp) {
      float acc = 0.f;
      const int n0 = grp * chunk, n1 = min(H, n0 + chunk);
#pragma unroll 8
      for (int n = n0; n < n1; ++n) acc += wo_s[n] * a.w1[(size_t)n * F + f];
      vpart[grp * F + f] = acc;
    }
    __syncthreads();
    for (int ff = tid; ff < F; ff += 768) {
      float acc = vpart[ff];
      for (int g = 1; g < ngrp; ++g) acc += vpart[g * F + ff];
      vbuf[ff] = acc;
    }
  }
  float cc = a.bout[0];
  for (int n = 0; n < H; ++n) cc += part[n] * part[FC_MAXH + n];        // (LDS broadcast reads: every thread the same sum, same order)
  __syncthreads();
  const f32x4 v4 = active ? *reinterpret_cast<const f32x4*>(vbuf + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 u4 = {0.f, 0.f, 0.f, 0.f};
  float dsum = 0.f, loss_acc = 0.f;

  auto slate = [&](const f32x4 (&xr)[KMAX], float yr, int b) {
    if (tid < FC_ROWS) ybuf[tid] = yr;
    if (active) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const int r = r0 + rpt * k;
        if (r < L) part[r * FL_PSTRIDE + c4] = (xr[k][0] * v4[0] + xr[k][1] * v4[1]) + (xr[k][2] * v4[2] + xr[k][3] * v4[3]);
      }
    }
    lds_barrier();
    float sc = 0.f, yy = a.pad;
    if (wave < 4) {
      const int r = tid;
      sc = cc;
      if (r < L)
        for (int q = 0; q < nc4; ++q) sc += part[r * FL_PSTRIDE + q];
      yy = ybuf[r];
      if (r < L) a.scores[(size_t)b * L + r] = sc;
    }
    const float g = listnet_rows(sc, yy, tid < L, wave, lane, misc, a.pad, a.eps, a.inv_div, loss_acc);
    if (wave < 4) {
      dbuf[tid] = g;
      if (a.dscores && tid < L) a.dscores[(size_t)b * L + tid] = g;
      dsum += g;
    }
    lds_barrier();
    if (active) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const int r = r0 + rpt * k;
        if (r < L) {
          const float d = dbuf[r];
          u4[0] += d * xr[k][0]; u4[1] += d * xr[k][1]; u4[2] += d * xr[k][2]; u4[3] += d * xr[k][3];
        }
      }
    }
    lds_barrier();                                              // part / dbuf / ybuf are free for the next slate
  };

  for (int b = blockIdx.x; b < a.B; b += 2 * gridDim.x) {
    load(xb, yb, b + gridDim.x);                                // next slate in flight while this one is processed
    slate(xa, ya, b);
    if (b + (int)gridDim.x < a.B) {
      load(xa, ya, b + 2 * gridDim.x);
      slate(xb, yb, b + gridDim.x);
    }
  }

  // ---- one partial per workgroup: u summed over the row groups in a fixed order, D, the loss ----
  if (active) ured[r0 * nc4 + c4] = u4;
  if (wave < 4) {
    const float t = wave_sum(dsum);
    if (lane == 0) misc[32 + wave] = t;
  }
  __syncthreads();
  const int H4 = (H + 3) & ~3, sstride = F + 4 + H4;
  float* slab = a.slab + (size_t)blockIdx.x * sstride;
  if (tid < nc4) {
    f32x4 t = ured[tid];
    for (int q = 1; q < rpt; ++q) t += ured[q * nc4 + tid];
    *reinterpret_cast<f32x4*>(slab + 4 * tid) = t;
    *reinterpret_cast<f32x4*>(vbuf + 4 * tid) = t;              // (v is dead: the workgroup's u for the product below)
  }
  const float Dw = (misc[32] + misc[33]) + (misc[34] + misc[35]);
  if (tid == 0) {
    slab[F] = Dw;
    slab[F + 1] = loss_acc;
    slab[F + 2] = 0.f;
    slab[F + 3] = 0.f;
  }
  __syncthreads();
  // this workgroup's share of dw_out = W1 u + D b1 (linear in (u, D), so the shares add up): 8 threads per hidden unit, 1 / 8 of the
  // features each, combined in a fixed order -- the W1 loads of all units are in flight at once
  {
    float* rpart = part;                                        // [8][H]
    const int n = tid % FC_MAXH, sl = tid / FC_MAXH;            // 768 = 8 x 96
    if (n < H) {
      const int fchunk = (F + 7) / 8, f0 = sl * fchunk, f1 = min(F, f0 + fchunk);
      float acc = 0.f;
#pragma unroll 6
      for (int f = f0; f < f1; ++f) acc += a.w1[(size_t)n * F + f] * vbuf[f];
      rpart[sl * FC_MAXH + n] = acc;
    }
    __syncthreads();
    if (tid < H4) {
      float acc = 0.f;
      if (tid < H) {
        acc = rpart[tid];
        for (int q = 1; q < 8; ++q) acc += rpart[q * FC_MAXH + tid];
        acc += a.b1[tid] * Dw;
      }
      slab[F + 4 + tid] = acc;
    }
  }
}

// partials -> (u, D, loss, W1 u + D b1) -> the four gradients of the flat layout (dW1 = w_out (x) u, db1 = D w_out, dw_out, db_out = D).
// Every workgroup re-reduces the (tiny) partials in a fixed order.
__global__ void __launch_bounds__(1024) ltrx_fc_linear_finish_kernel(const float* __restrict__ slab, int nwg, int F, int H, int off_b1,
                                                                      int off_wout, int off_bout, int nflat, float* __restrict__ grads,
                                                                      float* __restrict__ loss_out, float inv_div, const FcAdam ad,
                                                                      const float* __restrict__ wo) {
  // every workgroup re-reduces the (tiny) partials -- nwg x (F + 4 + H4) floats -- in a fixed order, forms the gradients of its
  // 1024 flat entries (w_out from the step's snapshot, never from `params`, which other workgroups are updating) and applies the
  // optimizer to them in the same pass
  __shared__ __attribute__((aligned(16))) float tot[FC_MAXF + 4 + FC_MAXH];
  __shared__ f32x4 psum[16][(FC_MAXF + 4 + FC_MAXH) / 4];
  const int tid = threadIdx.x;
  const int H4 = (H + 3) & ~3, sstride = F + 4 + H4;
  {
    const int nc = sstride >> 2, col = tid % nc, gq = tid / nc;
    const int ng = (1024 / nc) < 16 ? (1024 / nc) : 16;        // complete groups of nc threads (nc <= 61): all loads in flight at once
    if (gq < ng) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int g = gq; g < nwg; g += ng) acc += *reinterpret_cast<const f32x4*>(slab + (size_t)g * sstride + 4 * col);
      psum[gq][col] = acc;
    }
    __syncthreads();
    if (tid < nc) {
      f32x4 t = psum[0][tid];
      for (int q = 1; q < ng; ++q) t += psum[q][tid];          // fixed order
      *reinterpret_cast<f32x4*>(&tot[4 * tid]) = t;
    }
    __syncthreads();
  }
  const float D = tot[F];
  if (blockIdx.x == 0 && tid == 0) loss_out[0] = tot[F + 1] * inv_div;
  const int i = blockIdx.x * 1024 + tid;
  if (i >= nflat) return;
  float g;
  if (i < off_b1) {
    const int n = i / F, f = i - n * F;
    g = wo[n] * tot[f];
  } else if (i < off_wout) {
    const int n = i - off_b1;
    g = (n < H) ? wo[n] * D : 0.f;
  } else if (i < off_bout) {
    const int n = i - off_wout;
    g = (n < H) ? tot[F + 4 + n] : 0.f;
  } else {
    g = (i == off_bout) ? D : 0.f;
  }
  grads[i] = g;
  if (!ad.p) return;
  const float l2 = ad.decoupled ? 0.f : ad.wd, shrink = ad.decoupled ? 1.0f - ad.lr * ad.wd : 1.0f;
  const float ts = ad.step[0];
  const float bc1 = 1.0f - powf(ad.b1, ts);
  const float bc2s = sqrtf(1.0f - powf(ad.b2, ts));
  const float step_size = ad.lr / bc1;
  const float pp = ad.p[i];
  const float gr = g * 1.0f + l2 * pp;
  const float mm = ad.b1 * ad.m[i] + (1.0f - ad.b1) * gr;
  const float vv = ad.b2 * ad.v[i] + (1.0f - ad.b2) * gr * gr;
  ad.m[i] = mm;
  ad.v[i] = vv;
  ad.p[i] = pp * shrink - step_size * (mm / (sqrtf(vv) / bc2s + ad.eps));
}

std::atomic<uint64_t> g_fl_attr{0};
std::atomic<uint64_t> g_fc_attr{0};

int fc_workgroups(int B) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return B < cus ? B : cus;
}

}  // namespace

#ifdef LTRX_FC_STAMP
extern "C" int ltrx_debug_fc_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fc_stamps), sizeof(g_fc_stamps)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int ltrx_fc_listnet_supported(int L, int F, int H) {
  return (L > 0 && L <= FC_ROWS && F > 0 && F <= FC_MAXF && (F & 3) == 0 && H > 0 && H <= FC_MAXH) ? 1 : 0;
}

extern "C" size_t ltrx_fc_listnet_workspace_bytes(int B, int L, int F, int H, size_t nflat) {
  (void)L;
  (void)F;
  (void)H;
  if (B <= 0) return 0;
  int cus = 1024;                                                    // (no device query here: an upper bound on the workgroups)
  const size_t nwg = (size_t)(B < cus ? B : cus);
  return nwg * (nflat + 4) * sizeof(float);
}

extern "C" int ltrx_fc_listnet_step(const float* x, const float* y, int B, int L, int F, int H, int act, float* params,
                                    size_t off_w1, size_t off_b1, size_t off_wout, size_t off_bout, size_t nflat, float eps,
                                    float pad_value, float batch_divisor, float* scores, float* dscores, float* hidden_out,
                                    float* loss_out, float* grads, float* exp_avg, float* exp_avg_sq, float* step_count, float lr,
                                    float beta1, float beta2, float adam_eps, float weight_decay, int decoupled, void* ws,
                                    ltrx_stream_t stream) {
  if (!x || !y || !params || !scores || !loss_out || !grads || !ws || B <= 0 || !(batch_divisor > 0.f)) return LTRX_EINVAL;
  if (act != 0 && act != 1) return LTRX_EINVAL;
  if (!ltrx_fc_listnet_supported(L, F, H)) return LTRX_EUNSUPPORTED;
  // the four tensors must be the adjacent 4-float-aligned segments of one flat buffer (every slot of a partial gradient is written)
  const size_t H4 = ((size_t)H + 3) & ~(size_t)3;
  if (off_w1 != 0 || off_b1 != (size_t)H * F || off_wout != off_b1 + H4 || off_bout != off_wout + H4 || nflat != off_bout + 4)
    return LTRX_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)params | (uintptr_t)grads | (uintptr_t)ws) & 15)) return LTRX_EINVAL;
  if (exp_avg && (!exp_avg_sq || !step_count)) return LTRX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int rc = ltrx_once_per_device(g_fc_attr, []() -> int {
    const void* ks[4] = {(const void*)ltrx_fc_listnet_kernel<false, false>, (const void*)ltrx_fc_listnet_kernel<true, false>,
                         (const void*)ltrx_fc_listnet_kernel<false, true>, (const void*)ltrx_fc_listnet_kernel<true, true>};
    for (int i = 0; i < 4; ++i)
      if (hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)FC_SMEM) != hipSuccess) return LTRX_EHIP;
    return LTRX_OK;
  });
  if (rc != LTRX_OK) return rc;
  FcArgs a;
  a.x = x;
  a.y = y;
  a.w1 = params + off_w1;
  a.b1 = params + off_b1;
  a.wout = params + off_wout;
  a.bout = params + off_bout;
  a.scores = scores;
  a.dscores = dscores;
  a.hidden = hidden_out;
  a.slab = (float*)ws;
  a.step_count = exp_avg ? step_count : nullptr;
  a.B = B;
  a.L = L;
  a.F = F;
  a.H = H;
  a.off_b1 = (int)off_b1;
  a.off_wout = (int)off_wout;
  a.off_bout = (int)off_bout;
  a.nflat = (int)nflat;
  a.stride = (int)nflat + 4;
  a.eps = eps;
  a.pad = pad_value;
  a.inv_div = 1.0f / batch_divisor;
  const int nwg = fc_workgroups(B);
  const int nhb = (H + 15) / 16 < 2 ? 2 : (H + 15) / 16;      // (at least four waves: ListNet runs one slate row per lane on waves 0-3)
  const bool full = F > 128;
  if (act && full)
    hipLaunchKernelGGL((ltrx_fc_listnet_kernel<true, true>), dim3(nwg), dim3(128 * nhb), FC_SMEM, s, a);
  else if (act)
    hipLaunchKernelGGL((ltrx_fc_listnet_kernel<true, false>), dim3(nwg), dim3(128 * nhb), FC_SMEM, s, a);
  else if (full)
    hipLaunchKernelGGL((ltrx_fc_listnet_kernel<false, true>), dim3(nwg), dim3(128 * nhb), FC_SMEM, s, a);
  else
    hipLaunchKernelGGL((ltrx_fc_listnet_kernel<false, false>), dim3(nwg), dim3(128 * nhb), FC_SMEM, s, a);
  LTRX_LAUNCH_CHECK();
  FcAdam ad;
  ad.p = exp_avg ? params : nullptr;
  ad.m = exp_avg;
  ad.v = exp_avg_sq;
  ad.step = step_count;
  ad.lr = lr;
  ad.b1 = beta1;
  ad.b2 = beta2;
  ad.eps = adam_eps;
  ad.wd = weight_decay;
  ad.decoupled = decoupled;
  const int n4 = (int)(nflat >> 2) + 1;
  hipLaunchKernelGGL(ltrx_fc_reduce_kernel, dim3((n4 + 15) / 16), dim3(1024), 0, s, (const float*)ws, nwg, a.stride, (int)nflat, grads,
                     loss_out, 1.0f / batch_divisor, ad);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}

extern "C" size_t ltrx_fc_linear_listnet_workspace_bytes(int B, int F) {
  if (B <= 0 || F <= 0) return 0;
  const size_t nwg = (size_t)(B < 1024 ? B : 1024);
  return (nwg * (size_t)(F + 4 + FC_MAXH) + FC_MAXH) * sizeof(float);
}

extern "C" int ltrx_fc_linear_listnet_step(const float* x, const float* y, int B, int L, int F, int H, float* params, size_t off_w1,
                                           size_t off_b1, size_t off_wout, size_t off_bout, size_t nflat, float eps, float pad_value,
                                           float batch_divisor, float* scores, float* dscores, float* loss_out, float* grads,
                                           float* exp_avg, float* exp_avg_sq, float* step_count, float lr, float beta1, float beta2,
                                           float adam_eps, float weight_decay, int decoupled, void* ws, ltrx_stream_t stream) {
  if (!x || !y || !params || !scores || !loss_out || !grads || !ws || B <= 0 || !(batch_divisor > 0.f)) return LTRX_EINVAL;
  if (!ltrx_fc_listnet_supported(L, F, H)) return LTRX_EUNSUPPORTED;
  const size_t H4 = ((size_t)H + 3) & ~(size_t)3;
  if (off_w1 != 0 || off_b1 != (size_t)H * F || off_wout != off_b1 + H4 || off_bout != off_wout + H4 || nflat != off_bout + 4)
    return LTRX_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)params | (uintptr_t)grads | (uintptr_t)ws) & 15)) return LTRX_EINVAL;
  if (exp_avg && (!exp_avg_sq || !step_count)) return LTRX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int rc = ltrx_once_per_device(g_fl_attr, []() -> int {
    if (hipFuncSetAttribute((const void*)ltrx_fc_linear_listnet_kernel<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL_SMEM) != hipSuccess)
      return LTRX_EHIP;
    return hipFuncSetAttribute((const void*)ltrx_fc_linear_listnet_kernel<FL_KMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL_SMEM) ==
                   hipSuccess ? LTRX_OK : LTRX_EHIP;
  });
  if (rc != LTRX_OK) return rc;
  FlArgs a;
  a.x = x;
  a.y = y;
  a.w1 = params + off_w1;
  a.b1 = params + off_b1;
  a.wout = params + off_wout;
  a.bout = params + off_bout;
  a.scores = scores;
  a.dscores = dscores;
  a.slab = (float*)ws;
  a.wout_snapshot = (float*)ws + (size_t)fc_workgroups(B) * (size_t)(F + 4 + (((size_t)H + 3) & ~(size_t)3));
  a.step_count = exp_avg ? step_count : nullptr;
  a.B = B;
  a.L = L;
  a.F = F;
  a.H = H;
  a.eps = eps;
  a.pad = pad_value;
  a.inv_div = 1.0f / batch_divisor;
  const int nwg = fc_workgroups(B);
  const int rpt = 768 / (F >> 2);
  if ((L + rpt - 1) / rpt <= 11)
    hipLaunchKernelGGL(ltrx_fc_linear_listnet_kernel<11>, dim3(nwg), dim3(768), FL_SMEM, s, a);
  else
    hipLaunchKernelGGL(ltrx_fc_linear_listnet_kernel<FL_KMAX>, dim3(nwg), dim3(768), FL_SMEM, s, a);
  LTRX_LAUNCH_CHECK();
  FcAdam ad;
  ad.p = exp_avg ? params : nullptr;
  ad.m = exp_avg;
  ad.v = exp_avg_sq;
  ad.step = step_count;
  ad.lr = lr;
  ad.b1 = beta1;
  ad.b2 = beta2;
  ad.eps = adam_eps;
  ad.wd = weight_decay;
  ad.decoupled = decoupled;
  hipLaunchKernelGGL(ltrx_fc_linear_finish_kernel, dim3((unsigned)((nflat + 1023) / 1024)), dim3(1024), 0, s, (const float*)ws, nwg, F, H,
                     (int)off_b1, (int)off_wout, (int)off_bout, (int)nflat, grads, loss_out, 1.0f / batch_divisor, ad,
                     (const float*)a.wout_snapshot);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
