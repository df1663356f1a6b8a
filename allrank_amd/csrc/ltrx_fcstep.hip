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
constexpr size_t FC_SMEM = 2 * (size_t)FC_PLANE + (size_t)(FC_MAXHB + 3) * FC_ROWS * sizeof(float);

#define FC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

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

// sum over the 16 lanes of a DPP row (every lane of the row gets the total; fixed order)
__device__ __forceinline__ float row16_sum(float v) {
  v += LTRX_DPP_F(0.f, v, 0xB1, 0xF, true);      // quad_perm [1,0,3,2]
  v += LTRX_DPP_F(0.f, v, 0x4E, 0xF, true);      // quad_perm [2,3,0,1]
  v += LTRX_DPP_F(0.f, v, 0x141, 0xF, true);     // row_half_mirror
  v += LTRX_DPP_F(0.f, v, 0x140, 0xF, true);     // row_mirror
  return v;
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

template <bool RELU>
__global__ void __launch_bounds__(768) ltrx_fc_listnet_kernel(const FcArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* hi = smem;
  unsigned char* lo = smem + FC_PLANE;
  float* s_part = reinterpret_cast<float*>(smem + 2 * FC_PLANE);      // [FC_MAXHB][256]
  float* ybuf = s_part + FC_MAXHB * FC_ROWS;
  float* dbuf = ybuf + FC_ROWS;                                        // d loss / d score of the slate (0 beyond L / padded)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n16 = lane & 15, kg = lane >> 4;
  const int nthr = blockDim.x, nhb = nthr >> 7;
  const int hb = wave >> 1, rh = wave & 1;
  const int L = a.L, F = a.F, H = a.H;
  const int nks = (F + 31) >> 5, nfb = (F + 15) >> 4;

  if (a.step_count && blockIdx.x == 0 && tid == 0) a.step_count[0] += 1.0f;   // (the reduce + Adam launch reads it)

  // ---- once per workgroup: zero the image (rows >= L and columns >= F stay zero for the whole kernel) ----
  {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int o = tid * 16; o < 2 * FC_PLANE; o += nthr * 16) *reinterpret_cast<f32x4*>(smem + o) = z;
    for (int i = tid; i < (FC_MAXHB + 3) * FC_ROWS; i += nthr) s_part[i] = 0.f;
  }
  // ---- once per workgroup: this wave's rows of W1 as pre-split B fragments (lane (n16, kg): hidden unit 16 hb + n16, features
  //      32 ks + 8 kg .. +7), its bias and output weight ----
  const int hrow = hb * 16 + n16;
  const bool hok = hrow < H;
  bf16x8 wh[FC_NKS], wl[FC_NKS];
#pragma unroll
  for (int ks = 0; ks < FC_NKS; ++ks) {
    const int c0 = 32 * ks + 8 * kg;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (hok && c0 < F) {
      const float4 p = *reinterpret_cast<const float4*>(a.w1 + (size_t)hrow * F + c0);
      v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w;
      if (c0 + 4 < F) {
        const float4 q = *reinterpret_cast<const float4*>(a.w1 + (size_t)hrow * F + c0 + 4);
        v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
      }
    }
    split8(v, wh[ks], wl[ks]);
  }
  const float b1n = hok ? a.b1[hrow] : 0.f;
  const float won = hok ? a.wout[hrow] : 0.f;
  const float bout = a.bout[0];

  f32x4 out[FC_NFB];                       // dW1^T tiles: out[fb][i] = d W1[hrow][16 fb + 4 kg + i], summed over this wave's rows
#pragma unroll
  for (int fb = 0; fb < FC_NFB; ++fb) out[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dwo = 0.f, db1 = 0.f;              // d w_out[hrow], d b1[hrow] (this lane's rows)
  float dbo = 0.f, loss_acc = 0.f;         // wave 0: d b_out, sum of the per-slate losses

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
  __syncthreads();

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    // hipcc hoists every address / predicate that does not depend on the slate out of this loop and then spills what it hoisted
    // (150 VGPRs, 126 SGPRs): opaque re-definitions keep those one-instruction values inside the loop
    int L_ = L, total4_ = total4;
    asm volatile("" : "+s"(L_), "+s"(total4_));
    asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(ba[0]), "+v"(ba[1]), "+v"(ba[2]), "+v"(ba[3]));
    int ft_ = ft, bt_ = bt, tid_ = tid;
    asm volatile("" : "+v"(ft_), "+v"(bt_), "+v"(tid_));
    // ---- stage the slate: coalesced float4 loads of the contiguous [L_, F] block, split, 8-byte LDS stores ----
    {
      const float4* x4 = reinterpret_cast<const float4*>(a.x + (size_t)b * L_ * F);
      float4 v[FC_NLD];
#pragma unroll
      for (int j = 0; j < FC_NLD; ++j) {
        const int q = tid_ + nthr * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < total4_) v[j] = x4[q];
      }
      for (int i = tid_; i < FC_ROWS; i += nthr) ybuf[i] = (i < L_) ? a.y[(size_t)b * L_ + i] : a.pad;
#pragma unroll
      for (int j = 0; j < FC_NLD; ++j) {
        const int q = tid_ + nthr * j;
        if (q < total4_) {
          const int row = (int)(((float)q + 0.5f) * inv_nf4);       // == q / nf4 (exact for q < 2^16; checked offline)
          const int c4 = q - row * nf4;
          bf16x4 h, l;
          const float xv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            h[e] = (__bf16)xv[e];
            l[e] = (__bf16)(xv[e] - (float)h[e]);
          }
          const int o = plane_off(row, c4 >> 1) + (c4 & 1) * 8;
          *reinterpret_cast<bf16x4*>(hi + o) = h;
          *reinterpret_cast<bf16x4*>(lo + o) = l;
        }
      }
      // slates longer than FC_NLD * blockDim float4s (small workgroups): the rest in a plain loop
      for (int q = tid_ + nthr * FC_NLD; q < total4_; q += nthr) {
        const float4 p = x4[q];
        const int row = (int)(((float)q + 0.5f) * inv_nf4);
        const int c4 = q - row * nf4;
        bf16x4 h, l;
        const float xv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h[e] = (__bf16)xv[e];
          l[e] = (__bf16)(xv[e] - (float)h[e]);
        }
        const int o = plane_off(row, c4 >> 1) + (c4 & 1) * 8;
        *reinterpret_cast<bf16x4*>(hi + o) = h;
        *reinterpret_cast<bf16x4*>(lo + o) = l;
      }
    }
    __syncthreads();

    // ---- forward: h[t] = act(x[rows of tile t] W1[hrow]^T + b1)   (D layout: lane (n16, kg) holds rows 4 kg + i of the tile) ----
    f32x4 h[8];
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      if (128 * rh + 32 * tp < L_) {                       // (wave-uniform) tiles entirely beyond the slate: h = act(b1), unused
#pragma unroll
        for (int ks = 0; ks < FC_NKS; ++ks) {
          if (ks < nks) {
            bf16x8 ah, al, bh, bl;
            if (ks < 4) {
              const int o = fa[ks & 1] + tp * 4096 + (ks >> 1) * FC_PANEL;
              ah = *reinterpret_cast<const bf16x8*>(hi + o);
              al = *reinterpret_cast<const bf16x8*>(lo + o);
              bh = *reinterpret_cast<const bf16x8*>(hi + o + 2048);
              bl = *reinterpret_cast<const bf16x8*>(lo + o + 2048);
            } else {                                        // features 128 .. 143 live in the tail panel; 144 .. 159 do not exist
              const int o = ft_ + tp * 1024;
              ah = *reinterpret_cast<const bf16x8*>(hi + o);
              al = *reinterpret_cast<const bf16x8*>(lo + o);
              bh = *reinterpret_cast<const bf16x8*>(hi + o + 512);
              bl = *reinterpret_cast<const bf16x8*>(lo + o + 512);
              if (kg >= 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  ah[e] = (__bf16)0.f; al[e] = (__bf16)0.f; bh[e] = (__bf16)0.f; bl[e] = (__bf16)0.f;
                }
              }
            }
            acc0 = FC_MFMA(al, wh[ks], acc0);
            acc1 = FC_MFMA(bl, wh[ks], acc1);
            acc0 = FC_MFMA(ah, wl[ks], acc0);
            acc1 = FC_MFMA(bh, wl[ks], acc1);
            acc0 = FC_MFMA(ah, wh[ks], acc0);
            acc1 = FC_MFMA(bh, wh[ks], acc1);
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
    __syncthreads();

    // ---- scores + ListNet (listNet.py:8-30) by wave 0: four rows per lane ----
    if (wave == 0) {
      float sv[4], yv[4];
      bool val[4];
      float smax = -INFINITY, ymax = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = lane + 64 * j;
        float s = bout;
        for (int q = 0; q < nhb; ++q) s += s_part[q * FC_ROWS + r];
        const float yy = ybuf[r];
        val[j] = (r < L_) && (yy != a.pad);
        if (r < L_) a.scores[(size_t)b * L_ + r] = s;
        sv[j] = val[j] ? s : -INFINITY;
        yv[j] = val[j] ? yy : -INFINITY;
        smax = fmaxf(smax, sv[j]);
        ymax = fmaxf(ymax, yv[j]);
      }
      smax = wave_max(smax);
      ymax = wave_max(ymax);
      float e[4], f[4], ssum = 0.f, ysum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e[j] = val[j] ? expf(sv[j] - smax) : 0.f;
        f[j] = val[j] ? expf(yv[j] - ymax) : 0.f;
        ssum += e[j];
        ysum += f[j];
      }
      ssum = wave_sum(ssum);
      ysum = wave_sum(ysum);
      const float inv_s = ssum > 0.f ? 1.0f / ssum : 0.f;        // fully padded slate: contributes 0 (reference: NaN)
      const float inv_y = ysum > 0.f ? 1.0f / ysum : 0.f;
      float lsum = 0.f, rsum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e[j] *= inv_s;                                            // P
        f[j] *= inv_y;                                            // T
        if (f[j] > 0.f) lsum += f[j] * logf(e[j] + a.eps);
        rsum += f[j] * (e[j] / (e[j] + a.eps));
      }
      lsum = wave_sum(lsum);
      rsum = wave_sum(rsum);
      loss_acc += -lsum;
      float gs = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = lane + 64 * j;
        const float P = e[j], T = f[j];
        const float rr = (P > 0.f) ? P / (P + a.eps) : 0.f;
        const float g = (P * rsum - T * rr) * a.inv_div;          // padded / beyond L_: P = T = 0 -> exactly 0
        dbuf[r] = g;
        gs += g;
        if (a.dscores && r < L_) a.dscores[(size_t)b * L_ + r] = g;
      }
      dbo += wave_sum(gs);
    }
    __syncthreads();

    // ---- backward: dh = dscore (x) w_out (* relu'), d w_out += dscore h, d b1 += dh, dW1^T += x^T dh on the matrix cores ----
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
      const int R0 = 128 * rh + 32 * kp;
      if (R0 < L_) {                                               // (wave-uniform)
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
          if (fb < nfb) {
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
    __syncthreads();                                              // the image is free for the next slate
  }

  // ---- one partial gradient per workgroup: the two row halves are added through LDS (the image is dead), fixed order ----
  float* slab = a.slab + (size_t)blockIdx.x * a.stride;
  dwo += __shfl_xor(dwo, 16, 64);
  dwo += __shfl_xor(dwo, 32, 64);
  db1 += __shfl_xor(db1, 16, 64);
  db1 += __shfl_xor(db1, 32, 64);
  f32x4* sc = reinterpret_cast<f32x4*>(smem);                     // [hb][fb][lane]
  float* sc2 = reinterpret_cast<float*>(smem + (size_t)FC_MAXHB * FC_NFB * 64 * sizeof(f32x4));   // [hb][2][16]
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
  if (wave == 0 && lane < 4) slab[a.off_bout + lane] = (lane == 0) ? dbo : 0.f;
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
  __shared__ f32x4 part[16][64];
  const int col = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + col, n4 = (nflat >> 2) + 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c4 < n4)
    for (int g = gq; g < nwg; g += 16) acc += *reinterpret_cast<const f32x4*>(slab + (size_t)g * stride + 4 * c4);
  part[gq][col] = acc;
  __syncthreads();
  if (gq != 0 || c4 >= n4) return;
  f32x4 t = part[0][col];
#pragma unroll
  for (int q = 1; q < 16; ++q) t += part[q][col];
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
    if (hipFuncSetAttribute((const void*)ltrx_fc_listnet_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FC_SMEM) != hipSuccess)
      return LTRX_EHIP;
    if (hipFuncSetAttribute((const void*)ltrx_fc_listnet_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FC_SMEM) != hipSuccess)
      return LTRX_EHIP;
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
  const int nhb = (H + 15) / 16;
  if (act)
    hipLaunchKernelGGL(ltrx_fc_listnet_kernel<true>, dim3(nwg), dim3(128 * nhb), FC_SMEM, s, a);
  else
    hipLaunchKernelGGL(ltrx_fc_listnet_kernel<false>, dim3(nwg), dim3(128 * nhb), FC_SMEM, s, a);
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
  hipLaunchKernelGGL(ltrx_fc_reduce_kernel, dim3((n4 + 63) / 64), dim3(1024), 0, s, (const float*)ws, nwg, a.stride, (int)nflat, grads,
                     loss_out, 1.0f / batch_divisor, ad);
  LTRX_LAUNCH_CHECK();
  return LTRX_OK;
}
