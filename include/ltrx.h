/*
 * ltrx.h -- C ABI of libltrx.so: MI355X (gfx950) kernels for the listwise-LTR training hot path.
 *
 * This is the drop-in boundary of the engine.  allegro/allRank has no
 * FFI of its own: its plugin surface is Python name lookup (allrank/main.py:75,82-83).  Each entry
 * point below therefore cites the reference *Python* function it replaces; the Python side
 * (allrank_amd/) binds them with ctypes and mirrors the reference signatures one-to-one.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, scalars; no torch / HIP types in the signatures.
 *     `ltrx_stream_t` is a hipStream_t passed as void* (NULL = the null stream).
 *   - every tensor is caller-owned, dense row-major fp32 unless stated; kernels never allocate, never
 *     retain pointers, never synchronise the device, and enqueue on the given stream only.
 *   - y_true uses `pad_value` (-1 in allRank, allrank/data/dataset_loading.py:15) to mark padded slots.
 *   - re-entrant: no entry point reads or writes process-global mode switches; arithmetic / path / tile choices are
 *     arguments of the call that uses them.  (The only state kept is a per-device "kernel attributes already set" bit.)
 *   - return value: 0 = ok; LTRX_EINVAL bad argument; LTRX_EUNSUPPORTED shape outside the supported
 *     range (slate length above LTRX_MAX_SLATE_LEN for a loss, LTRX_MAX_METRIC_SLATE_LEN for a metric -- the
 *     Python wrappers name the limit in the exception; there is no silent fallback); LTRX_EHIP a HIP launch error (hipGetLastError code is returned as -(1000+code)).
 *   - `batch_divisor`: the number of slates the reference would have averaged over.  On one GPU it is B;
 *     under slate sharding it is the GLOBAL batch so that summing per-rank results reproduces the
 *     reference's loss on the gathered batch.
 *   - sort tie policy: stable descending (lower original index first), SURVEY.md §9.2.
 */
#ifndef LTRX_H
#define LTRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTRX_VERSION 130 /* 0.3.0 (round 6): ltrx_first_nonfinite; ltrx_fixlength_positions keys its draw by the slate id (not the batch row) */

#define LTRX_OK 0
#define LTRX_EINVAL (-1)
#define LTRX_EUNSUPPORTED (-2)
#define LTRX_EHIP (-1000)

#define LTRX_MAX_SLATE_LEN 2048        /* the loss kernels' tuned form: a slate and its per-item work arrays in LDS.  Every loss takes slates
                                          up to here; the four hot losses (listNet, listMLE, approxNDCG, lambdaLoss) go on: */
#define LTRX_MAX_LONG_SLATE_LEN 16384  /* ... their work arrays stay in LDS while they fit the CU's 160 KB (listNet: every length, listMLE
                                          6.7 k, approxNDCG 5.8 k, lambdaLoss 3.1 k items) and move to the call's workspace beyond that
                                          (same kernels through a global pointer; *_workspace_bytes(B, L) grows accordingly) -- the
                                          reference pads a validation set to its longest slate, allrank/data/dataset_loading.py:185-194 */
#define LTRX_MHA_DS_BUDGET_BYTES (2147483648ull) /* ltrx_mha_bwd, modes 1 / 2: the dS hand-over workspace is B h LK^2 floats (LK = L rounded
                                                    up to 64); a call that would need more than this runs the exact-fp32 kernels
                                                    (workspace B L h floats) instead -- ltrx_mha_bwd_workspace_bytes says which */
#define LTRX_MAX_METRIC_SLATE_LEN 8192 /* ltrx_ndcg_at / ltrx_mrr_at: validation sets are padded to their longest slate
                                          (allrank/data/dataset_loading.py:185-194); 16 B per item of the 160 KB LDS */

typedef void* ltrx_stream_t;

int ltrx_version(void);

/* ---------------------------------------------------------------------------------------------
 * Listwise losses.  All of them: inputs y_pred[B,L], y_true[B,L]; outputs loss_out[1] (scalar, as the
 * reference returns), optional per_slate_out[B] (the per-slate term before the batch reduction),
 * optional grad_out[B,L] = d loss / d y_pred (exactly 0 at padded slots).  `ws` is a scratch buffer
 * of at least the matching *_workspace_bytes(); it is write-only scratch, contents undefined after.
 * ------------------------------------------------------------------------------------------- */

/* allrank/models/losses/listNet.py:8-30   listNet(y_pred, y_true, eps, padded_value_indicator) */
size_t ltrx_listnet_workspace_bytes(int B, int L);
int ltrx_listnet_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps, float pad_value,
                         float batch_divisor, float* loss_out, float* per_slate_out, float* grad_out, void* ws,
                         ltrx_stream_t stream);

/* allrank/models/losses/listMLE.py:7-38   listMLE(y_pred, y_true, eps, padded_value_indicator)
 * `perm[L]` (int64, device) is the column shuffle the reference draws with torch.randperm (listMLE.py:17);
 * order_out[B,L] (int64, optional) receives the original item index at each sorted position. */
size_t ltrx_listmle_workspace_bytes(int B, int L);
int ltrx_listmle_fwd_bwd(const float* y_pred, const float* y_true, const int64_t* perm, int B, int L, float eps,
                         float pad_value, float batch_divisor, float* loss_out, float* per_slate_out,
                         float* grad_out, int64_t* order_out, void* ws, ltrx_stream_t stream);

/* allrank/models/losses/approxNDCG.py:7-53   approxNDCGLoss(y_pred, y_true, eps, padded_value_indicator, alpha) */
size_t ltrx_approxndcg_workspace_bytes(int B, int L);
int ltrx_approxndcg_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps, float pad_value,
                            float alpha, float batch_divisor, float* loss_out, float* per_slate_out,
                            float* grad_out, void* ws, ltrx_stream_t stream);

/* allrank/models/losses/lambdaLoss.py:7-114   lambdaLoss(y_pred, y_true, eps, padded_value_indicator,
 *                                              weighing_scheme, k, sigma, mu, reduction, reduction_log) */
enum ltrx_lambda_scheme {
  LTRX_SCHEME_NONE = 0,                 /* weighing_scheme=None            (lambdaLoss.py:58-59)  */
  LTRX_SCHEME_NDCGLOSS1 = 1,            /* ndcgLoss1_scheme                (:84-85)               */
  LTRX_SCHEME_NDCGLOSS2 = 2,            /* ndcgLoss2_scheme                (:88-94)               */
  LTRX_SCHEME_LAMBDARANK = 3,           /* lambdaRank_scheme               (:97-98)               */
  LTRX_SCHEME_NDCGLOSS2PP = 4,          /* ndcgLoss2PP_scheme              (:101-102)             */
  LTRX_SCHEME_RANKNET = 5,              /* rankNet_scheme                  (:105-106)             */
  LTRX_SCHEME_RANKNET_GTDIFF = 6,       /* rankNetWeightedByGTDiff_scheme  (:109-110)             */
  LTRX_SCHEME_RANKNET_GTDIFF_POWED = 7  /* rankNetWeightedByGTDiffPowed_scheme (:113-114)         */
};
enum ltrx_reduction { LTRX_REDUCE_SUM = 0, LTRX_REDUCE_MEAN = 1 };
enum ltrx_logbase { LTRX_LOG_BINARY = 0, LTRX_LOG_NATURAL = 1 };
/* k <= 0 means k=None (no truncation).  For LTRX_REDUCE_MEAN the divisor is the number of selected pairs
 * in THIS call's batch; pair_count_out[1] (optional) receives it.  When `ext_pair_count` (device, optional)
 * is given it is used as the divisor instead (global count under slate sharding). */
size_t ltrx_lambdaloss_workspace_bytes(int B, int L);
int ltrx_lambdaloss_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps, float pad_value,
                            int scheme, int k, float sigma, float mu, int reduction, int logbase,
                            const float* ext_pair_count, float* loss_out, float* pair_count_out, float* grad_out,
                            int64_t* order_out, void* ws, ltrx_stream_t stream);

/* allrank/models/losses/neuralNDCG.py:10-70 (neuralNDCG) and :73-136 (neuralNDCG_transposed), deterministic
 * NeuralSort (loss_utils.py:34-67) + Sinkhorn scaling (loss_utils.py:8-31).
 *   step 1  ltrx_neuralndcg_prepare: per-slate ideal DCG@k (metrics.py:41-77 on (y_true,y_true)) and
 *           nonzero_count_out[1] = number of slates with idcg != 0 (the loss normaliser, neuralNDCG.py:69).
 *           Under slate sharding the caller all-reduces nonzero_count between the two steps.
 *   step 2  ltrx_neuralndcg_fwd_bwd: loss and gradient; `nonzero_count` is read from DEVICE memory.
 * transposed != 0 selects the neuralNDCG_transposed conventions (same value; idcg stays "powered" when
 * powered_relevancies == 0, neuralNDCG.py:126).  k <= 0 means k=None.  The Sinkhorn early exit
 * (loss_utils.py:25) is batch-global as in the reference; iters_out[1] (int32, optional) = iterations used.
 * k_rows[B] (int32, device, optional): a per-slate cap on the ranks that carry a discount (min(k, k_rows[b])) -- the
 * stochastic variant (loss_utils.py:84-112) masks the permutation rows by the TRUE slate's padding (neuralNDCG.py:44)
 * while sorting each perturbed copy under another slate's mask (mask.repeat_interleave, :36/:41).
 * path: 0 = automatic (register-resident Sinkhorn kernels when L <= 240), 1 = always the general L2-streaming kernels (same
 * results; a per-call argument so that tests can pin either path without any process state). */
size_t ltrx_neuralndcg_workspace_bytes(int B, int L, int max_iter);
int ltrx_neuralndcg_prepare(const float* y_true, int B, int L, float pad_value, int k, int idcg_powered,
                            float* idcg_out, float* nonzero_count_out, void* ws, ltrx_stream_t stream);
int ltrx_neuralndcg_fwd_bwd(const float* y_pred, const float* y_true, const float* idcg, const float* nonzero_count,
                            int B, int L, float pad_value, float temperature, int powered_relevancies, int k,
                            const int32_t* k_rows, int transposed, int max_iter, float tol, float* loss_out, float* per_slate_out,
                            float* grad_out, int32_t* iters_out, int path, void* ws, ltrx_stream_t stream);

/* allrank/models/metrics.py:7-77   ndcg(y_pred, y_true, ats, gain=2^x-1, padding_indicator, filler_value)
 * ats[n_ats] is a HOST array.  ndcg_out[B,n_ats]; dcg_out[B,n_ats] optional; order_out[B,L] (int64, optional)
 * = stable descending argsort of the masked predictions (padded slots last, in original order). */
size_t ltrx_ndcg_workspace_bytes(int B, int L);
int ltrx_ndcg_at(const float* y_pred, const float* y_true, int B, int L, const int* ats, int n_ats,
                 float pad_value, float filler_value, float* ndcg_out, float* dcg_out, int64_t* order_out,
                 void* ws, ltrx_stream_t stream);
/* the same with a caller-supplied gain_function (metrics.py:7-8,41-42,67): gains[B,L] = gain_function evaluated per item on the
 * masked labels (padded items -> label 0 first, metrics.py:35; they then carry gain_function(0) at their tail positions).  Both
 * rankings -- by prediction and the ideal one, which sorts by LABEL (metrics.py:21) -- still come from y_pred / y_true. */
int ltrx_ndcg_at_gains(const float* y_pred, const float* y_true, const float* gains, int B, int L, const int* ats, int n_ats,
                       float pad_value, float filler_value, float* ndcg_out, float* dcg_out, int64_t* order_out,
                       void* ws, ltrx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Pointwise / pairwise members of allrank.models.losses and MRR.
 * Count-normalised losses follow the lambdaLoss protocol: *_count_out[1] (optional) receives this call's
 * count, ext_*count (device, optional) replaces it as the divisor (global count under slate sharding).
 * ------------------------------------------------------------------------------------------- */

/* allrank/models/losses/rankNet.py:8-79  rankNet / rankNet_weightByGTDiff / rankNet_weightByGTDiff_pow:
 * BCEWithLogits(target 1) over the pairs y_i > y_j of valid items, mean over ALL pairs of the batch.
 * weight_mode 0: unweighted, 1: |y_i - y_j|, 2: |y_i^2 - y_j^2|.  No pair at all -> loss NaN, gradient 0 (torch). */
size_t ltrx_ranknet_workspace_bytes(int B, int L);
int ltrx_ranknet_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float pad_value, int weight_mode,
                         const float* ext_pair_count, float* loss_out, float* pair_count_out, float* grad_out, void* ws,
                         ltrx_stream_t stream);

/* allrank/models/losses/bce.py:8-32 (n == 0: y_pred[B,L] probabilities, divisor = slates with a valid item) and
 * ordinal.py:8-50 (n >= 1: y_pred[B,L,n], targets [y_true >= k+1], divisor = valid items); torch.nn.BCELoss arithmetic.
 * grad_out has the shape of y_pred. */
size_t ltrx_bce_workspace_bytes(int B, int L, int n);
int ltrx_bce_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, int n, float pad_value, const float* ext_count,
                     float* loss_out, float* count_out, float* grad_out, void* ws, ltrx_stream_t stream);

/* allrank/models/losses/pointwise.py:6-32  pointwise_rmse(y_pred, y_true, no_of_levels) */
size_t ltrx_pointwise_rmse_workspace_bytes(int B, int L);
int ltrx_pointwise_rmse_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float no_of_levels, float pad_value,
                                float batch_divisor, float* loss_out, float* grad_out, void* ws, ltrx_stream_t stream);

/* allrank/models/losses/binary_listNet.py:8-33 */
size_t ltrx_binary_listnet_workspace_bytes(int B, int L);
int ltrx_binary_listnet_fwd_bwd(const float* y_pred, const float* y_true, int B, int L, float eps, float pad_value,
                                float batch_divisor, float* loss_out, float* grad_out, void* ws, ltrx_stream_t stream);

/* allrank/models/metrics.py:80-113  mrr(y_pred, y_true, ats, padding_indicator): mrr_out[B,n_ats]; ats is a HOST array.
 * Ties in the predictions are ranked by the stable descending order (padded slots last). */
size_t ltrx_mrr_workspace_bytes(int B, int L, int n_ats);
int ltrx_mrr_at(const float* y_pred, const float* y_true, int B, int L, const int* ats, int n_ats, float pad_value,
                float* mrr_out, void* ws, ltrx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Scoring model kernels (allrank/models/transformer.py).
 * ------------------------------------------------------------------------------------------- */

/* transformer.py:59-81  custom LayerNorm: y = a*(x-mean)/(std_unbiased+eps)+b over the last dim D, with an
 * optional fused residual input (transformer.py:105: the sum x + sublayer(...) that feeds the next norm):
 *   xsum = x (+ res);  y = LN(xsum).   xsum_out may be NULL when res is NULL.
 * Saves mean[rows], rstd[rows] (= 1/(std+eps)) for the backward.  res_drop_p > 0 applies the SublayerConnection dropout
 * (transformer.py:105) to the residual branch inside the add: xsum = x + drop(res), counter-based mask keyed by
 * drop_seed ^ hash(drop_step[0]) (drop_step may be NULL); the backward of the branch is ltrx_dropout_apply. */
int ltrx_layernorm_fwd(const float* x, const float* res, const float* a, const float* b, int rows, int D, float eps,
                       float* xsum_out, float* y_out, float* mean_out, float* rstd_out, float res_drop_p, uint32_t drop_seed,
                       const uint32_t* drop_step, ltrx_stream_t stream);
/* the same with y written as a pre-split operand image (ltrx_split_image's layout; for an output that only feeds GEMMs, see
 * ltrx_gemm_nt_img).  D must be 256, 512, 768 or 1024 and the buffers 16-byte aligned (LTRX_EUNSUPPORTED otherwise). */
int ltrx_layernorm_fwd_image(const float* x, const float* res, const float* a, const float* b, int rows, int D, float eps,
                             float* xsum_out, void* y_image_out, float* mean_out, float* rstd_out, float res_drop_p, uint32_t drop_seed,
                             const uint32_t* drop_step, ltrx_stream_t stream);
/* dx = LN backward of dy (+ dres_in if given: the gradient arriving through the residual branch);
 * da_out/db_out (fp32, length D) are written via a deterministic two-stage reduction through ws. */
size_t ltrx_layernorm_bwd_workspace_bytes(int rows, int D);
int ltrx_layernorm_bwd(const float* dy, const float* xsum, const float* a, const float* mean, const float* rstd,
                       const float* dres_in, int rows, int D, float eps, float* dx_out, float* da_out, float* db_out,
                       void* ws, ltrx_stream_t stream);
/* ltrx_layernorm_bwd without the parameter-gradient reduction: dx_out is final; ws holds *partial_rows_out rows of [da(D) | db(D)]
 * partials (row stride 2 D) that the caller sums, e.g. as two entries of ltrx_reduce_group. */
int ltrx_layernorm_bwd_partial(const float* dy, const float* xsum, const float* a, const float* mean, const float* rstd,
                               const float* dres_in, int rows, int D, float eps, float* dx_out, void* ws, int* partial_rows_out,
                               ltrx_stream_t stream);

/* transformer.py:137-156 attention() as used by MultiHeadedAttention.forward (:178-203), fused flash-style:
 * q,k,v,o are [B, L, h, d_k] views of the projection outputs (element (b,l,head,c) at ((b*L+l)*h+head)*d_k + c
 * scaled by the given row stride), key_pad_mask u8[B,L] (1 = padded key, filled with -inf, transformer.py:150-151);
 * softmax over keys; dropout on the probabilities (transformer.py:154-155) with rate p_drop (0 = off) from a counter-
 * based generator keyed by `seed` (the backward must be given the same p_drop and seed; the stream differs from torch's
 * Philox, equivalence is statistical).  lse_out[B,h,L] = log-sum-exp of the
 * scaled, masked scores (the only tensor saved for backward).  fp32 in/out; contractions on the fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), exact fp32 products.  d_k % 4 == 0, d_k <= 128 (zero-padded to a multiple of 32).
 * cu_seqlens (i32[B+1] in device memory, or NULL): variable-length layout -- slate b occupies rows cu[b] .. cu[b+1]-1 of
 * q/k/v/o (its valid items packed, ltrx_gather_rows), L is then only the maximum slate length (grid size and the stride of
 * lse / delta); key_pad_mask may be NULL in that layout (every packed row is a valid key).
 * slate_order (i32[B] in device memory, or NULL): a permutation of the slates giving the order in which their workgroups are
 * launched -- longest first balances the CUs on ragged batches; results do not depend on it. */
/* `mode` = arithmetic of the attention contractions OF THIS CALL (the library keeps no mode; the backward must be given the
 * mode of its forward): 1 = split-bf16 on the bf16 MFMA (3 products per fp32 product, fp32-class,
 * like the dense projections) with the whole slate resident in LDS, used wherever the shape fits (slate length <= 256,
 * 32 < d_k <= 64; dropout and variable-length batches included); 0 = exact fp32 MFMA (bit-exact fp32 products) for every
 * shape -- the strict reference; 2 = the kernels of mode 1 with ONE bf16 product per contraction (plain-bf16 throughput mode,
 * about 2^-9 relative error per product: outside the parity contract, reported separately by bench.py).  Shapes that do not
 * fit always run the exact kernels. */
#define LTRX_MHA_EXACT_FP32 0
#define LTRX_MHA_SPLIT_BF16 1
#define LTRX_MHA_PLAIN_BF16 2
int ltrx_mha_fwd(const float* q, const float* k, const float* v, const uint8_t* key_pad_mask, int B, int L, int h,
                 int d_k, int row_stride, float* o, int o_row_stride, float* lse_out, float p_drop, uint32_t seed,
                 const uint32_t* seed_step, const int32_t* cu_seqlens, const int32_t* slate_order, int mode,
                 ltrx_stream_t stream);
/* backward: dq,dk,dv from do.  `ws` holds delta[B,h,L] (rowsum(do*o)) for the exact kernels, or -- in modes 1 / 2 where the
 * LDS-resident kernels run (32 < d_k <= 64, L <= LTRX_MAX_SLATE_LEN) -- the dS exchange between the two backward kernels,
 * B*h*LK*LK floats with LK = L rounded up to 64: S, P, dP and delta are computed once, by the dK/dV kernel, which hands dS to a
 * dQ = dS K kernel through HBM (deterministic, no atomics).  Size it with ltrx_mha_bwd_workspace_bytes for the SAME (d_k, mode)
 * the call will use. */
size_t ltrx_mha_bwd_workspace_bytes(int B, int L, int h, int d_k, int mode);
int ltrx_mha_bwd(const float* q, const float* k, const float* v, const uint8_t* key_pad_mask, const float* o,
                 const float* dout, const float* lse, int B, int L, int h, int d_k, int row_stride, int o_row_stride,
                 float* dq, float* dk, float* dv, int d_row_stride, float p_drop, uint32_t seed,
                 const uint32_t* seed_step, const int32_t* cu_seqlens, const int32_t* slate_order, int mode, void* ws,
                 ltrx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training-step glue (allrank/training/train_utils.py:18-29 around the model): the pieces between the library
 * GEMMs that the explicit, graph-capturable step of allrank_amd/engine.py needs.
 * ------------------------------------------------------------------------------------------- */

/* The optimizers `getattr(torch.optim, config.optimizer.name)` (allrank/main.py:82) resolves to in practice, over one flat fp32
 * buffer of n elements.  ltrx_adam_step = torch.optim.Adam.step (Adam in every shipped config; weight_decay = the L2 term added to
 * the gradient) or, with decoupled != 0, torch.optim.AdamW.step (p *= 1 - lr * weight_decay first): step_count[1] (device, float) is
 * incremented first and used for the bias corrections.  ltrx_sgd_step = torch.optim.SGD.step with dampening 0 (momentum_buf may be
 * NULL when momentum == 0).  Gradients are multiplied by grad_scale (1.0 normally) and, when given, by grad_scale_dev[0] (device; the
 * clipping coefficient) on the fly.  (amsgrad / maximize / foreach variants: not implemented -- the Python layer keeps those jobs
 * on torch's optimizer.) */
int ltrx_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int decoupled, float* step_count, float grad_scale,
                   const float* grad_scale_dev, ltrx_stream_t stream);
int ltrx_sgd_step(float* params, const float* grads, float* momentum_buf, size_t n, float lr, float momentum, int nesterov,
                  float weight_decay, float grad_scale, const float* grad_scale_dev, ltrx_stream_t stream);

/* torch.nn.utils.clip_grad_norm_ (allrank/training/train_utils.py:24-25) over the flat gradient buffer:
 * scale_out[0] = min(1, max_norm / (||grads||_2 + 1e-6)), norm_out[0] (optional) = the norm; deterministic two-stage sum. */
size_t ltrx_clip_workspace_bytes(size_t n);
int ltrx_clip_grad_norm_scale(const float* grads, size_t n, float max_norm, float* scale_out, float* norm_out, void* ws,
                              ltrx_stream_t stream);

/* nn.Linear bias gradient: out[n] (+)= sum_m a[m*ld + n] for a row-major [M,N] matrix; deterministic two-stage. */
size_t ltrx_colsum_workspace_bytes(int M, int N);
int ltrx_colsum(const float* a, int M, int N, int ld, float* out, int accumulate, void* ws, ltrx_stream_t stream);

/* ReLU backward in place (transformer.py:227, FCModel activation): dr[i] = r[i] > 0 ? dr[i] * scale : 0; n % 4 == 0. */
int ltrx_relu_bwd(float* dr_inout, const float* r_post_act, size_t n, float scale, ltrx_stream_t stream);

/* dropout plumbing of the explicit step: dst[i] = src[i] * keep_scale(i) with the same counter-based mask the forward
 * kernels use (seed ^ hash(step[0])); ltrx_bump_u32 advances the per-step word (so a replayed hipGraph re-keys). */
int ltrx_dropout_apply(const float* src, float* dst, size_t n, float p, uint32_t seed, const uint32_t* drop_step,
                       ltrx_stream_t stream);
int ltrx_bump_u32(uint32_t* word, ltrx_stream_t stream);

/* Finiteness check of the explicit step -- the replacement of torch.autograd.detect_anomaly() (allrank/main.py:89,
 * config.detect_anomaly, config.py:77) for a step that has no autograd graph: scans buf[0..n) (the flat gradient buffer; 16-byte
 * aligned) once and writes out[0] = index of the first segment that holds a NaN / Inf (segment s = [seg_start[s], seg_start[s+1]),
 * seg_start sorted ascending, seg_start[0] = 0: the parameter tensors in flat-buffer order), 0x7fffffff if every element is finite;
 * out[1] = number of non-finite elements.  HBM-bound, 4 B per element.  allrank_amd/csrc/ltrx_train.hip. */
int ltrx_first_nonfinite(const float* buf, size_t n, const int64_t* seg_start, int n_seg, int* out, ltrx_stream_t stream);

/* all transposed weight copies of the explicit step in one launch: matrix m = rows x cols floats at src_base + desc[4m],
 * written transposed (cols x rows) at dst_base + desc[4m+1]; desc[4m+2..3] = rows, cols; tile_start[n+1] = prefix sums of
 * ceil(rows/32)*ceil(cols/32) (all arrays in DEVICE memory except the scalars). */
int ltrx_transpose_batch(const float* src_base, float* dst_base, const int64_t* desc, const int32_t* tile_start, int n,
                         int total_tiles, ltrx_stream_t stream);

/* Compacted (variable-length) batches: pack the valid items of a padded [B, L] batch (dataset.py:28-38 pads with -1 rows)
 * into consecutive rows and back.  gather: dst[i,:] = src[idx[i],:] for i < n, zero rows for n <= i < n_pad;
 * scatter: dst[idx[i],:] = src[i,:] for i < n.  idx in DEVICE memory. */
int ltrx_gather_rows(const float* src, int ld_src, const int32_t* idx, int n, int n_pad, int cols, float* dst, int ld_dst,
                     ltrx_stream_t stream);
/* idx[r] = b*L + (r - cu[b]) for packed row r of slate b (valid items first in every slate); n = cu[B]. */
int ltrx_packed_row_index(const int32_t* cu_seqlens, int B, int L, int n, int32_t* idx, ltrx_stream_t stream);
int ltrx_scatter_rows(const float* src, int ld_src, const int32_t* idx, int n, int cols, float* dst, int ld_dst,
                      ltrx_stream_t stream);

/* y = act(y + bias) in place over a contiguous [M,N] matrix (model.py:42-43); act 0 = identity, 1 = ReLU; N % 4 == 0. */
int ltrx_bias_act(float* y_inout, const float* bias, int M, int N, int act, ltrx_stream_t stream);

/* OutputLayer with d_output == 1 (model.py:111-117): scores[m] = <x[m,:], w> + b, and its backward
 * (dx[m,:] = dscores[m] * w; dw = sum_m dscores[m] x[m,:]; db = sum_m dscores[m]). */
int ltrx_score_head_fwd(const float* x, const float* w, const float* b, int M, int D, float* scores, ltrx_stream_t stream);
size_t ltrx_score_head_bwd_workspace_bytes(int M, int D);
int ltrx_score_head_bwd(const float* dscores, const float* x, const float* w, int M, int D, float* dx, float* dw,
                        float* db, void* ws, ltrx_stream_t stream);

/* fp32-accurate dense projections on the bf16 matrix cores ("split-bf16": each fp32 operand is split into bf16 hi + lo
 * while staged into LDS; A B^T ~= Ahi Bhi^T + Ahi Blo^T + Alo Bhi^T with fp32 accumulation).  `strict` is the precision
 * code of a call: 0 = the three products above (parity arithmetic, default); 1 = a 3-term split and 6 products (true-fp32
 * error); 2 = ONE product Ahi Bhi^T (plain bf16 operands, fp32 accumulate: the throughput mode, about 2^-9 relative error
 * per product -- outside the parity contract, reported separately by bench.py).  Replaces the nn.Linear GEMMs of model.py:35-44 and transformer.py:193-203,221-227.
 *   ltrx_gemm_nt: C[M,N] (ld ldc) = epi( A[M,K] (ld lda) * B[N,K]^T (ld ldb) + bias[N] )
 *                 -- forward (B = weight) and input gradient (B = weight^T);  K, lda, ldb multiples of 4.
 *                 epilogue `act`: 0 none, 1 ReLU (transformer.py:227 fused), 2 multiply by (aux[m,n] > 0): the ReLU
 *                 backward fused into the input-gradient GEMM (aux = the saved post-activation tensor, ld ldaux);
 *                 3 add aux[m,n] AFTER the dropout: C = aux + drop_p(A B^T + bias), the residual connection of
 *                 SublayerConnection (transformer.py:98-106) written by the projection that closes the sublayer.
 *                 4 / 5: the ReLU and its backward with the mask carried as ONE BIT per element: act 4 = act 1 that also writes
 *                 the mask (of the post-dropout activation) to aux, act 5 = act 2 reading that mask instead of the fp32 activation
 *                 (aux = a 16-byte aligned buffer of ltrx_gemm_nt_relu_bits_bytes(M, N, K) bytes, ldaux ignored; both launches must
 *                 have the same M and N, and the function must be non-zero for the K of EACH launch: N % 256 == 0, K % 32 == 0 and a
 *                 tile count the large-tile kernel takes; LTRX_EUNSUPPORTED where it returns 0 -- use acts 1 / 2 there).  Same results as acts 1 / 2.
 *                 drop_p > 0: nn.Dropout after the activation (model.py:43, transformer.py:227) fused in the epilogue
 *                 (act 0/1: counter-based mask over the [M,N] output; act 2: the mask is carried by aux, only 1/(1-p)).
 *   ltrx_gemm_tn: C[NP,KP] (dense) = A[M,NP]^T * B[M,KP]  -- weight gradient dW = dY^T X (split over M, deterministic);
 *                 bias_out[NP] (optional) = column sums of A = the bias gradient, produced in the same pass. */
size_t ltrx_gemm_nt_relu_bits_bytes(int M, int N, int K);
int ltrx_gemm_nt(const float* A, int lda, const float* B, int ldb, const void* B_image, float* C, int ldc, int M, int N, int K, const float* bias,
                 int act, const float* aux, int ldaux, float drop_p, uint32_t drop_seed, const uint32_t* drop_step, int strict,
                 int tile, ltrx_stream_t stream);
/* B_image (optional, NULL = none): the operand B pre-split by ltrx_split_image -- every 4 consecutive floats of B replaced, at the
 * same address offset, by the 16 bytes {hi0..hi3, lo0..lo3} of their bf16 hi / lo split.  The large-tile kernels then copy B into LDS
 * instead of splitting it on the fly (identical bits, identical results, less VALU work); other kernels ignore it and read B.  What it
 * is for: nn.Linear weights (model.py:35-44, transformer.py:193-227) change once per optimizer step but are staged by every tile of
 * every GEMM of the step.  ltrx_split_image: n floats (multiple of 4), src and dst 16-byte aligned. */
int ltrx_split_image(const float* src, void* dst, size_t n, ltrx_stream_t stream);
/* Round 5: ACTIVATIONS as operand images.  An activation that only ever feeds GEMMs -- the LayerNorm output in front of the q/k/v and
 * feed-forward projections (transformer.py:105, 193-196, 227), the post-ReLU feed-forward activation (transformer.py:227) -- is
 * written by its producer directly in ltrx_split_image's layout (same bytes, same addresses, no fp32 copy), and its consumers stage it
 * with plain copies: ltrx_gemm_nt_img = ltrx_gemm_nt + operand_flags (LTRX_GEMM_A_IS_IMAGE: A holds an image -- requires B_image;
 * LTRX_GEMM_C_AS_IMAGE: C is written as an image, epilogue applied first); ltrx_layernorm_fwd_image (below); ltrx_gemm_tn_group_img
 * (b_is_image[p]: operand B of problem p is an image).  Results are bit-identical to the fp32 hand-over (the split is the same
 * expression, evaluated once by the producer instead of once per consuming tile).  Images exist in the large-tile kernel families
 * only: ltrx_gemm_nt_image_ok(M, N, K) says whether ltrx_gemm_nt_img will take them for a shape (ask for every consumer BEFORE producing
 * an image; a call that cannot returns LTRX_EUNSUPPORTED); ltrx_gemm_tn_group_img takes them whenever the grouped kernel runs
 * (ltrx_debug_tn_group_map(...) > 0 and the workspace fits). */
#define LTRX_GEMM_A_IS_IMAGE 1
#define LTRX_GEMM_C_AS_IMAGE 2
int ltrx_gemm_nt_image_ok(int M, int N, int K);
int ltrx_gemm_nt_img(const float* A, int lda, const float* B, int ldb, const void* B_image, float* C, int ldc, int M, int N, int K,
                     const float* bias, int act, const float* aux, int ldaux, float drop_p, uint32_t drop_seed, const uint32_t* drop_step,
                     int strict, int tile, int operand_flags, ltrx_stream_t stream);
/* The engine's whole per-step weight refresh in one launch: the image of the flat parameter buffer (src_base[0..nflat) -> src_image,
 * as ltrx_split_image) and, for the n matrices of desc / tile_start (as ltrx_transpose_batch), the transposed fp32 copy in dst_base
 * AND its image in dst_image (same offsets).  Requires nflat % 4 == 0, 16-byte aligned buffers, 4-float aligned dst offsets and
 * rows % 4 == 0 for every matrix.  Bit-identical to ltrx_transpose_batch + 2 x ltrx_split_image.
 * ENGINE-INTERNAL: desc / tile_start are DEVICE tables, so the call can check the host-visible preconditions only (pointers, nflat % 4,
 * 16-byte alignment of the five buffers); the per-matrix ones -- rows % 4 == 0, dst offset % 4 == 0 -- are the caller's contract
 * (allrank_amd/engine.py `_fused_images` verifies them when it builds the table, and uses ltrx_transpose_batch + ltrx_split_image
 * otherwise).  A table that violates them yields float4 stores that straddle rows: silent corruption of the copies, never a fault
 * outside the buffers.  Other callers should use the two public calls it fuses. */
int ltrx_weight_images(const float* src_base, size_t nflat, void* src_image, float* dst_base, void* dst_image, const int64_t* desc,
                       const int32_t* tile_start, int n, int total_tiles, const float* pad_src, int pad_rows, int pad_cols, int pad_ld,
                       float* pad_dst, void* pad_dst_image, ltrx_stream_t stream);
/*   pad_src (optional, NULL = none): one more matrix [pad_rows][pad_cols] copied to pad_dst[pad_rows][pad_ld] (fp32) and
 *   pad_dst_image (its image) in the same launch -- the engine keeps the first FC weight [H, F] with rows of ld = F rounded up to 32
 *   so that the input projection runs the large-tile GEMM (padding columns: whatever the buffers hold, zeros in the engine). */
/* A batch into the step's static input buffers in one launch: x (nx floats = rows of F) into rows of ld_dst >= F floats of x_dst
 * (padding columns untouched; nx may be 0), y_dst = y (ny floats),
 * mask_dst[i] = (y[i] == pad_value) -- the padding mask the reference derives per batch (allrank/training/train_utils.py:19,
 * allrank/data/dataset_loading.py:15 PADDED_Y_VALUE). */
int ltrx_ingest_batch(const float* x, const float* y, size_t nx, size_t ny, int F, int ld_dst, float pad_value, float* x_dst,
                      float* y_dst, unsigned char* mask_dst, ltrx_stream_t stream);
/* `tile` (both GEMMs) is a per-call tuning argument: 0 = automatic choice per shape (what every product call passes);
 * ltrx_gemm_nt: 1 128x128x32, 2 128x128x64, 3 256x128x32, 4 256x128x64, 6 256x256x32, 7 128x256x32 (the large-tile forms
 * need N % 256 == 0, K % 32 == 0); ltrx_gemm_tn: 1 = the 128x128 kernel even where the 256x256 one applies; 9 = the caller states that
 * every row of B (the last one included) is READABLE up to column KP rounded up to 256 (ldb >= that): the 256x256 kernel then also
 * serves a KP that is no multiple of 256 -- it reads the padding, drops its products, C stays dense [NP, KP] (the engine's input
 * buffer keeps F = 136 features in rows of 256 floats).  8 (ltrx_gemm_nt) = 64x256x32 tiles, two workgroups per CU.  Results do not
 * depend on the tile beyond fp32 summation order. */
/* workspace for ltrx_gemm_tn sized for M rows: sufficient for EVERY call with the same NP, KP and any row count <= M
 * (variable-length batches re-use one workspace); ltrx_gemm_tn_splits = the split count a call with exactly M rows uses
 * (each split owns an [NP,KP] slab + 2 bias rows of the workspace). */
size_t ltrx_gemm_tn_workspace_bytes(int M, int NP, int KP);
int ltrx_gemm_tn_splits(int M, int NP, int KP);
int ltrx_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, float* bias_out, int M, int NP, int KP, int strict,
                 int tile, void* ws, ltrx_stream_t stream);
/* Up to LTRX_GEMM_TN_GROUP_MAX weight gradients over the SAME M rows in one launch (the four nn.Linear projections of an encoder
 * layer: allrank/models/transformer.py:174, 217-218 -- loss.backward() computes each dW = dY^T X separately, allrank/training/
 * train_utils.py:23): C[p][NP[p], KP[p]] = A[p][M, NP[p]]^T * B[p][M, KP[p]], bias_out[p] (may be NULL per problem) = column
 * sums of A[p].  The tiles of all problems share one grid, so the row split is chosen for the SUM of the tiles: 4x fewer partial
 * slabs than four ltrx_gemm_tn calls.  Each result is BIT-IDENTICAL run to run (fixed split and reduction order) but not to the
 * single-problem call's (different split count).  Shapes outside the 256x256-tile kernel fall back to one ltrx_gemm_tn per problem.
 * The pointer / int tables are HOST arrays, read before the call returns.  ws: ltrx_gemm_tn_group_workspace_bytes (sufficient for
 * every row count <= M). */
#define LTRX_GEMM_TN_GROUP_MAX 4
size_t ltrx_gemm_tn_group_workspace_bytes(int nprob, int M, const int* NP, const int* KP);
/* slabs_out / bias_slabs_out / splits_out: all NULL = the call reduces its partial slabs itself (one launch per problem); all given =
 * the reduction is left to the caller: problem p's dW is the fixed-order sum of *splits_out slabs of NP[p]*KP[p] floats at slabs_out[p],
 * its bias gradient the column sums of *splits_out rows of NP[p] floats at bias_slabs_out[p] (NULL where bias_out[p] is NULL) -- entries
 * for ltrx_reduce_group, which sums them together with other partials of the same layer in one launch.  *splits_out == 0: the call took
 * the per-problem path and C / bias_out are already final. */
int ltrx_gemm_tn_group(int nprob, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                       float* const* bias_out, int M, const int* NP, const int* KP, int strict, void* ws, size_t ws_bytes,
                       const float** slabs_out, const float** bias_slabs_out, int* splits_out, ltrx_stream_t stream);
/* the same with b_is_image[p] != 0 marking problems whose operand B (the layer INPUT x of dW = dY^T x) is a pre-split image (see
 * ltrx_gemm_nt_img); NULL = none.  LTRX_EUNSUPPORTED if an image is given and the grouped kernel cannot run the call. */
int ltrx_gemm_tn_group_img(int nprob, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                           float* const* bias_out, int M, const int* NP, const int* KP, int strict, void* ws, size_t ws_bytes,
                           const float** slabs_out, const float** bias_slabs_out, int* splits_out, const int* b_is_image,
                           ltrx_stream_t stream);
/* n <= LTRX_REDUCE_GROUP_MAX independent reductions in one launch: dst[i][c] = sum_{s < splits[i]} src[i][s * row_stride[i] + c] for
 * c < cols[i], each in a fixed order (deterministic).  Entries with splits[i] <= 0 are skipped.  (The engine sums the weight-gradient
 * slabs of an encoder layer and the parameter-gradient partials of its LayerNorms with it: 11 launches -> 1; autograd's per-tensor
 * accumulation in loss.backward(), allrank/training/train_utils.py:23.) */
/* test hook, no GPU needed: the workgroup -> (tile, split) table ltrx_gemm_tn_group launches with for these shapes (the tiles of one
 * (problem, split) group on one XCD where they fit); tile_out / split_out: 256 bytes each; returns the workgroup count, 0 when the shapes
 * take the per-problem path. */
int ltrx_debug_tn_group_map(int nprob, int M, const int* NP, const int* KP, unsigned char* tile_out, unsigned char* split_out);
#define LTRX_REDUCE_GROUP_MAX 16
int ltrx_reduce_group(int n, const float* const* src, const int* splits, const size_t* row_stride, const size_t* cols, float* const* dst,
                      ltrx_stream_t stream);

/* Model options around the encoder on the explicit step (allrank_amd/csrc/ltrx_extras.hip):
 *   ltrx_layernorm_torch_fwd: FCModel.input_norm = nn.LayerNorm(n_features) (model.py:27,39): biased variance, eps inside the
 *       sqrt; saves mean and rstd.  Its parameter gradients come from ltrx_layernorm_bwd called with these statistics.
 *   ltrx_posenc_fwd: positional encoding (positional.py:15-77, transformer.py:51-52): y = scale * x + table[row(m)],
 *       row = padding_idx for masked items and ranks outside [0, padding_idx), else indices[m]; mask may be NULL.
 *   ltrx_posenc_table_bwd: gradient of a LEARNED table: dtable[r] = sum of dx[m] over the rows m with row(m) == r; the padding
 *       row gets 0 (nn.Embedding(padding_idx)); deterministic.      ltrx_scale_inplace: x *= s (the sqrt(d_model) factor).
 *   ltrx_out_act_fwd / _bwd: OutputLayer activation (model.py:106-117), kind 1 = Sigmoid, 2 = Tanh; bwd: dz = dy * act'(y). */
int ltrx_layernorm_torch_fwd(const float* x, const float* w, const float* b, int rows, int D, float eps, float* y, float* mean_out,
                             float* rstd_out, ltrx_stream_t stream);
int ltrx_posenc_fwd(const float* x, const float* table, const int64_t* indices, const uint8_t* mask, int M, int D, int padding_idx,
                    float scale, float* y, ltrx_stream_t stream);
int ltrx_posenc_table_bwd(const float* dx, const int64_t* indices, const uint8_t* mask, int M, int D, int padding_idx, float* dtable,
                          ltrx_stream_t stream);
int ltrx_scale_inplace(float* x, size_t n, float s, ltrx_stream_t stream);
int ltrx_out_act_fwd(const float* z, size_t n, int kind, float* y, ltrx_stream_t stream);
int ltrx_out_act_bwd(const float* dy, const float* y, size_t n, int kind, float* dz, ltrx_stream_t stream);

/* Slate-resident training step of an FCModel([H]) -> OutputLayer(H, 1) -> listNet model (BASELINE configs[1]): forward, loss,
 * backward and -- optionally -- the Adam update in TWO launches that read the features from HBM once
 * (allrank_amd/csrc/ltrx_fcstep.hip).  Replaces, for this model family, the call sequence of loss_batch
 * (allrank/training/train_utils.py:18-29): FCModel.forward (allrank/models/model.py:35-44, one Linear + activation `act`:
 * 0 = None, 1 = ReLU, no dropout, no input_norm), OutputLayer.forward (model.py:111-117, d_output 1, no activation),
 * listNet (allrank/models/losses/listNet.py:8-30), autograd's backward of the three, and torch.optim.Adam / AdamW.step.
 *   x[B,L,F], y[B,L] (pad_value marks padded slots); `params` = ONE flat fp32 buffer holding W1[H,F] at off_w1 (= 0), b1[H] at
 *   off_b1, w_out[H] at off_wout, b_out[1] at off_bout: adjacent segments, each padded to a multiple of 4 floats, nflat in total;
 *   `grads` has the same layout and receives d loss / d params; scores[B,L] = model.score(x); dscores (optional) = d loss / d scores;
 *   hidden_out (optional, tests) = the FC activations [B,L,H]; loss_out[1] = sum of the per-slate losses / batch_divisor.
 *   exp_avg != NULL: the Adam update is applied to `params` in the reducing launch (step_count: device float, bumped by the
 *   call; decoupled != 0: AdamW) -- same update rule as ltrx_adam_step; NULL: gradients only (sharded runs all-reduce them first).
 * Arithmetic: the two contractions (x W1^T and x^T dh) are three bf16 MFMA products per fp32 product with fp32 accumulation, as
 * in ltrx_gemm_nt / ltrx_gemm_tn; everything else fp32.  Partial gradients are summed in a fixed order (deterministic).
 * ltrx_fc_listnet_supported: 1 when (L, F, H) fit the kernel (L <= 256, F <= 144 and F % 4 == 0, H <= 96), else 0 and the step
 * call returns LTRX_EUNSUPPORTED (the caller then runs the GEMM launch sequence). */
int ltrx_fc_listnet_supported(int L, int F, int H);
size_t ltrx_fc_listnet_workspace_bytes(int B, int L, int F, int H, size_t nflat);
int ltrx_fc_listnet_step(const float* x, const float* y, int B, int L, int F, int H, int act, float* params, size_t off_w1,
                         size_t off_b1, size_t off_wout, size_t off_bout, size_t nflat, float eps, float pad_value,
                         float batch_divisor, float* scores, float* dscores, float* hidden_out, float* loss_out, float* grads,
                         float* exp_avg, float* exp_avg_sq, float* step_count, float lr, float beta1, float beta2, float adam_eps,
                         float weight_decay, int decoupled, void* ws, ltrx_stream_t stream);

/* The same step for a LINEAR scorer (FC activation None), opt-in: score = w_out . (W1 x + b1) + b_out = x . v + c with v = W1^T w_out --
 * two consecutive linear maps evaluated as one, exactly -- and dW1 = w_out (x) u, db1 = D w_out, dw_out = W1 u + D b1, db_out = D with
 * u = sum dscore_l x_l, D = sum dscore_l.  Per slate two matrix-vector products in fp32 FMAs (no matrix cores, no operand split); the
 * slate is held in registers, the next one is in flight while it is processed; a workgroup's partial gradient is F + 2 floats.  Same
 * arguments, layout of `params` / `grads`, outputs and optimizer semantics as ltrx_fc_listnet_step (without `act` / hidden_out).
 * Two launches (slates; partials -> gradients + optimizer).  allrank_amd/csrc/ltrx_fcstep.hip. */
size_t ltrx_fc_linear_listnet_workspace_bytes(int B, int F);
int ltrx_fc_linear_listnet_step(const float* x, const float* y, int B, int L, int F, int H, float* params, size_t off_w1, size_t off_b1,
                                size_t off_wout, size_t off_bout, size_t nflat, float eps, float pad_value, float batch_divisor,
                                float* scores, float* dscores, float* loss_out, float* grads, float* exp_avg, float* exp_avg_sq,
                                float* step_count, float lr, float beta1, float beta2, float adam_eps, float weight_decay,
                                int decoupled, void* ws, ltrx_stream_t stream);

/* On-device batch assembly for a CSR training set resident in HBM (allrank_amd/csrc/ltrx_data.hip):
 *   ltrx_fixlength_positions: FixLength (dataset_loading.py:32-93) for the B slates `slates` of a batch: positions[b][l] = the
 *       position inside slate b that fills slot l, -1 = padding.  Short slates are padded (:81-93); slates of >= L items are
 *       sampled without replacement in random order (:70) with the reference's relevance rule (:72-76).  Counter-based
 *       randomness keyed by (`seed`, the slate's id in the dataset): the same seed draws the same sample for a slate whichever
 *       batch or batch row it arrives in (a rank assembling only its block of a global batch == the one-rank run);
 *       max_slate_len <= 12288.
 *   ltrx_assemble_batch: xb[B,L,F], yb[B,L] (-1 on padding), indices[B,L] (= positions) from the CSR arrays (ToTensor +
 *       collate, :19-29). */
int ltrx_fixlength_positions(const int64_t* offsets, const float* y_items, const int64_t* slates, int B, int L, int max_slate_len,
                             uint64_t seed, int64_t* positions, ltrx_stream_t stream);
int ltrx_assemble_batch(const float* x_items, const float* y_items, const int64_t* offsets, const int64_t* slates,
                        const int64_t* positions, int B, int L, int F, float* xb, float* yb, int64_t* indices, ltrx_stream_t stream);

/* libsvm / SVMlight text parsed on the device (the reference: sklearn's load_svmlight_file on the host, dataset_loading.py:130).
 * text = the file's bytes in device memory, line_start[n_lines] = byte offset of every line.  One thread per line.
 * Pass 1 (X NULL): y[line], qid[line], minmax_index = {smallest, largest} feature index (initialise to {INT_MAX, -1}).
 * Pass 2: X[line][index - index_base] = value (X zero-initialised by the caller, dense [n_lines, n_features]).
 * bad_lines (device int, zero-initialised) counts malformed lines. */
int ltrx_libsvm_parse(const uint8_t* text, const int64_t* line_start, int64_t n_lines, int64_t n_bytes, float* y, int64_t* qid, float* X,
                      int n_features, int index_base, int* minmax_index, int* bad_lines, ltrx_stream_t stream);

/* Test hook: D[32x32] = A[32x2] * B[2x32] with ONE v_mfma_f32_32x32x2_f32, written through the operand / result
 * lane layout the attention kernels assume.  Lets the parity suite tell a layout bug from a logic bug. */
int ltrx_selftest_mfma32x32x2(const float* A, const float* B, float* D, ltrx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LTRX_H */
