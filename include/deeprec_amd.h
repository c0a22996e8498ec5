/* deeprec_amd.h -- C ABI of the MI355X-native DeepRec embedding/interaction hot path.
 *
 * The reference (cheungdaven/DeepRec) has no FFI: its hot path runs inside TensorFlow
 * ops called from Python model classes.  Each entry point below therefore cites the
 * reference call site (file:line under /root/reference) whose TensorFlow op(s) it
 * replaces; INTEGRATION.md shows the ctypes binding a maintainer adds on the Python side.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (DR_ERR_*); dr_last_error()
 *     returns the message of the calling thread's last failure;
 *   - all pointers except `stream` are DEVICE pointers owned by the caller (torch
 *     tensors); the library never allocates or frees memory: scratch comes in through
 *     a workspace whose size the matching *_workspace_bytes() query returns;
 *   - `stream` is a hipStream_t (NULL = the null stream); all work is enqueued
 *     asynchronously on it; entry points are re-entrant across streams as long as
 *     workspaces are not shared;
 *   - tables and activations are fp32 row-major [rows, k]; ids are int32 (id_bits=32)
 *     or int64 (id_bits=64);
 *   - duplicate-row sums use the "blocked position order" defined in DESIGN.md
 *     (deterministic, bit-repeatable; DR_SEG_CHUNK contributions per block).
 */
#ifndef DEEPREC_AMD_H
#define DEEPREC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_OK 0
#define DR_ERR_ARG (-1)       /* null pointer, bad size, unsupported k / id_bits ... */
#define DR_ERR_WORKSPACE (-2) /* workspace too small                                 */
#define DR_ERR_LAUNCH (-3)    /* HIP reported a launch / runtime failure             */

#define DR_SEG_CHUNK 32

/* out-of-range id policy of dr_gather_f32 ([TF-ext] tf.nn.embedding_lookup raises on
 * CPU and returns a zero row on GPU; both are explicit here). */
#define DR_OOB_ZERO 0  /* write a zero row, count it in *oob_count if non-NULL        */
#define DR_OOB_ERROR 1 /* same device behaviour; callers must check *oob_count != 0   */

/* optimizer kinds */
#define DR_OPT_SGD 0
#define DR_OPT_ADAM 1    /* [TF-ext] ApplyAdam kernel; eps 1e-8 (TF1) or 1e-7 (Keras) */
#define DR_OPT_ADAGRAD 2 /* [TF-ext] ApplyAdagrad; accumulators start at 0.1          */
#define DR_OPT_RMSPROP 3 /* [TF-ext] ApplyRMSProp (nnmf.py:84)                        */

/* L2 regulariser modes */
#define DR_REG_TOUCHED 0 /* grad = segsum + reg*w on gathered rows only (deviation)   */
#define DR_REG_DENSE 1   /* reference semantics: every row gets reg*w (bprmf.py:118-120) */
#define DR_REG_NONE 2    /* true sparse gradient, no L2 (cml.py:77-79)                */

/* score kinds of dr_score_topk_f32 */
#define DR_SCORE_DOT 0        /* bprmf.py:91 / Caser eval SeqRecMetrics.py:213        */
#define DR_SCORE_NEG_SQDIST 1 /* cml.py:151-153                                       */
#define DR_SCORE_TWO_DIST 2   /* SeqRecMetrics.py:122-123 (AttRec / PRME)             */

/* hyper-parameters of one optimizer application (all fp32) */
typedef struct dr_opt_params {
    int32_t kind;    /* DR_OPT_*                                               */
    float lr;        /* SGD / Adagrad / RMSProp learning rate                  */
    float alpha;     /* Adam: lr*sqrt(1-beta2^t)/(1-beta1^t), computed by host */
    float beta1;     /* Adam                                                   */
    float beta2;     /* Adam                                                   */
    float eps;       /* Adam / RMSProp epsilon                                 */
    float rho;       /* RMSProp decay                                          */
    float momentum;  /* RMSProp momentum                                       */
    float reg;       /* L2 coefficient (reg_rate)                              */
    int32_t reg_mode;/* DR_REG_*                                               */
    const float* alpha_dev; /* optional DEVICE scalar overriding `alpha` (read by the kernels at run time), so a
                               captured hipGraph of a step keeps following Adam's bias correction; see
                               dr_adam_tick_f32.  NULL = use `alpha`.                                  */
    const float* reg_dev;   /* optional DEVICE scalar overriding `reg` the same way: a coefficient that depends on the
                               weights themselves -- nnmf.py:82-83 regularises with tf.norm(W), whose gradient is
                               (reg_rate / ||W||_F) * W -- stays on the device, so the step has no host read-back and
                               can be captured into a hipGraph.  NULL = use `reg`.                      */
} dr_opt_params;

/* The one call that creates anything (no counterpart in the reference): for the calling host thread and `device`, two
 * helper streams and five events (steps that build their plan beside the forward kernel: dr_bpr_step_f32; long-segment
 * passes under graph capture) and the device address of dr_bpr_step_static_f32's scratch array.  Idempotent.  Every
 * other entry point creates no stream or event and allocates nothing; without dr_init() the steps run on the caller's
 * stream alone (identical results). */
int32_t dr_init(int32_t device);

const char* dr_last_error(void);
int32_t dr_version(void);

/* ---- K1: embedding lookup -------------------------------------------------------
 * Replaces tf.nn.embedding_lookup at bprmf.py:37,84-86, gmf.py:66-67, neumf.py:73-76,
 * cml.py:68-70, mf.py:50-53, nfm.py:54,74, nnmf.py:51-54, Caser.py:61-64,
 * AttRec.py:142,145,190-193, PRME.py:62-63.
 * out[n, :] = table[ids[n], :]  (ids of any rank are passed flattened). */
int32_t dr_gather_f32(const float* table, int64_t rows, int32_t k, const void* ids, int64_t n,
                      int32_t id_bits, float* out, int32_t oob_policy, int32_t* oob_count,
                      void* stream);

/* out[e, :] = coef[e] * src[idx[e], :] (idx NULL = e).  No reference counterpart as a separate
 * op: it materialises the per-sample gradient rows g_b * p_u of bprmf.py:121 for the row-sharded
 * path (a21), where they are sent to the rank that owns the destination row before being summed. */
int32_t dr_scale_rows_f32(const float* src, int64_t src_rows, int32_t k, const int32_t* idx,
                          const float* coef, int64_t n, float* out, void* stream);

/* ---- K2: backward of the lookup -------------------------------------------------
 * Replaces the implicit IndexedSlices gradient + de-duplicating segment sum behind
 * tape.gradient (bprmf.py:121) and Optimizer.minimize (gmf.py:77, neumf.py:129,
 * cml.py:79, mf.py:103, nfm.py:82, nnmf.py:84, Caser.py:85, AttRec.py:110, PRME.py:98).
 *
 * A "segment plan" is the stable sort of an id list plus its run boundaries; it lives
 * in a caller-provided workspace (sized for rows of up to k_max floats: it also holds the
 * block partials of long segments) and is consumed by the kernels below. */
int64_t dr_segment_plan_bytes(int64_t n, int32_t k_max);
int32_t dr_segment_plan(const void* ids, int64_t n, int32_t id_bits, int64_t rows, void* plan,
                        int64_t plan_bytes, void* stream);
/* Device pointers into a built plan (valid until the workspace is reused). */
int32_t dr_segment_plan_view(void* plan, int64_t n, const int32_t** sorted_pos,
                             const int32_t** uniq_ids, const int32_t** seg_start,
                             const int32_t** num_segments);

/* gtable[ids[e], :] += coef[e] * src[src_idx[e], :], duplicates combined in blocked
 * position order.  src_idx NULL = e; coef NULL = 1.  Contributions whose src_idx is
 * outside [0, src_rows) and destinations outside [0, rows) are skipped. */
int32_t dr_scatter_add_sorted_f32(float* gtable, int64_t rows, int32_t k, const float* src,
                                  int64_t src_rows, const int32_t* src_idx, const float* coef,
                                  int64_t n, void* plan, int64_t plan_bytes, void* stream);

/* ---- K5: sparse row update ------------------------------------------------------
 * Replaces Optimizer.apply_gradients / minimize on embedding tables (same call sites
 * as K2; hyper-parameters bprmf.py:62, gmf.py:77, cml.py:79, mf.py:103, nfm.py:82).
 * For every segment s of the plan: g = blocked_sum_e coef[e]*src[src_idx[e]] (+ reg*w
 * in DR_REG_TOUCHED), then one optimizer application on row uniq_ids[s] of `table`
 * and its slot tables (Adam: slot0=m, slot1=v; Adagrad: slot0=acc; RMSProp: ms, mom).
 * DR_REG_DENSE is served by dr_dense_update_f32 over a dense gradient instead. */
int32_t dr_sparse_row_update_f32(float* table, float* slot0, float* slot1, int64_t rows, int32_t k,
                                 const float* src, int64_t src_rows, const int32_t* src_idx,
                                 const float* coef, int64_t n, void* plan, int64_t plan_bytes,
                                 const dr_opt_params* opt, void* stream);

/* Whole-tensor optimizer application: g = grad[i] (may be NULL = 0) + reg*w[i].
 * Serves DR_REG_DENSE tables (bprmf.py:118-122: the l2_loss term makes the table
 * gradient dense) and the small dense weights of NeuMF/NFM/NNMF/AttRec/Caser. */
int32_t dr_dense_update_f32(float* w, float* slot0, float* slot1, const float* grad, int64_t count,
                            const dr_opt_params* opt, void* stream);
/* The dense (DR_REG_DENSE) step of a table whose gradient is sparse, without a dense gradient tensor: rows with
 * skip[row] != 0 are left alone (dr_sparse_row_update_f32 with DR_REG_TOUCHED gives them sum + reg*w, the same
 * expression), every other row gets g = 0 + reg*w.  dr_mark_rows_u8 sets flags[ids[e]] = (value != 0) for the valid ids
 * (set before, clear after: the flag array stays all zero between steps).  Replaces zero-fill + scatter-add + a pass
 * that reads the dense gradient ([TF-ext] the reference's densified IndexedSlices + reg*l2_loss gradient). */
int32_t dr_dense_update_masked_f32(float* table, float* slot0, float* slot1, int64_t rows, int32_t k,
                                   const uint8_t* skip, const dr_opt_params* opt, void* stream);
int32_t dr_mark_rows_u8(uint8_t* flags, int64_t rows, const int32_t* ids, int64_t n, int32_t value, void* stream);

/* cml.py:80-81  tf.clip_by_norm(table, clip, axes=[1]) applied in place. */
int32_t dr_rownorm_clip_f32(float* table, int64_t rows, int32_t k, float clip, void* stream);

/* ---- K3/K4/K5 fused: pairwise (BPR) training step -------------------------------
 * Replaces BPRMF.train_op (bprmf.py:114-123): 3 gathers, 2 dot products, -log(sigmoid)
 * loss, gradient, de-duplication and optimizer step.
 * Forward: g[b] = dL/dx_b, loss_b[b], stash[b,:] = g[b]*(Q[i_b]-Q[j_b]).
 * Then Q rows are updated from (ids=[i;j], coef=[g;-g], src=P[u]) and P rows from the
 * stash, both through dr_sparse_row_update_f32 semantics (or the dense path).
 * ws must hold dr_pairwise_step_bytes(...) (the same query serves the pointwise and hinge
 * steps).  loss_out[0] = sum_b loss_b (fixed-order reduction; the L2 term is NOT included
 * -- hosts add it if they display it).  Samples with an out-of-range id contribute nothing. */
int64_t dr_pairwise_step_bytes(int64_t B, int32_t k, int64_t num_user, int64_t num_item,
                               int32_t reg_mode);
int32_t dr_bpr_step_f32(float* P, float* Pslot0, float* Pslot1, int64_t num_user, float* Q,
                        float* Qslot0, float* Qslot1, int64_t num_item, int32_t k,
                        const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                        const dr_opt_params* opt, float* loss_out, void* ws, int64_t ws_bytes,
                        void* stream);

/* The same step split for input pipelines: the id lists and segment plans of a batch depend on its ids
 * only (the reference builds the whole epoch's id lists up front, bprmf.py:126-140), so a host may build
 * them for batch t+1 on another stream while step t runs.  dr_bpr_plan_f32 fills the plan part of `ws`;
 * dr_bpr_step_planned_f32 is dr_bpr_step_f32 minus that work and must see the plan of the SAME (u, i, j). */
int32_t dr_bpr_plan_f32(int64_t num_user, int64_t num_item, int32_t k, const int32_t* u, const int32_t* i,
                        const int32_t* j, int64_t B, int32_t reg_mode, void* ws, int64_t ws_bytes,
                        void* stream);
int32_t dr_bpr_step_planned_f32(float* P, float* Pslot0, float* Pslot1, int64_t num_user, float* Q,
                                float* Qslot0, float* Qslot1, int64_t num_item, int32_t k,
                                const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                                const dr_opt_params* opt, float* loss_out, void* ws, int64_t ws_bytes,
                                void* stream);

/* dr_bpr_step_f32 for callers that cannot allocate device memory (code under HIP graph capture, tight serving /
 * online-update loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same arithmetic, same
 * results bit for bit as dr_bpr_step_f32, but the step's workspace is a file-scope device array inside the library
 * (4 MiB per GPU, held until the process ends) instead of a caller buffer, and the call allocates / frees nothing
 * and creates no streams or events.  Limits: B <= DR_STATIC_MAX_B, k <= DR_STATIC_MAX_K, opt->reg_mode !=
 * DR_REG_DENSE (that mode needs table-sized dense gradients); anything beyond is DR_ERR_ARG / DR_ERR_WORKSPACE with
 * a message.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads)
 * race -- serialise them.  The array keeps the last call's contents; every field is rewritten before it is read.
 * The array's device address is looked up by dr_init(device) (or, failing that, by the first call on a device): call
 * dr_init() before capturing this entry point into a graph. */
#define DR_STATIC_MAX_B 1024 /* the reference's BPRMF batch size (bprmf.py:41) */
#define DR_STATIC_MAX_K 128
int32_t dr_bpr_step_static_f32(float* P, float* Pslot0, float* Pslot1, int64_t num_user, float* Q,
                               float* Qslot0, float* Qslot1, int64_t num_item, int32_t k,
                               const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                               const dr_opt_params* opt, float* loss_out, void* stream);

/* Forward only: x_ui - x_uj and -log(sigmoid) per sample (bprmf.py:84-92,118-119). */
int32_t dr_bpr_forward_f32(const float* P, const float* Q, int64_t num_user, int64_t num_item,
                           int32_t k, const int32_t* u, const int32_t* i, const int32_t* j,
                           int64_t B, float* g_out, float* loss_b, float* stash, void* stream);

/* ---- pointwise models: GMF (gmf.py:57-79, loss_kind 0 = BCE on sigmoid(p.q)) and biased MF
 * (mf.py:37-55,96-103, loss_kind 1 = squared error on p.q + b_u + b_i + mu).
 * forward: pred[b] (sigmoid / rating), and when labels y are given loss_b[b], g[b] = dL/dlogit,
 * stash[b,:] = g[b]*Q[i_b].  Any of g_out / loss_b / pred / stash may be NULL (predict path:
 * gmf.py:160-161, mf.py:117-118).  The step updates BU, BI (k=1 tables), Q, P in that order. */
int32_t dr_pointwise_forward_f32(const float* P, const float* Q, const float* BU, const float* BI,
                                 float mu, int64_t num_user, int64_t num_item, int32_t k,
                                 int32_t loss_kind, const int32_t* u, const int32_t* i, const float* y,
                                 int64_t B, float* g_out, float* loss_b, float* pred, float* stash,
                                 void* stream);
int32_t dr_pointwise_step_f32(float* P, float* Pslot0, float* Pslot1, int64_t num_user, float* Q,
                              float* Qslot0, float* Qslot1, int64_t num_item, int32_t k, float* BU,
                              float* BUslot0, float* BUslot1, float* BI, float* BIslot0,
                              float* BIslot1, float mu, int32_t loss_kind, const int32_t* u,
                              const int32_t* i, const float* y, int64_t B, const dr_opt_params* opt,
                              float* loss_out, void* ws, int64_t ws_bytes, void* stream);

/* ---- CML step (cml.py:72-81,120): d = sum(dropout((p-q)^2)), hinge max(d_pos-d_neg+margin,0),
 * sparse Adagrad on the gathered rows (opt->reg_mode must be DR_REG_NONE), then
 * clip_by_norm(rows, clip) over both tables (clip <= 0 skips it: CMLwarp-style callers).
 * mask_pos / mask_neg are [B,k] 0/1 dropout keep masks (NULL = keep all); kept elements are
 * scaled by 1/keep ([TF-ext] tf.nn.dropout). */
int32_t dr_hinge_step_f32(float* P, float* Pacc, int64_t num_user, float* Q, float* Qacc,
                          int64_t num_item, int32_t k, const int32_t* u, const int32_t* i,
                          const int32_t* j, const uint8_t* mask_pos, const uint8_t* mask_neg,
                          float keep, float margin, float clip, int64_t B, const dr_opt_params* opt,
                          float* loss_out, void* ws, int64_t ws_bytes, void* stream);

/* ---- predict: per-pair scores ---------------------------------------------------
 * Replaces BPRMF.predict (bprmf.py:168-173) and the dot inside GMF/MF predict.
 * out[n] = k-ordered fp32 fmaf chain of A[a_idx[n],:] . B[b_idx[n],:]  (the same chain
 * the MFMA scoring kernel produces, so predict() and score_topk agree bit for bit). */
int32_t dr_pair_dot_f32(const float* A, int64_t rowsA, const float* B, int64_t rowsB, int32_t k,
                        const int32_t* a_idx, const int32_t* b_idx, int64_t n, float* out,
                        void* stream);

/* ---- NFM bi-interaction (nfm.py:54-61) ------------------------------------------
 * fm[b,:] = 0.5*((sum_f v)^2 - sum_f v^2), v = E[feats[b,f],:]; sum_out[b,:] = sum_f v.
 * bwd: gv[b,f,:] = gfm[b,:] * (sum[b,:] - v[b,f,:])  (row gradients for K2). */
int32_t dr_bi_interaction_fwd_f32(const float* E, int64_t rows, int32_t k, const int32_t* feats,
                                  int64_t B, int32_t F, float* fm, float* sum_out, void* stream);
int32_t dr_bi_interaction_bwd_f32(const float* E, int64_t rows, int32_t k, const int32_t* feats,
                                  int64_t B, int32_t F, const float* gfm, const float* sum_in,
                                  float* gv, void* stream);

/* ---- K6: full-catalog scoring + masked top-K + rank-of-target -------------------
 * Replaces the per-user predict + Python sort of RankingMetrics.evaluate
 * (RankingMetrics.py:82-101) and the numpy all-pairs scoring of SeqRecMetrics
 * (SeqRecMetrics.py:122-123, 213).  Scores are an fp32 MFMA GEMM (k-ordered fmaf chain)
 * and never leave the chip.
 *   U[Nu,k], I[Ni,k] (+ U2/I2 and alpha for DR_SCORE_TWO_DIST, item_bias[Ni] optional)
 *   train CSR (train_ptr[Nu+1], train_idx sorted ascending per user): masked items
 *   targets CSR (tgt_ptr[Nu+1], tgt_idx): held-out items whose rank is wanted
 * Outputs: topk_idx/topk_score [Nu,K] ordered by (score desc, item id asc) with -1 / -inf
 * padding; tgt_rank[nnz targets] = number of unmasked items ranked before the target
 * (or -1 if the target is masked). */
int64_t dr_score_topk_bytes(int64_t Nu, int64_t Ni, int32_t k, int32_t K, int64_t n_targets,
                            int32_t score_kind);
int32_t dr_score_topk_f32(const float* U, const float* U2, int64_t Nu, const float* I,
                          const float* I2, const float* item_bias, int64_t Ni, int32_t k,
                          int32_t score_kind, float alpha, const int32_t* train_ptr,
                          const int32_t* train_idx, const int32_t* tgt_ptr, const int32_t* tgt_idx,
                          int64_t n_targets, int32_t K, int32_t* topk_idx, float* topk_score,
                          int32_t* tgt_rank, void* ws, int64_t ws_bytes, void* stream);

/* Ranks of the held-out items in the REFERENCE's float32 operation order.  The reference's evaluators
 * score in numpy -- SeqRecMetrics.py:122-123: -alpha * np.sum((p[:,None] - X)**2, axis=2) - (1 - alpha) * np.sum(...),
 * i.e. separately rounded differences and squares added by numpy's pairwise summation -- while dr_score_topk_f32
 * computes the expanded form as fmaf chains on the matrix cores; the two agree to a few ulps, so near-tied
 * neighbours can swap.  This entry point keeps the MFMA pass but lets it decide only what is certain: a candidate whose
 * MFMA score differs from the held-out item's reference-order score by more than a proven bound on the rounding
 * difference is counted (or not) at once; the few inside the band are re-scored in the reference's order and
 * compared exactly (ties: lower item id first, RankingMetrics.py:98).  tgt_rank therefore equals the rank the
 * reference's stable sort gives, bit for bit of the float32 scores.  Score kinds, in numpy float32 arithmetic:
 *   DR_SCORE_DOT         (p * q).sum() [+ item_bias]     DR_SCORE_NEG_SQDIST   -((p - q)**2).sum()
 *   DR_SCORE_TWO_DIST    (-w1) * ((p - X)**2).sum() - w2 * ((s - Q)**2).sum()   (w1 = float32(alpha), w2 = float32(1 - alpha))
 * np.dot (BLAS, SeqRecMetrics.py:213) has no fixed operation order and is not offered.  k <= 8192.
 * ws: dr_score_rank_exact_bytes(...) (holds the band list: 12 bytes x (64 n_targets + 65536) entries; if it ever
 * overflows -- a large part of the catalog within rounding distance of a held-out item -- every rank is recounted
 * from reference-order scores of the whole catalog, slower but still exact). */
int64_t dr_score_rank_exact_bytes(int64_t Nu, int64_t Ni, int32_t k, int64_t n_targets, int32_t score_kind);
int32_t dr_score_rank_exact_f32(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2,
                                const float* item_bias, int64_t Ni, int32_t k, int32_t score_kind, float w1,
                                float w2, const int32_t* train_ptr, const int32_t* train_idx,
                                const int32_t* tgt_ptr, const int32_t* tgt_idx, int64_t n_targets,
                                int32_t* tgt_rank, void* ws, int64_t ws_bytes, void* stream);

/* Same selection on a caller-provided score matrix scores[Nu, Ni] (NeuMF / any model
 * whose score is not a bilinear form; RankingMetrics.py:91-101).  ws: 2 * round_up(4 *
 * n_targets, 256) bytes when n_targets > 0. */
int32_t dr_topk_rank_f32(const float* scores, int64_t Nu, int64_t Ni, const int32_t* train_ptr,
                         const int32_t* train_idx, const int32_t* tgt_ptr, const int32_t* tgt_idx,
                         int64_t n_targets, int32_t K, int32_t* topk_idx, float* topk_score,
                         int32_t* tgt_rank, void* ws, int64_t ws_bytes, void* stream);

/* Ranks only: tgt_rank as dr_score_topk_f32 computes it, without a top-K list -- what the reference's evaluators consume
 * (SeqRecMetrics.py:33,59,75-80: HR@50 / MRR / AUC depend on the ranked list only through the position of the held-out
 * items; RankingMetrics.py:31,52 likewise).  k <= 256 with at most one held-out item per user runs a streaming kernel
 * whose per-tile epilogue is 16 compares per lane counted by scalar popcounts; train items are taken off the count
 * afterwards from gathered correction tiles scored by the same MFMA sequence (bit-identical scores).  Other calls run
 * dr_score_topk_f32's kernels with K = 1.  Same scores, ties and masking rules; ws: dr_score_topk_bytes(..., K = 1, ...). */
int32_t dr_score_rank_f32(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2,
                          const float* item_bias, int64_t Ni, int32_t k, int32_t score_kind, float alpha,
                          const int32_t* train_ptr, const int32_t* train_idx, const int32_t* tgt_ptr,
                          const int32_t* tgt_idx, int64_t n_targets, int32_t* tgt_rank, void* ws, int64_t ws_bytes,
                          void* stream);

/* Dense score matrix out[Nu, Ni] through the same MFMA GEMM (GMF.predict over a catalog,
 * gmf.py:160-161 with act = 1 (sigmoid); SeqRecMetrics.py:122-123, 213 with act = 0).
 * Distance kinds need a workspace of dr_score_topk_bytes(Nu, Ni, k, 1, 0, kind). */
int32_t dr_score_matrix_f32(const float* U, const float* U2, int64_t Nu, const float* I,
                            const float* I2, const float* item_bias, int64_t Ni, int32_t k,
                            int32_t score_kind, float alpha, int32_t act, float* out, void* ws,
                            int64_t ws_bytes, void* stream);

/* ---- helpers --------------------------------------------------------------------
 * Counter-based ~N(0, stddev^2) init: w[i] = stddev * normal(hash(seed, offset + i)); the
 * same function is restated in oracle/ so 5 GB tables never travel (stands in for
 * tf.random.normal at bprmf.py:75-76, whose RNG stream cannot be reproduced). */
int32_t dr_init_normal_f32(float* w, int64_t count, uint64_t seed, float stddev, uint64_t offset,
                           void* stream);
/* Device-side Adam step state, like TensorFlow's beta1_power / beta2_power variables ([TF-ext]
 * AdamOptimizer._finish): state[0] *= beta1, state[1] *= beta2 (fp32 running products, initialise to 1),
 * state[2] = lr * sqrt(1 - state[1]) / (1 - state[0]).  Point dr_opt_params.alpha_dev at &state[2]. */
int32_t dr_adam_tick_f32(float* state, float lr, float beta1, float beta2, void* stream);
/* n ticks in one launch (same operations, same bits as n calls of dr_adam_tick_f32): alphas[t] = state[2] after tick
 * t + 1, the state ends where n single ticks would leave it.  An eager step loop points dr_opt_params.alpha_dev at
 * alphas[t] for its t-th step and pays one tiny launch per n steps instead of one per step; a captured graph keeps
 * dr_adam_tick_f32 (fixed address). */
int32_t dr_adam_tick_many_f32(float* state, float lr, float beta1, float beta2, float* alphas, int32_t n,
                              void* stream);
/* Fixed-order sum of x[0..n) into out[0] (deterministic loss reduction). */
int32_t dr_sum_f32(const float* x, int64_t n, float* out, void* stream);

/* ---- negative sampler ------------------------------------------------------------
 * The reference draws, per positive, one (BPRMF bprmf.py:130-133, CML cml.py:108-111) or n_neg (GMF / NeuMF
 * gmf.py:108-112, AttRec AttRec.py:448-466) items uniformly from neg_items[u] = all items minus the user's
 * train items, a U x I Python structure (bprmf.py:175-181).  Here: train CSR (idx ascending per user) and
 * out[e*n_neg + d] = neg_items[users[e]][s] with s = floor(r * |neg_items[u]| / 2^32), r = the 32 high bits of
 * the counter-based generator of dr_init_normal_f32 at counter offset + e*n_neg + d: the same distribution as
 * the reference (its Python / numpy RNG stream cannot be reproduced), the same numbers as
 * oracle/c/oracle.c:or_sample_negatives.  Candidate items are first_item .. num_item-1 (first_item = 1 for the
 * sequence loaders' +1 shift).  A user with no candidate, or out of range, gets -1 (the step kernels skip it). */
int32_t dr_sample_negatives_i32(const int32_t* train_ptr, const int32_t* train_idx, const int32_t* users, int64_t n,
                                int32_t n_neg, int64_t num_user, int64_t num_item, int64_t first_item,
                                uint64_t seed, uint64_t offset, int32_t* out, void* stream);

/* The sequence models' sampler (AttRec.py:448-466 _get_neg_items_sbpr, used by AttRec.py:356-387 and PRME.py:133-160):
 * the pool of sample e is every item first_item .. num_item-1 except the sample's own targets[e, 0..T) (T <= 16);
 * out[e*n_neg + d] = pool[floor(r * |pool| / 2^32)] with the same counter-based r as above.  Same distribution as the
 * reference's np.random.randint(len(pool)) per draw; the same numbers as oracle/c/oracle.c:or_sample_negatives_excl. */
int32_t dr_sample_negatives_excl_i32(const int32_t* targets, int64_t n, int32_t T, int32_t n_neg, int64_t num_item,
                                     int64_t first_item, uint64_t seed, uint64_t offset, int32_t* out, void* stream);

/* ---- multi-GPU routing helper (no counterpart in the reference, which is single-device) ----
 * Stable bucketing of n small keys in [0, nkeys), nkeys <= 256 (key = owning rank of a looked-up row):
 * order[x] = position of the x-th element in ascending key order, ties in input order; counts[b] = number of
 * keys equal to b.  Keys outside the range sort last and are not counted.  ws: dr_bucket_order_bytes(n). */
int64_t dr_bucket_order_bytes(int64_t n);
int32_t dr_bucket_order_i32(const int32_t* keys, int64_t n, int32_t nkeys, int32_t* order, int64_t* counts, void* ws,
                            int64_t ws_bytes, void* stream);

/* ---- one route per training step of a data-parallel model (no counterpart in the reference) ----
 * The lookups of one AttRec / Caser step (AttRec.py:142,190-193,205,215; Caser.py:61-64,165-166) read several tables
 * through a few id lists (item sequence, users, targets | negatives).  Every rank all-gathers its concatenated lists
 * (ids_all [world, n_tot], list l = entries list_start[l] .. list_start[l+1] of each rank's part) and derives BOTH roles
 * locally: for entry x = s * n_tot + p (rank s, position p, list l), owner = id mod world, local_rows[x] = floor(id /
 * world); keys_mine[p] = l * world + owner for this rank's own entries (a stable bucketing gives the order its rows
 * travel in, per owner); keys_owned[x] = l * world + s where this rank owns the row, else n_lists * world (the
 * bucketing's overflow bucket) -- a stable bucketing lists, per list and source rank, the rows it serves, in the order
 * a single process sees for the concatenated global batch.  n_lists <= DR_ROUTE_MAX_LISTS, n_lists * world <= 256. */
#define DR_ROUTE_MAX_LISTS 16
int32_t dr_step_route_keys_i32(const int32_t* ids_all, int32_t world, int32_t me, int64_t n_tot,
                               const int64_t* list_start, int32_t n_lists, int32_t* keys_mine, int32_t* keys_owned,
                               int32_t* local_rows, void* stream);
/* Wire buffers of that route: all tables' rows travel in ONE all-to-all (and all gradient rows in one back).  `flat` is
 * a float array; entries e in [seg_start[s], seg_start[s+1]) form segment s (HOST arrays, nseg <= DR_WIRE_MAX_SEG,
 * seg_start[0] = 0, seg_start[nseg] = n), whose rows lie contiguously from flat + seg_off[s] (floats; a multiple of 4,
 * 2 or 1 as k is).  pack: flat row of entry e = src[idx[e] - idx_bias] (idx NULL: src[e]; a row outside [0, src_rows)
 * packs zeros, like dr_gather_f32).  unpack: dst[idx[e] - idx_bias] = flat row of entry e (idx NULL: dst[e]; rows
 * outside [0, dst_rows) are skipped; idx must not repeat a row). */
#define DR_WIRE_MAX_SEG 64
int32_t dr_rows_pack_f32(const float* src, int64_t src_rows, int32_t k, const int32_t* idx, int32_t idx_bias,
                         int64_t n, const int64_t* seg_start, const int64_t* seg_off, int32_t nseg, float* flat,
                         int64_t flat_floats, void* stream);
int32_t dr_rows_unpack_f32(const float* flat, int64_t flat_floats, int32_t k, const int32_t* idx, int32_t idx_bias,
                           int64_t n, const int64_t* seg_start, const int64_t* seg_off, int32_t nseg, float* dst,
                           int64_t dst_rows, void* stream);

/* ---- small dense towers behind the embeddings -------------------------------------------
 * h_0 = x [B, dims[0]];  h_{l+1} = act_l(h_l W_l + b_l), W_l [dims[l], dims[l+1]] row-major, l = 0 .. L-1 (L <= 8, every
 * width <= 256).  Replaces the tf.layers.dense chains of neumf.py:80-113, nnmf.py:60-80 and nfm.py:64-69: one kernel for
 * the whole forward, one for the deltas, one for all weight / bias gradients (fp32 fmaf chains, k resp. batch rows in
 * ascending order).  dims / act / W / b / gW / gb are HOST arrays (of ints resp. device pointers).
 * acts [B, sum(dims[1..L])]: every layer's output, row-major, layer l at column offset sum(dims[1..l]); the tower's
 * result is the last dims[L] columns.  dr_mlp_backward_f32: gout [B, dims[L]] = dLoss/dh_L; deltas = scratch shaped
 * like acts; gx [B, dims[0]] (NULL: not wanted); gW[l] / gb[l] are overwritten (gb or gb[l] NULL: not wanted). */
#define DR_ACT_NONE 0
#define DR_ACT_RELU 1
#define DR_ACT_SIGMOID 2
int32_t dr_mlp_forward_f32(const float* x, int64_t B, int32_t L, const int32_t* dims, const int32_t* act,
                           const float* const* W, const float* const* b, float* acts, void* stream);
int32_t dr_mlp_backward_f32(const float* x, int64_t B, int32_t L, const int32_t* dims, const int32_t* act,
                            const float* const* W, const float* acts, const float* gout, float* deltas, float* gx,
                            float* const* gW, float* const* gb, void* stream);

/* ---- full-catalog scores of NeuMF's pairwise tower (reference models/item_ranking/neumf.py:80-115 evaluated
 * for every (user, item) pair by utils/evaluation/RankingMetrics.py:82-101 through predict, neumf.py:213-214) -------
 * The first dense layer is linear in concat[item, user] and is passed split: uh [nu, dims[0]] = mlp_P[u] W0[user rows]
 * + b0, ih [ni, dims[0]] = mlp_Q W0[item rows] (two small GEMMs of the caller).  For every pair:
 *   h1 = relu(uh[u] + ih[i]);  h_{l+1} = relu(h_l W[l] + b[l]), W[l] [dims[l], dims[l+1]] row-major, l = 0..3;
 *   out[u * ldo + i] += sum_n h5[n]          (out arrives holding the other summand of the logit: the GMF dot <P[u], Q[i]>)
 * One MFMA kernel, activations never leave registers; fp32, fixed summation order (features ascending in groups of 4).
 * dims / W / b are HOST arrays (5 ints, 4 + 4 device pointers); dims[0] % 16 == 0, every width <= 128; shapes are
 * instantiated per 16-wide tile counts (the reference's defaults 128-80-40-20-10 and a few others), anything else is
 * refused with DR_ERR_ARG. */
int32_t dr_tower_catalog_scores_f32(const float* uh, int64_t nu, const float* ih, int64_t ni, const int32_t* dims,
                                    const float* const* W, const float* const* b, float* out, int64_t ldo, void* stream);

/* ---- AttRec self-attention pooling (reference models/seq_rec/AttRec.py:147-186) ----------------
 * pre [B, L, d] = (Q[item_seq] + timing signal) W, the pre-activation of the shared "linear_project" layer
 * (AttRec.py:150-160; the projection itself stays a library GEMM); v [B, L, d] = Q[item_seq] (AttRec.py:142).
 *   q = relu(pre);  s = softmax(q q^T / sqrt(d)) over the last axis (AttRec.py:174-177);
 *   w = s with the diagonal zeroed after the softmax (AttRec.py:178-180);
 *   out [B, d] = reduce_mean(w v, axis 1) (AttRec.py:182-186) = (sum_m c[m] v[m]) / L, c[m] = sum_{l != m} s[l][m].
 * forward writes out and s [B, L, L] (the unmasked softmax: the backward's saved tensor; the reference's `weights`
 * output is s with a zero diagonal).  backward: gout [B, d] = dLoss/dout -> dpre [B, L, d] (through the softmax, both
 * uses of q and the ReLU) and dv [B, L, d] = (c[m] / L) gout, the direct path into the values only -- the caller adds
 * dpre W^T for the path through the projection.  1 <= L <= 64, 1 <= d <= 256.  fp32, every sum in a fixed order. */
int32_t dr_attn_pool_forward_f32(const float* pre, const float* v, int64_t B, int32_t L, int32_t d, float* out,
                                 float* s, void* stream);
int32_t dr_attn_pool_backward_f32(const float* pre, const float* v, const float* s, const float* gout, int64_t B,
                                  int32_t L, int32_t d, float* dpre, float* dv, void* stream);

/* ---- Caser horizontal convolutions (reference models/seq_rec/Caser.py:132-154) ------------------
 * For every height h = 1 .. L: conv2d(kernel (h, d), C filters, valid) over e = Q[item_seq] [B, L, d], ReLU, max over
 * the L - h + 1 positions; the pooled outputs are concatenated in ascending h (Caser.py:153-154): out [B, C L], column
 * (h - 1) C + c.  All heights share one filter matrix W [C L (L + 1) / 2, d]: row C h (h - 1) / 2 + c h + r = row r of
 * filter c of height h; bias [C L] in the order of `out`.
 * dr_caser_hconv_pool_f32: G [B L, ldg] = e W[rows of heights h0 .. h1]^T (a library GEMM; column 0 = row
 * C h0 (h0 - 1) / 2 of W) -> out / arg for those heights: out = max(0, max_t (sum_r G[b, t + r, (h, c, r)] + bias)),
 * arg = the winning position t, or -1 where the ReLU is inactive.
 * dr_caser_hconv_direct_f32: the same out / arg for ALL heights straight from e and W (no G): only the L (L+1) (L+2) / 6
 * (row, filter row) products per filter that some window uses, on v_mfma_f32_16x16x4_f32 with 16 sequences as the M
 * dimension.  Needs C <= 16 and d % 16 == 0 (DR_ERR_ARG otherwise: use the pooling form).
 * dr_caser_hconv_backward_input_f32: gout [B, C L] = dLoss/dout, arg -> ge [B, L, d] (overwritten):
 * ge[b, arg + r] += gout[b, (h, c)] W[(h, c, r)], summed in ascending (h, c, r).
 * dr_caser_hconv_backward_filter_f32: -> partial [S, C L (L + 1) / 2, d]: the filter gradient of batch chunk s (S
 * chunks of ceil(B / S) sequences, ascending inside a chunk); dLoss/dW = the sum over s.  The bias gradient is the
 * column sum of gout where arg >= 0.  1 <= L <= 64, 1 <= C <= 64, 1 <= d <= 256, 1 <= S <= 64. */
int32_t dr_caser_hconv_pool_f32(const float* G, int64_t ldg, const float* bias, int64_t B, int32_t L, int32_t C,
                                int32_t h0, int32_t h1, float* out, int32_t* arg, void* stream);
int32_t dr_caser_hconv_direct_f32(const float* e, const float* W, const float* bias, int64_t B, int32_t L, int32_t C,
                                  int32_t d, float* out, int32_t* arg, void* stream);
/* Vertical convolution (Caser.py:118-130): n_v filters [L, 1] over e [B, L, d]; out [B, d * n_v], column col * n_v + v
 * (the reference's [B, d, n_v] flattening) = bias[v] + sum_l w[v, l] e[b, l, col].  Backward: g [B, d * n_v] ->
 * ge [B, L, d] (overwritten) and partial [S, n_v, L + 1]: workgroup s's share of dLoss/dw (columns 0 .. L-1) and of
 * dLoss/dbias (column L); the gradients are the sums over s.  1 <= L <= 64, 1 <= n_v <= 16; S = the launch's workgroups. */
int32_t dr_caser_vconv_forward_f32(const float* e, const float* w, const float* bias, int64_t B, int32_t L, int32_t d,
                                   int32_t nv, float* out, void* stream);
int32_t dr_caser_vconv_backward_f32(const float* e, const float* w, const float* g, int64_t B, int32_t L, int32_t d,
                                    int32_t nv, int32_t S, float* ge, float* partial, void* stream);
int32_t dr_caser_hconv_backward_input_f32(const float* gout, const int32_t* arg, const float* W, int64_t B, int32_t L,
                                          int32_t C, int32_t d, float* ge, void* stream);
int32_t dr_caser_hconv_backward_filter_f32(const float* gout, const int32_t* arg, const float* e, int64_t B, int32_t L,
                                           int32_t C, int32_t d, int32_t S, float* partial, void* stream);

/* ---- interaction + loss of the sequence models, fused with their item lookups ------------
 * dr_seq_dist_loss_f32 replaces the lookups, squared distances, loss and their backward of AttRec.py:190-232 + :97 and
 * PRME.py:104-131 + :73:  s(b, i) = alpha * ||X[i] - P[users[b]]||^2 + beta * ||o[b] - Q[i]||^2  (beta = 1 - alpha; o =
 * the sequence representation [B, k]: AttRec's attention output, PRME's previous-item row);
 *   loss_kind 0: sum_t max(s(b, tgt[b,t]) - s(b, neg[b,t]) + margin, 0)     (AttRec.py:97; gradient 1 where >= 0)
 *   loss_kind 1: -sum_t log(sigmoid(-s(b, tgt[b,t]) + s(b, neg[b,t])))      (PRME.py:73)
 * tgt, neg int32 [B, T] (pair t = target t with negative t; 2 T <= 64).  Outputs: loss_b [B];
 * gX, gQ [B, 2T, k] = d loss / d X[item], d loss / d Q[item] per (sample, slot), slots = the T targets then the T
 * negatives (feed them to the segment update with ids [tgt | neg] row by row); gp [B, k] = d loss / d P[users[b]];
 * go [B, k] = d loss / d o[b].  A pair with an id out of range contributes nothing.
 * dr_seq_dot_loss_f32 is the same stage of Caser (Caser.py:79-80,184-192): s(b, i) = <x[b], W[i]> + bias[i] over D
 * columns, loss = scale * (-mean_{b,t} log(sigmoid(s_tgt) + 1e-10) - mean_{b,t} log(1 - sigmoid(s_neg) + 1e-10))
 * (the means over B T resp. B Tn entries; scale = 1 / world for a data-parallel slice of the batch); outputs loss_b [B]
 * (its sum is the loss), gW [B, T+Tn, D], gb [B, T+Tn], gx [B, D]. */
int32_t dr_seq_dist_loss_f32(const float* P, int64_t num_user, const float* X, const float* Q, int64_t num_item,
                             int32_t k, const int32_t* users, const float* o, const int32_t* tgt,
                             const int32_t* neg, int64_t B, int32_t T, float alpha, float beta,
                             int32_t loss_kind, float margin, float* loss_b, float* gX, float* gQ, float* gp,
                             float* go, void* stream);
int32_t dr_seq_dot_loss_f32(const float* W, const float* bias, int64_t num_item, int32_t D, const float* x,
                            const int32_t* tgt, const int32_t* neg, int64_t B, int32_t T, int32_t Tn,
                            float scale, float* loss_b, float* gW, float* gb, float* gx, void* stream);

/* ---- row-sharded BPR step (no counterpart in the reference, which is single-device; the step is bprmf.py:114-123;
 * SURVEY 8e, deeprec_amd/sharding.py, csrc/shard.hip) ----
 * Tables are dealt round-robin (owner = row % world, local row = row / world); the id lists of all ranks are REPLICATED
 * (one fixed-size all-gather: ids = [world][3][B] int32, rank s's u, i, j; global sample g = s * B + position).  A rank
 * PROCESSES the samples whose user row it owns and OWNS the item entries (part i / j, sample) with item % world == me,
 * in the canonical order a single process sees for the concatenated batch (i-part before j-part, then g).  Its combined
 * key list [local user rows of its samples ; local item rows it owns] is planned like the single-GPU step's (and is
 * that list at world 1); item rows of its samples that live elsewhere are fetched (distinct rows per owner, ascending:
 * first all-to-all-v), their gradient rows (+-g) p_u leave in the same (owner, row, canonical) order (second
 * all-to-all-v) and are added by the owner with coefficient 1 at their canonical place.  Tables are bit-identical at
 * every world size.  num_user / num_item: GLOBAL table sizes; P / Q and their slots: this rank's shards.
 * cap_s / cap_g: the most samples a rank may process / item entries it may own; every rank sees every rank's load
 * (counts) and refuses a batch beyond them.  k must be V * 2^n (one vector column per lane: 4, 8, ..., 256, or 2 / 1).
 * ws: dr_shard_bpr_bytes(...), 256-byte aligned; it holds the route, the plan, the stash and the per-sample losses of ONE
 * batch (route a batch ahead into a second workspace).  Nothing is allocated and no host synchronisation happens inside.
 *
 * dr_shard_bpr_route    ids only: 2 launches + (world > 1: sort of the remote entries + 2) + the plan (7).  At world 1 ids
 *                       may be NULL and the batch given as u, i, j [B]; counts_host (may be NULL): pinned host memory
 *                       that receives the counts (below) by an asynchronous copy on `stream`.
 * dr_shard_bpr_list     device address of a list inside ws.  which = 0: counts (int32 [8 + 6 world]: [0] samples processed here, [1] entries
 *                       of the combined list, [2] gradient rows sent, [3] sent + received, [4] samples of the GLOBAL batch
 *                       with an id out of range, [5] some rank over a capacity, [6] item entries owned, [7] gradient rows
 *                       received; then per rank r: distinct rows fetched from r | gradient rows sent to r | rows r
 *                       fetches from me | gradient rows r sends me | samples r processes | entries r owns); 1: the local
 *                       rows the peers fetch from this rank (int32 [cap_g], peer-major, ascending); 2 - 7 (tests): combined
 *                       key list, entry sources, item row references, list / wire positions, solo flags, coefficients.
 * dr_shard_bpr_forward  lookups, dots, loss; rows with ONE contribution, and that one local, updated in place; gradient
 *                       rows of remote items -> send (counts[2] rows); fetched: rows received for counts' first list.
 * dr_shard_bpr_finish   every other row of both tables (received: counts[7] rows, peer-major) + the loss sum of the
 *                       samples processed here. */
int64_t dr_shard_bpr_bytes(int32_t world, int64_t B, int32_t k, int64_t cap_s, int64_t cap_g);
int32_t dr_shard_bpr_list(int32_t world, int64_t B, int32_t k, int64_t cap_s, int64_t cap_g, void* ws, int32_t which,
                          void** ptr);
int32_t dr_shard_bpr_route(const int32_t* ids, const int32_t* u, const int32_t* i, const int32_t* j, int32_t world,
                           int32_t me, int64_t B, int64_t num_user, int64_t num_item, int32_t k, int64_t cap_s,
                           int64_t cap_g, void* ws, int64_t ws_bytes, int32_t* counts_host, void* stream);
int32_t dr_shard_bpr_forward(float* P, float* Pslot0, float* Pslot1, float* Q, float* Qslot0, float* Qslot1,
                             int32_t world, int32_t me, int64_t B, int64_t num_user, int64_t num_item, int32_t k,
                             int64_t cap_s, int64_t cap_g, const float* fetched, float* send, const dr_opt_params* opt,
                             void* ws, int64_t ws_bytes, void* stream);
int32_t dr_shard_bpr_finish(float* P, float* Pslot0, float* Pslot1, float* Q, float* Qslot0, float* Qslot1,
                            int32_t world, int32_t me, int64_t B, int64_t num_user, int64_t num_item, int32_t k,
                            int64_t cap_s, int64_t cap_g, const float* received, const dr_opt_params* opt, float* loss_out,
                            void* ws, int64_t ws_bytes, void* stream);

/* ---- measurement hooks (no counterpart in the reference) ---------------------------------------------
 * dr_profile_enable(1): from now on every instrumented kernel launch of this library (the gather kernel, the fused step's forward
 * kernel, the short / long segment update kernels) carries a start / stop HIP event pair that the runtime stamps
 * with the dispatch's own begin and end (hipExtLaunchKernelGGL), on the stream it is launched on;
 * dr_profile_enable(0) stops it.  Do not enable while a stream is being captured into a graph.  Not
 * thread-safe: one profiling host thread at a time.
 * dr_profile_collect: waits for the recorded launches, writes up to `cap` durations (milliseconds, launch
 * order) to ms[] and their kernel names, newline-separated, to names[]; returns the number of records and
 * clears the list.  bench.py's `roofline` comes from these numbers. */
int32_t dr_profile_enable(int32_t on);
int32_t dr_profile_collect(char* names, int64_t names_bytes, float* ms, int32_t cap);
/* The same, plus start_ms[i] = start of launch i relative to the start of the first recorded launch (GPU clock): where
 * the time between a step's kernels goes, without a profiler attached (tools/step_timeline.py). */
int32_t dr_profile_collect_timeline(char* names, int64_t names_bytes, float* ms, float* start_ms, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* DEEPREC_AMD_H */
