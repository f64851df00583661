/*
 * dalm_hip.h — C ABI of libdalm_hip.so: the MI355X (gfx950) kernels behind the
 * RAG-end2end / retriever-only training-step loss path of arcee-ai/DALM.
 *
 * The reference has no FFI of its own (it is pure Python on torch); the seam this
 * library replaces is the eager-torch op sequences listed below.  Every entry
 * point cites the reference lines (relative to the DALM repo root) it stands in
 * for.  The Python host side (dalm_amd/hip.py) binds these with ctypes.
 *
 * Conventions
 *   - All pointers are DEVICE pointers, borrowed for the duration of the call.
 *     The library never allocates or frees caller memory; scratch space is
 *     passed in (`ws`, size from the matching *_workspace_bytes()).
 *   - Work is enqueued on `stream` (a hipStream_t; pass torch's current stream).
 *     Calls are asynchronous and re-entrant; there is no global mutable state
 *     apart from the thread-local last-error string.
 *   - Return value: 0 = ok; negative = argument error (DALM_E_*); positive =
 *     a hipError_t from the launch.  dalm_last_error_string() describes the last
 *     non-zero return on the calling thread.
 *   - Integer tensors (ids, masks, lengths) are int64, exactly what
 *     transformers.default_data_collator hands the reference trainer.
 *   - `dtype` arguments: DALM_F32 or DALM_BF16 (storage type of hidden states /
 *     logits; all arithmetic is fp32 in registers, as in the reference where
 *     accelerate up-casts model outputs to fp32 before the loss code).
 */
#ifndef DALM_HIP_H
#define DALM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dalm_stream_t; /* hipStream_t */

enum { DALM_F32 = 0, DALM_BF16 = 1 };

enum {
  DALM_OK = 0,
  DALM_E_NULL = -1,     /* required pointer is NULL            */
  DALM_E_SHAPE = -2,    /* negative / inconsistent sizes        */
  DALM_E_DTYPE = -3,    /* unknown dtype code                   */
  DALM_E_ALIGN = -4,    /* pointer not aligned to element size  */
  DALM_E_WORKSPACE = -5 /* workspace too small                  */
};

/* ---- library ----------------------------------------------------------- */
int dalm_version(void);                     /* 10000*major + 100*minor + patch */
const char* dalm_last_error_string(void);   /* thread-local, never NULL        */

/* ---- K1: masked mean-pool + L2 normalise -------------------------------
 * Replaces AutoModelForRagE2E.mean_pooling + F.normalize
 *   dalm/models/rag_e2e_base_model.py:95-97,108-111
 *   dalm/models/retriever_only_base_model.py:60-68
 *   u_b = sum_t m_bt h_bt / max(sum_t m_bt, 1e-9);  e_b = u_b / max(|u_b|, 1e-12)
 * h: [B,T,D] contiguous (dtype), mask: [B,T] int64 (any integer weights).
 * Outputs: emb [B,D] f32 (e if normalize else u), norm [B] f32 (|u_b|),
 *          inv_count [B] f32 (1/max(sum m,1e-9)); norm/inv_count feed the bwd.
 */
int dalm_pool_l2norm_fwd(const void* h, int dtype, const int64_t* mask,
                         int64_t B, int64_t T, int64_t D, int normalize,
                         float* emb, float* norm, float* inv_count,
                         dalm_stream_t stream);
/* Same, with scratch for the token-sliced form used when B alone cannot occupy the chip (few, long
 * samples): ws = dalm_pool_l2norm_fwd_workspace_bytes(...) bytes (0 when no slicing is planned; ws may then
 * be NULL).  dalm_pool_l2norm_fwd == this call with ws = NULL (one workgroup per sample, one launch). */
size_t dalm_pool_l2norm_fwd_workspace_bytes(int64_t B, int64_t T, int64_t D, int dtype);
int dalm_pool_l2norm_fwd_ws(const void* h, int dtype, const int64_t* mask,
                            int64_t B, int64_t T, int64_t D, int normalize,
                            float* emb, float* norm, float* inv_count,
                            void* ws, size_t ws_bytes, dalm_stream_t stream);
/* dh: [B,T,D] (dtype) is fully written (zeros where mask == 0). */
int dalm_pool_l2norm_bwd(const float* d_emb, const float* emb, const float* norm,
                         const float* inv_count, const int64_t* mask,
                         int64_t B, int64_t T, int64_t D, int normalize,
                         void* dh, int dtype, dalm_stream_t stream);

/* ---- K2: similarity matmul (materialising) -----------------------------
 * Replaces get_cosine_sim: dalm/training/utils/train_utils.py:76-77
 *   S[m,n] = (A[m,D] . B[n,D]^T) * scale     (exact-f32 MFMA)
 */
int dalm_sim_matmul(const float* A, const float* Bm, int64_t m, int64_t n,
                    int64_t D, float scale, float* S, int64_t ldS,
                    dalm_stream_t stream);

/* General f32 GEMM on the same MFMA kernel, C = alpha * op(A) op(B):
 *   transA == 0: A is [M,K] (lda >= K);  transA != 0: A is [K,M] (lda >= M)
 *   transB == 0: B is [K,N] (ldb >= N);  transB != 0: B is [N,K] (ldb >= K)
 * Used for autograd of get_cosine_sim (dQ = scale dS P, dP = scale dS^T Q). */
int dalm_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                  float alpha, const float* A, int64_t lda, const float* Bm,
                  int64_t ldb, float* C, int64_t ldc, dalm_stream_t stream);

/* ---- K2-K4 fused: in-batch-negatives row statistics ---------------------
 * Replaces get_cosine_sim + get_nt_xent_loss (+ the log_softmax(scores).diag()
 * of compute_marginalized_loss_from_logits) without materialising S:
 *   dalm/training/utils/train_utils.py:76-88,124
 *   dalm/training/rag_e2e/train_rage2e.py:441-446
 * For S = scale * A[m,D] . B[n,D]^T:
 *   row_lse[i] = logsumexp_j S[i,j],   diag[i] = S[i, diag_offset + i]
 * Called once with (A,B) = (Q_local, P_all) and once with (P_local, Q_all): the
 * second call's row statistics are the column statistics of the first.
 * diag_offset = rank * B_local (0 on one GPU).
 */
size_t dalm_sim_rowstats_workspace_bytes(int64_t m, int64_t n, int64_t D);
int dalm_sim_rowstats(const float* A, const float* Bm, int64_t m, int64_t n,
                      int64_t D, float scale, int64_t diag_offset,
                      float* row_lse, float* diag, void* ws, size_t ws_bytes,
                      dalm_stream_t stream);

/* The same row statistics on the bf16 matrix cores at f32 accuracy ("bf16x3", round 4): scale*A and B are split into
 * three bf16 thirds each (x = hi + mid + lo, 24 significand bits), the six products down to 2^-16 relative are laid out
 * along K (depth 6 D, smallest first) and contracted by the 256 x 256 x 64 bf16 MFMA kernel of dalm_lm_head_lse_fwd with
 * its row log-sum-exp epilogue: 6 x the flops of the f32 kernel on a pipe with 16 x its rate, no score matrix.
 * S agrees with the f32 result to a few 1e-7 relative of |scale| (products of bf16 values are exact in f32, accumulation
 * is f32).  Needs D % 64 == 0, D <= 65536, operand images below 4 GB, 16-byte aligned A / B.
 * dalm_sim_rowstats routes to it for m, n >= 3072 (DALM_SIM_BF16X3=0 keeps the f32 MFMA kernel).
 * Reference: get_cosine_sim / get_nt_xent_loss, dalm/training/utils/train_utils.py:76-88; dalm/eval/utils.py:44-68. */
/* dalm_sim_rowstats pinned to the exact-f32 MFMA kernels whatever the size (same arguments, same workspace query). */
int dalm_sim_rowstats_f32(const float* A, const float* Bm, int64_t m, int64_t n,
                          int64_t D, float scale, int64_t diag_offset,
                          float* row_lse, float* diag, void* ws, size_t ws_bytes,
                          dalm_stream_t stream);
int dalm_sim_rowstats_bf16x3_supported(int64_t m, int64_t n, int64_t D);
size_t dalm_sim_rowstats_bf16x3_workspace_bytes(int64_t m, int64_t n, int64_t D);
int dalm_sim_rowstats_bf16x3(const float* A, const float* Bm, int64_t m, int64_t n,
                             int64_t D, float scale, int64_t diag_offset,
                             float* row_lse, float* diag, void* ws, size_t ws_bytes,
                             dalm_stream_t stream);

/* Closed-form backward of the above:
 *   dS[i,j] = row_coef[i] exp(S_ij - row_lse[i]) + col_coef[j] exp(S_ij - col_lse[j])
 *             - [j == diag_offset+i] (row_coef[i] + col_coef[j])
 *   dA = scale * dS . B
 * row_* are length m, col_* length n.
 * D a multiple of 128 and <= 1024: flash-style - S tiles are recomputed, transformed and contracted with B
 * inside one kernel, nothing of size m x n exists; ws only holds per-split partial outputs when m/32 row
 * blocks cannot fill the chip (<= ~512 row blocks x D floats, 16 bytes otherwise).  Other D: an m x n dS
 * panel in ws + a second GEMM.  dA and ws must be 16-byte aligned. */
size_t dalm_sim_grad_workspace_bytes(int64_t m, int64_t n, int64_t D);
int dalm_sim_grad(const float* A, const float* Bm, int64_t m, int64_t n,
                  int64_t D, float scale, int64_t diag_offset,
                  const float* row_coef, const float* row_lse,
                  const float* col_coef, const float* col_lse, float* dA,
                  void* ws, size_t ws_bytes, dalm_stream_t stream);

/* ---- K2-K4, small-batch form (the batch sizes the trainers really run) ----
 * Same reference lines as above (train_utils.py:76-88,124; train_rage2e.py:441-446;
 * train_retriever_only.py:369-374).  For m x n <= 2^20 (m <= 1024, n <= 8192) S is computed ONCE:
 *   fwd: S[m,n] (saved, ldS >= n), row_lse[m], diag[m] and - when col_lse != NULL -
 *        col_lse[n] = logsumexp_i S[i,j]     (2 launches: split-K partial tiles, statistics)
 *   bwd: dA = scale dS . B and/or dB = scale dS^T . A from the saved S (1 launch; pass NULL
 *        for the output that is not wanted), dS as in dalm_sim_grad.
 * One GPU: one fwd with col_lse and one bwd with both outputs replace two dalm_sim_rowstats
 * and two dalm_sim_grad calls (4 computations of S, ~10 launches). */
int dalm_sim_small_supported(int64_t m, int64_t n, int64_t D);
size_t dalm_sim_small_workspace_bytes(int64_t m, int64_t n, int64_t D, int want_cols);
/* dalm_sim_small_fwd1: the same forward in ONE launch (round 4).  The statistics pass is finished inside the partial-tile
 * kernel by the last workgroups to arrive (arrival tickets; merge orders fixed by tile / slice index, so the result does
 * not depend on who arrives last: deterministic, same S bits as the two-launch form, row / column log-sum-exp equal to
 * it within f32 rounding of a different - fixed - merge tree).  `tickets`: dalm_sim_small_fwd1_ticket_words(m, n)
 * 32-bit words that MUST BE ZERO on entry; the kernel leaves them zero (zero them once, when allocating; calls sharing
 * a ticket buffer must be stream-ordered).  Workspace: dalm_sim_small_fwd1_workspace_bytes. */
int dalm_sim_small_fwd1_preferred(int64_t m, int64_t n, int64_t D);   /* 1: measured faster than the two-launch form */
size_t dalm_sim_small_fwd1_workspace_bytes(int64_t m, int64_t n, int64_t D, int want_cols);
size_t dalm_sim_small_fwd1_ticket_words(int64_t m, int64_t n);
int dalm_sim_small_fwd1(const float* A, const float* Bm, int64_t m, int64_t n,
                        int64_t D, float scale, int64_t diag_offset, float* S,
                        int64_t ldS, float* row_lse, float* diag, float* col_lse,
                        void* ws, size_t ws_bytes, unsigned* tickets, dalm_stream_t stream);
int dalm_sim_small_fwd(const float* A, const float* Bm, int64_t m, int64_t n,
                       int64_t D, float scale, int64_t diag_offset, float* S,
                       int64_t ldS, float* row_lse, float* diag, float* col_lse,
                       void* ws, size_t ws_bytes, dalm_stream_t stream);
int dalm_sim_small_bwd(const float* S, int64_t ldS, const float* A, const float* Bm,
                       int64_t m, int64_t n, int64_t D, float scale,
                       int64_t diag_offset, const float* row_coef,
                       const float* row_lse, const float* col_coef,
                       const float* col_lse, float* dA, float* dB,
                       dalm_stream_t stream);
/* The same backward with a workspace.  When only ONE of dA / dB is asked for (the per-rank block of a sharded batch: few
 * row tiles, a long contraction - 150 x 1200, 18 x 144) the contraction is cut into slices that run as separate workgroups
 * and a second launch adds the partial outputs in fixed order; dalm_sim_small_bwd_workspace_bytes() is 0 when the shape is
 * not sliced.  ws may be NULL (the unsliced form runs); 16-byte aligned otherwise. */
size_t dalm_sim_small_bwd_workspace_bytes(int64_t m, int64_t n, int64_t D, int want_dA, int want_dB);
int dalm_sim_small_bwd_ws(const float* S, int64_t ldS, const float* A, const float* Bm,
                          int64_t m, int64_t n, int64_t D, float scale,
                          int64_t diag_offset, const float* row_coef,
                          const float* row_lse, const float* col_coef,
                          const float* col_lse, float* dA, float* dB,
                          void* ws, size_t ws_bytes, dalm_stream_t stream);
/* dalm_sim_small_bwd1: the sliced backward (one direction of a long contraction: the per-rank blocks of a sharded batch) in
 * ONE launch - the last slice of every output tile to arrive adds the slices in slice order (same bits as the two-launch
 * form, whoever arrives last).  `tickets`: dalm_sim_small_bwd1_ticket_words words, ZERO on entry, left zero (may be the
 * forward's ticket buffer: calls sharing one must be stream-ordered).  Workspace: dalm_sim_small_bwd1_workspace_bytes. */
size_t dalm_sim_small_bwd1_workspace_bytes(int64_t m, int64_t n, int64_t D, int want_dA, int want_dB);
size_t dalm_sim_small_bwd1_ticket_words(int64_t m, int64_t n, int64_t D);
int dalm_sim_small_bwd1(const float* S, int64_t ldS, const float* A, const float* Bm, int64_t m, int64_t n,
                        int64_t D, float scale, int64_t diag_offset, const float* row_coef,
                        const float* row_lse, const float* col_coef, const float* col_lse,
                        float* dA, float* dB, void* ws, size_t ws_bytes, unsigned* tickets,
                        dalm_stream_t stream);

/* ---- K3 drop-in: get_nt_xent_loss on a materialised square S -----------
 * dalm/training/utils/train_utils.py:80-88 (cross_entropy(S, arange(n)), mean).
 * S may be a transposed view: element (i,j) at S[i*stride_r + j*stride_c].
 * Writes loss[0] and row_lse[n] (kept for the backward). */
int dalm_nt_xent_fwd(const float* S, int64_t n, int64_t stride_r,
                     int64_t stride_c, float* loss, float* row_lse,
                     dalm_stream_t stream);
/* dS[i,j] (+)= gscale[0]/n * (exp(S_ij - row_lse[i]) - [i==j]);  dS uses the
 * same strides as S; accumulate != 0 adds into dS. */
int dalm_nt_xent_bwd(const float* S, int64_t n, int64_t stride_r,
                     int64_t stride_c, const float* row_lse,
                     const float* gscale, float* dS, int accumulate,
                     dalm_stream_t stream);

/* ---- K5-K7: marginalised causal-LM cross-entropy ------------------------
 * Replaces compute_marginalized_loss_from_logits / marginalize_log_probs /
 * get_nll:  dalm/training/utils/train_utils.py:91-138.
 *   m_bt = mask[b,t+1], y_bt = ids[b,t+1], t in [0,Tg-2]
 *   M = sum m_bt,  N_b = sum_t m_bt [t >= qlen_b - 1]
 *   L_gen = ( sum_bt m_bt (lse_bt - logits[b,t,y_bt]) - sum_b N_b doc_lp_b ) / M
 */
/* prep: stats[0] = M, stats[1] = B (stats holds 2 floats); Nb[B] as above;
 * Mb[B] = sum_t m_bt (per-sample token counts).  qlen may be NULL (N_b = 0).
 * "t >= qlen_b - 1" follows python slice semantics of train_utils.py:100-103
 * (a negative qlen_b - 1 counts from the end of the Tg-1 rows). */
int dalm_marg_ce_prep(const int64_t* mask, const int64_t* qlen, int64_t B,
                      int64_t Tg, float* stats, float* Nb, float* Mb,
                      dalm_stream_t stream);
/* main pass: one read of logits.  row_lse/row_nll are [B*Tg] (entry b*Tg+t; the
 * t = Tg-1 slot and masked rows hold 0).  If dlogits != NULL the gradient for an
 * upstream grad of 1 is written in the same pass,
 *   dlogits[b,t,:] = (m_bt / M) (softmax(logits[b,t,:]) - onehot(y_bt)),  0 at t=Tg-1,
 * and dlogits may alias logits (in-place).  strides are in elements. */
int dalm_marg_ce_fwd(const void* logits, int dtype, int64_t B, int64_t Tg,
                     int64_t V, int64_t stride_b, int64_t stride_t,
                     const int64_t* ids, const int64_t* mask, const float* stats,
                     float* row_lse, float* row_nll, void* dlogits,
                     dalm_stream_t stream);
/* separate backward from the saved row_lse: dlogits = gscale[0] * (m/M)(softmax - onehot). */
int dalm_marg_ce_bwd(const void* logits, int dtype, int64_t B, int64_t Tg,
                     int64_t V, int64_t stride_b, int64_t stride_t,
                     const int64_t* ids, const int64_t* mask, const float* stats,
                     const float* row_lse, const float* gscale, void* dlogits,
                     dalm_stream_t stream);
/* the same with a weight PER ROW in place of the scalar m/M: dlogits[row] = gscale[0] * row_weight[row] * (softmax - onehot)
 * on rows with mask != 0, zeros elsewhere.  row_weight [B*Tg] = the `weights` of dalm_marg_ce_finalize_topk (k retrieved
 * contexts per sample: B here is B*k sequences).  dalm_marg_ce_bwd is this entry point with row_weight = m/M. */
int dalm_marg_ce_bwd_weighted(const void* logits, int dtype, int64_t B, int64_t Tg,
                              int64_t V, int64_t stride_b, int64_t stride_t,
                              const int64_t* ids, const int64_t* mask, const float* stats,
                              const float* row_lse, const float* gscale,
                              const float* row_weight, void* dlogits, dalm_stream_t stream);
/* x *= gscale[0] in place (n elements); no-op kernel exit when gscale[0] == 1. */
int dalm_scale_inplace(void* x, int dtype, int64_t n, const float* gscale,
                       dalm_stream_t stream);
/* deterministic reduction:
 *   out[0] = ( sum_r row_nll[r] - sum_b Nb[b] doc_lp[b] ) / M      (L_gen)
 * doc_lp may be NULL (no retrieval term).  num_rows = B*Tg. */
int dalm_marg_ce_finalize(const float* row_nll, int64_t num_rows,
                          const float* Nb, const float* doc_lp, int64_t B,
                          const float* stats, float* out, dalm_stream_t stream);

/* The same reduction with a CONTEXT extent k ("marginalisation over top-k passages": the reference is the k = 1 case,
 * train_utils.py:123-124, and only muses about more at train_rage2e.py:461-462).  Sample b is generated under k retrieved
 * contexts: row_nll [B,k,Tg] holds, per sequence (b,c), what dalm_marg_ce_fwd leaves (m (lse - x_y) per shifted row);
 * cut [B,k] = first row of (b,c) that belongs to the answer (qlen - 1, the python slice start of train_utils.py:106);
 * Nb [B] = live answer rows (the answer text is the same under every context); doc_lp [B,k] = log p(context c | query b);
 * stats[0] = M = (live rows over all B k sequences) / k.
 *   out[0] = ( sum_b [ 1/k sum_c sum_{t < cut_bc} row_nll[b,c,t]
 *                      - sum_{j < Nb_b} logsumexp_c( doc_lp[b,c] - row_nll[b,c,cut_bc + j] ) ] ) / M
 * (RAG-token: every answer token is marginalised over the contexts; prompt rows, which differ per context, enter with
 * weight 1/k).  k = 1 with weights == NULL IS dalm_marg_ce_finalize (same kernel, same bits; cut may be NULL).
 * weights [B,k,Tg] (may be NULL) receives -d out[0] / d(log-prob of the label at that row): 1/(k M) before the cut,
 * softmax_c(...) / M on answer rows - the per-row factor a backward pass multiplies (softmax - onehot) with. */
int dalm_marg_ce_finalize_topk(const float* row_nll, int64_t B, int64_t k, int64_t Tg,
                               const int64_t* cut, const float* Nb, const float* doc_lp,
                               const float* stats, float* out, float* weights,
                               dalm_stream_t stream);

/* Scores of the k retrieved contexts of every query and their log-softmax over the k (RAG-token, p(c | q_b)):
 *   scores[b,c] = scale * q[b] . P[b,c],   doc_lp[b,c] = scores[b,c] - logsumexp_c' scores[b,c']      q [B,D], P [B,k,D]
 * and the closed-form backward of the k-context loss from the weights of dalm_marg_ce_finalize_topk:
 *   g[b,c] = -gscale[0] * sum_{j < Nb[b]} weights[b,c,cut[b,c]+j],  ds[b,c] = g[b,c] - exp(doc_lp[b,c]) sum_c' g[b,c'],
 *   dq[b] = scale sum_c ds[b,c] P[b,c],   dP[b,c] = scale ds[b,c] q[b]            (dq / dP / dscores may be NULL; k <= 64)
 * The reference marginalises over ONE context (train_utils.py:123-124; TODO at train_rage2e.py:461-462). */
int dalm_doc_scores_topk_fwd(const float* q, const float* P, int64_t B, int64_t k, int64_t D,
                             float scale, float* scores, float* doc_lp, dalm_stream_t stream);
int dalm_doc_scores_topk_bwd(const float* q, const float* P, int64_t B, int64_t k, int64_t D,
                             int64_t Tg, float scale, const float* doc_lp, const float* weights,
                             const int64_t* cut, const float* Nb, const float* gscale,
                             float* dq, float* dP, float* dscores, dalm_stream_t stream);

/* doc_lp[b] = S[b,b] - logsumexp_j S[b,j] on a materialised S
 * (train_utils.py:124), plus its backward
 *   dS[b,j] += coef[b] * ([j==b] - exp(S_bj - row_lse[b])). */
int dalm_doc_logprob_fwd(const float* S, int64_t n, int64_t ldS, float* doc_lp,
                         float* row_lse, dalm_stream_t stream);
int dalm_doc_logprob_bwd(const float* S, int64_t n, int64_t ldS,
                         const float* row_lse, const float* coef, float* dS,
                         int64_t lddS, int accumulate, dalm_stream_t stream);

/* get_nll (train_utils.py:91-93): out[r] = -lp[r, labels[r]] for R rows of V. */
int dalm_gather_nll(const float* lp, const int64_t* labels, int64_t R, int64_t V,
                    float* out, dalm_stream_t stream);
/* marginalize_log_probs (train_utils.py:96-110): out[t,:] = lp[t,:] + (t >= qlen-1 ? doc_lp[0] : 0). */
int dalm_marginalize_rows(const float* lp, int64_t T, int64_t V,
                          const float* doc_lp, int64_t qlen, float* out,
                          dalm_stream_t stream);
/* same with the length read from device memory (qlen_dev[0]): the reference's per-sample loop
 * (train_utils.py:127-129) hands over elements of a device tensor; no host synchronisation per sample. */
int dalm_marginalize_rows_dev(const float* lp, int64_t T, int64_t V,
                              const float* doc_lp, const int64_t* qlen_dev,
                              float* out, dalm_stream_t stream);

/* contrastive loss assembly (train_rage2e.py:443-446) from row/col statistics:
 *   out[0] = 0.5 ( sum_i (row_lse[i]-diag[i]) + sum_j (col_lse[j]-diag[j]) ) / n_global
 *   doc_lp[i] = diag[i] - row_lse[i]      (may be NULL)
 * n_local entries are summed (this rank's rows/columns). */
int dalm_contrastive_finalize(const float* row_lse, const float* col_lse,
                              const float* diag, int64_t n_local,
                              int64_t n_global, float* out, float* doc_lp,
                              dalm_stream_t stream);

/* ---- exact inner-product top-k (eval retrieval) ----
 * Replaces the hnswlib index of dalm/eval/utils.py:18-68 (construct_search_index / knn_query, space "ip") for
 * corpora that fit the GPU: out_val[m,k] / out_idx[m,k] = the k largest scale * Q[i] . C[j] per query row, sorted
 * descending (ties: lower corpus index first), exact f32, without materialising the m x n score matrix
 * (one pass of the streaming MFMA kernel leaves a maximum per 32 corpus columns; a per-row threshold then picks the
 * ~k groups that can hold top-k members and only those are re-evaluated).  n*D*4 < 2^31 per call
 * (search larger corpora block by block and merge).  *overflow (device int) is set non-zero when a row had more
 * than 8k+64 groups or scores >= its threshold (massive ties, or k > n/32): the caller must then fall back to a materialising search. */
/* 1 when (D, k) fits the fused search (k <= 1024 and padded D + 3*(8k+64) floats of LDS <= 60 KB), else 0: callers test
 * this BEFORE choosing the fused path; dalm_sim_topk itself rejects unsupported shapes before anything is enqueued. */
int dalm_sim_topk_supported(int64_t D, int64_t k);
size_t dalm_sim_topk_workspace_bytes(int64_t m, int64_t n, int64_t D, int64_t k);
int dalm_sim_topk(const float* Q, const float* C, int64_t m, int64_t n, int64_t D,
                  float scale, int64_t k, float* out_val, int64_t* out_idx,
                  int* overflow, void* ws, size_t ws_bytes, dalm_stream_t stream);

/* lm_head + log-sum-exp + label gather without the logits (forward / evaluation): replaces
 *   logits = lm_head(hidden)                         dalm/models/rag_e2e_base_model.py:104-106
 *   logsumexp over the vocabulary, gather of the label  dalm/training/utils/train_utils.py:113-131
 * for R rows.  hidden [R,K] and weight [V,K] are contiguous bf16 (K % 64 == 0, 16-byte aligned), labels [R] int64:
 * a negative label marks a row without loss (row_nll 0), a label >= V gives NaN (torch.gather would raise).
 * row_lse[r] = log sum_v exp(hidden[r] . weight[v]) (f32 accumulation on the bf16 matrix cores), row_nll[r] = row_lse[r] -
 * logit of the label.  One bf16 MFMA kernel (128 x 128 tiles reduced in registers) + a merge launch; the backward is not
 * provided.  Workspace: dalm_lm_head_lse_workspace_bytes(R, V) bytes. */
size_t dalm_lm_head_lse_workspace_bytes(int64_t R, int64_t V);
int dalm_lm_head_lse_fwd(const void* hidden, const void* weight, const int64_t* labels, int64_t R,
                         int64_t V, int64_t K, float* row_lse, float* row_nll, void* ws, size_t ws_bytes,
                         dalm_stream_t stream);

/* The whole loss assembly of the RAG-e2e step (train_rage2e.py:443-467) in one launch:
 *   out[1] = L_con (as dalm_contrastive_finalize), doc_lp[i] = diag[i] - row_lse[i] (may be NULL),
 *   out[2] = L_gen (as dalm_marg_ce_finalize with that doc_lp), out[0] = L_con + L_gen.
 * Nb, row_lse, col_lse, diag have n_local entries (this rank's rows). */
int dalm_rag_loss_finalize(const float* row_nll, int64_t num_rows, const float* Nb,
                           const float* row_lse, const float* col_lse,
                           const float* diag, int64_t n_local, int64_t n_global,
                           const float* stats, float* out, float* doc_lp,
                           dalm_stream_t stream);

/* ---- dalm_comm_*: the collectives of the sharded in-batch negatives, on RCCL, owned by the library ----
 * Stands in for what the reference gets from accelerate/DDP (train_rage2e.py:416-418,471), plus the embedding
 * all-gathers the reference does not have (it uses rank-local negatives).  One process per GPU.  RCCL is bound at
 * run time (dlopen: the copy already in the process, else /opt/rocm/lib/librccl.so).  Collectives are enqueued on
 * a side HIP stream owned by the communicator; order them against your own streams with
 *   dalm_comm_wait_stream(c, producer)   - the comm stream waits for work already queued on `producer`
 *   dalm_comm_stream_wait(c, consumer)   - `consumer` waits for the collectives already queued
 * (hipEvents; nothing blocks the host).  Bootstrap: rank 0 calls dalm_comm_unique_id and ships the 128 bytes to
 * the other ranks (file, store, env); every rank then calls dalm_comm_init.  Errors: negative = argument,
 * 999 = RCCL not loadable, 1000 + ncclResult_t, other positive = hipError_t. */
typedef struct dalm_comm dalm_comm_t;
int dalm_comm_unique_id(void* id128);
int dalm_comm_init(dalm_comm_t** out, const void* id128, int rank, int world, int device);
int dalm_comm_destroy(dalm_comm_t* c);
int dalm_comm_rank(const dalm_comm_t* c);
int dalm_comm_world(const dalm_comm_t* c);
int dalm_comm_wait_stream(dalm_comm_t* c, dalm_stream_t producer);
int dalm_comm_stream_wait(dalm_comm_t* c, dalm_stream_t consumer);
int dalm_comm_allgather(dalm_comm_t* c, const void* send, void* recv, size_t bytes_per_rank);
int dalm_comm_allreduce_sum_f32(dalm_comm_t* c, float* buf, size_t n);
/* The same collectives on a stream the CALLER names (its current stream, a side stream of its own, a capturing stream):
 * stream-ordered like a kernel launch, no library-owned stream or event involved - what dalm_amd.comm.NativeRcclComm
 * uses.  Safe to call from several host threads (one mutex per communicator around every enqueue / record+wait). */
int dalm_comm_allgather_on(dalm_comm_t* c, const void* send, void* recv, size_t bytes_per_rank, dalm_stream_t stream);
int dalm_comm_allreduce_sum_f32_on(dalm_comm_t* c, float* buf, size_t n, dalm_stream_t stream);

/* ---- nf4: 4-bit NormalFloat storage of the frozen base weights (`use_bnb`) ----------------------------------
 * Replaces what the reference delegates to bitsandbytes through BitsAndBytesConfig(load_in_4bit=True,
 * bnb_4bit_quant_type="nf4", bnb_4bit_compute_dtype=bfloat16): dalm/models/rag_e2e_base_model.py:137-142,
 * dalm/models/retriever_only_base_model.py:26,86 (quantise at load; dequantise to the compute dtype in front of every
 * matmul with a frozen weight).  Blocks of 64 consecutive elements of the flattened weight, one f32 absmax per
 * block, the 16 NF4 levels of QLoRA (appendix E), two indices per byte with element 2j in the high nibble.
 *   packed: dalm_nf4_packed_bytes(n) = ceil(n/2) bytes;  absmax: dalm_nf4_absmax_count(n) = ceil(n/64) floats.
 * `w` / `out`: n elements of `dtype` (DALM_F32 or DALM_BF16), 16-byte aligned.  Streaming kernels: quantise reads
 * 4n (2n) bytes, dequantise writes 4n (2n) bytes; 0.5625 bytes per weight stay resident. */
size_t dalm_nf4_packed_bytes(int64_t n);
size_t dalm_nf4_absmax_count(int64_t n);
int dalm_nf4_quantize(const void* w, int dtype, int64_t n, uint8_t* packed, float* absmax, dalm_stream_t stream);
int dalm_nf4_dequantize(const uint8_t* packed, const float* absmax, int64_t n, int dtype, void* out,
                        dalm_stream_t stream);

/* ---- generator-tower elementwise chains (Llama family) ---------------------------------------------------------
 * The reference runs the generator through transformers (dalm/models/rag_e2e_base_model.py:104-106 calls
 * `self.generator_model(...)`); inside, every decoder layer evaluates
 *   apply_rotary_pos_emb:  q*cos + rotate_half(q)*sin  for q and k   (modeling_llama.py, 8 eager launches + ~14 backward)
 *   LlamaMLP:              down(silu(gate(x)) * up(x))               (2 eager launches + 4 backward, one saved activation)
 * as chains of elementwise torch ops.  These entry points evaluate each chain in ONE streaming launch per direction and
 * round at exactly the points the eager chain (and autograd's backward of it) rounds, so results are the same values.
 *
 * dalm_rope_qk: q / k / outputs are [B, H, T, hd] views with element strides {b, h, t} and a contiguous last dimension;
 *   cos / sin are [B, T, hd] with element strides {b, t}.  backward != 0: q / k are the gradients of the outputs and the
 *   outputs receive the gradients of the inputs.  Vector path when hd/2 is a multiple of 16 bytes' worth of elements and all
 *   pointers / strides are 16-byte aligned; any even hd otherwise.
 * dalm_swiglu_fwd:  act = silu(gate) * up over n contiguous elements.
 * dalm_swiglu_bwd:  d_gate, d_up from d_act, gate, up (the activation is recomputed, nothing is saved by the forward). */
int dalm_rope_qk(const void* q, const void* k, void* q_out, void* k_out, const void* cos, const void* sin, int dtype,
                 int64_t B, int64_t T, int64_t Hq, int64_t Hk, int64_t hd, const int64_t* q_strides,
                 const int64_t* k_strides, const int64_t* qo_strides, const int64_t* ko_strides,
                 const int64_t* cs_strides, int backward, dalm_stream_t stream);
int dalm_swiglu_fwd(const void* gate, const void* up, void* act, int dtype, int64_t n, dalm_stream_t stream);
int dalm_swiglu_bwd(const void* d_act, const void* gate, const void* up, void* d_gate, void* d_up, int dtype, int64_t n,
                    dalm_stream_t stream);
/* The same two functions on [R, C] VIEWS with row strides (in elements; C and the strides multiples of 16 bytes): gate and up
 * as the two halves of the ONE [R, 2 C] output of x [W_gate | W_up]^T (frozen projections that share their input run as one
 * GEMM, dalm_amd/models/frozen_linear.py), d_gate / d_up written into the halves of one [R, 2 C] buffer.  Same arithmetic and
 * rounding points as the contiguous forms. */
int dalm_swiglu_fwd_2d(const void* gate, const void* up, void* act, int dtype, int64_t R, int64_t C, int64_t ld_gate,
                       int64_t ld_up, int64_t ld_act, dalm_stream_t stream);
int dalm_swiglu_bwd_2d(const void* d_act, const void* gate, const void* up, void* d_gate, void* d_up, int dtype, int64_t R,
                       int64_t C, int64_t ld_dact, int64_t ld_gate, int64_t ld_up, int64_t ld_dgate, int64_t ld_dup,
                       dalm_stream_t stream);
/* RMSNorm of a decoder layer (modeling_llama.py LlamaRMSNorm: w * (x * rsqrt(mean(x^2) + eps)).to(dtype)), optionally with the
 * residual add in front of it, one wave per row, [R, D] row-major, D a multiple of 16 bytes, at most 8192 (bf16) / 4096 (f32):
 *   fwd: delta != NULL: h_out = x + delta (rounded to dtype) and the norm is taken of h_out; delta == NULL (then h_out == NULL
 *        too): of x.  y [R, D], rstd [R] (f32, kept for the backward).
 *   bwd: dx = rstd * (g - xh * mean(g * xh)), g = dy * w, xh = h * rstd, plus dres (the gradient reaching h through the
 *        residual path) when dres != NULL.  The weight gradient is not produced (LoRA freezes the norm weights). */
int dalm_rms_norm_fwd(const void* x, const void* delta, const void* w, int dtype, int64_t R, int64_t D, float eps, void* h_out,
                      void* y, float* rstd, dalm_stream_t stream);
int dalm_rms_norm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, int dtype, int64_t R,
                      int64_t D, void* dx, dalm_stream_t stream);

/* Elementwise chains of a Falcon-7B decoder layer (transformers modeling_falcon.py FalconDecoderLayer.forward /
 * FalconMLP.forward / dropout_add, reached by the reference through self.generator_model(...),
 * dalm/models/rag_e2e_base_model.py:104-106; BASELINE.json config 5).  bf16 tensors, 16-byte aligned.
 *   layer_norm fwd: y = bf16((x - mean) * rstd * w + b) with f32 statistics of the bf16 row - torch's autocast LayerNorm (f32)
 *        followed by its consumers' casts to bf16; b may be NULL; mean / rstd [R] f32 are kept for the backward.
 *        [R, D] row-major, D a multiple of 8, at most 8192.
 *   layer_norm bwd: dx = rstd * (g - mean(g) - xh * mean(g * xh)) (+ dres), g = dy * w, xh = (x - mean) * rstd.  Weight / bias
 *        gradients are not produced (LoRA freezes them).
 *   gelu: exact erf form, f32 arithmetic, one rounding (torch.nn.GELU() on bf16); bwd: dy * (Phi(x) + x phi(x)).
 *   add3: out = bf16(c + bf16(a + b))  (mlp_output += attention_output; then residual + that).  n a multiple of 8. */
int dalm_layer_norm_fwd(const void* x, const void* w, const void* b, int64_t R, int64_t D, float eps, void* y, float* mean,
                        float* rstd, dalm_stream_t stream);
int dalm_layer_norm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, const void* dres,
                        int64_t R, int64_t D, void* dx, dalm_stream_t stream);
int dalm_gelu_fwd(const void* x, void* y, int64_t n, dalm_stream_t stream);
int dalm_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, dalm_stream_t stream);
int dalm_add3(const void* a, const void* b, const void* c, void* out, int64_t n, dalm_stream_t stream);

/* BERT encoder layer: dropout + residual add + LayerNorm of BertSelfOutput / BertOutput (transformers modeling_bert.py; the
 * reference reaches them through self.retriever_model(...), dalm/models/rag_e2e_base_model.py:84-93) as one launch per direction,
 * rounding where the eager chain rounds under bf16 autocast (dalm_amd/csrc/bert.hip):
 *   fwd: d = bf16(a keep / (1 - p)); s = f32(d) + res; y32 = (s - mean) rstd w + b; y16 = bf16(y32).  a [R, D] bf16 (the dense
 *        output), res [R, D] f32, w / b [D] f32 (w_bf16 == 0) or bf16; D a multiple of 8, at most 2048.  mean / rstd [R] f32 and
 *        keep_bits [R][D / 8] (bit e of byte c = element 8 c + e survives; NULL when dropout_p == 0) are kept for the backward.
 *        The keep mask is regenerated from (the 64-bit word at `seed` in DEVICE memory, salt, flat element index):
 *        oracle/lora_mask.py::keep_mask_v2 restates it.
 *   bwd: dy = g32 + f32(g16) (either may be NULL); ds = LayerNorm backward (f32); d_res = ds; d_a = bf16(bf16(ds) keep / (1 - p)).
 *        The weight / bias gradients are not produced (frozen under LoRA; trainable LayerNorms keep transformers' code). */
int dalm_bert_add_norm_fwd(const void* a, const float* res, const void* w, const void* b, int w_bf16, int64_t R, int64_t D,
                           float eps, float dropout_p, const void* seed, uint32_t salt, float* y32, void* y16, uint8_t* keep_bits,
                           float* mean, float* rstd, dalm_stream_t stream);
int dalm_bert_add_norm_bwd(const float* g32, const void* g16, const void* a, const float* res, const void* w, int w_bf16,
                           const uint8_t* keep_bits, const float* mean, const float* rstd, int64_t R, int64_t D, float dropout_p,
                           float* d_res, void* d_a, dalm_stream_t stream);

/* Backward of scaled-dot-product attention, bf16, head width 128, boolean mask (dalm_amd/csrc/attn.hip).  Stands in for the
 * backward of torch.nn.functional.scaled_dot_product_attention as transformers' sdpa_attention_forward calls it inside
 * self.generator_model(...) (dalm/models/rag_e2e_base_model.py:104-106; loss.backward(), train_rage2e.py:466).
 *   dalm_attn_mask_bits: mask [B, 1, T, T] bytes (non-zero = attend; element strides mask_stride_b / mask_stride_row, last
 *        dimension contiguous; NULL = no mask) AND the causal flag -> bits_rows, bits_cols [B][32 W][W] u32 (W = ceil(T / 32);
 *        bit c of word w of row i = mask[i][32 w + c]; columns likewise over rows) and live [B][W][W] bytes (32 x 32 tile not
 *        empty).  Once per mask: every layer and head reads the same words.
 *   dalm_attn_bwd: q, k, v, o (the forward's output), d_o, and lse [B, H, T] f32 (natural log of the row sums, what torch's
 *        memory-efficient forward returns) -> dq, dk, dv.  strides: 8 x (batch, head, row) ELEMENT strides of
 *        q, k, v, o, d_o, dq, dk, dv (last dimension contiguous, multiples of 8); delta: [B, H, T] f32 scratch
 *        (D = rowsum(dO o O)).  P and dS are rounded to bf16 for their products, sums in f32.  T x (row stride) of q, k, v, d_o
 *        must stay below 2^30 elements (the streamed blocks are addressed with 32-bit offsets from the sequence's first row).
 *        Two forms with the same arithmetic and bit-identical outputs (register-staged tiles / LDS-DMA stages + transpose reads,
 *        the default); DALM_ATTN_FWD=1 / DALM_ATTN_DKDV=1 select the first, read once per process (tools/attn_ab.py).
 *        cos / sin (NULL, or [B or 1, T, hd] bf16 with element strides cs_stride_b (0 for one table) / cs_stride_t): q and k
 *        are the outputs of dalm_rope_qk and dq / dk leave as the gradients of its INPUTS - that kernel's backward applied
 *        in the epilogue, same rounding points.
 *   dropout_p > 0 (BERT's attention_probs_dropout_prob; T even): P o M / (1 - p) in front of P V, the keep mask M regenerated in
 *        every kernel from (the 64-bit word at `seed` in DEVICE memory, salt, element index ((b H + h) T + i) T + j) - never
 *        stored; pass the same three values to dalm_attn_fwd and dalm_attn_bwd.  oracle/attn_dropout.py restates the mask. */
/*   dalm_attn_fwd: o = softmax(scale q k^T + mask) v and lse [B, H, T] f32 (natural log; 0 for rows without a live key, whose
 *        output is 0).  strides: 4 x (batch, head, row) element strides of q, k, v, o. */
int dalm_attn_fwd(const void* q, const void* k, const void* v, const uint32_t* bits_rows, const uint8_t* live, int64_t B, int64_t H,
                  int64_t T, int64_t hd, float scale, const int64_t* strides, float dropout_p, const void* seed, uint32_t salt,
                  void* o, float* lse, dalm_stream_t stream);
int dalm_attn_mask_bits(const void* mask, int64_t B, int64_t T, int64_t mask_stride_b, int64_t mask_stride_row, int causal,
                        uint32_t* bits_rows, uint32_t* bits_cols, uint8_t* live, dalm_stream_t stream);
int dalm_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                  const uint32_t* bits_rows, const uint32_t* bits_cols, const uint8_t* live, int64_t B, int64_t H, int64_t T,
                  int64_t hd, float scale, const int64_t* strides, const void* cos, const void* sin, int64_t cs_stride_b,
                  int64_t cs_stride_t, float dropout_p, const void* seed, uint32_t salt, void* dq, void* dk, void* dv,
                  float* delta, dalm_stream_t stream);

/* PACKED (un-padded) forms of the three attention entry points: the towers run on the LIVE tokens of a batch only
 * (dalm_amd/packed.py: padding tokens contribute exactly zero to the reference's loss and gradients, train_utils.py:134-136 and
 * rag_e2e_base_model.py:108-111, so every row-wise kernel and GEMM of the towers skips them).  q, k, v, o, ... are
 * [n_tokens, H, hd] tensors; sequence b owns token rows cu_seqlens[b] .. cu_seqlens[b + 1] - 1 (int32 [B + 1], device memory),
 * T = the longest sequence (<= 2048).  strides keep the (batch, head, row) triples, the batch entries are ignored; cos / sin are
 * [n_tokens, hd] tables (row stride cs_stride_t) holding each token's ORIGINAL position.  lse / delta are [B, H, T] f32 and the
 * mask words [B][32 W][W] as above, built by
 *   dalm_attn_mask_bits_packed: key_live [n_tokens] bytes (NULL = all live; 0 = a token that only queries, e.g. the padding
 *        position in front of a left-padded sequence whose row predicts the first real token), causal flag: element (i, j) of
 *        sequence b is live when token j is a live key and (causal) j <= i - what the padded mask says about the same tokens. */
int dalm_attn_mask_bits_packed(const uint8_t* key_live, const int32_t* cu_seqlens, int64_t B, int64_t T, int causal,
                               uint32_t* bits_rows, uint32_t* bits_cols, uint8_t* live, dalm_stream_t stream);
int dalm_attn_fwd_packed(const void* q, const void* k, const void* v, const uint32_t* bits_rows, const uint8_t* live,
                         const int32_t* cu_seqlens, int64_t B, int64_t H, int64_t T, int64_t hd, float scale,
                         const int64_t* strides, float dropout_p, const void* seed, uint32_t salt, void* o, float* lse,
                         dalm_stream_t stream);
int dalm_attn_bwd_packed(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                         const uint32_t* bits_rows, const uint32_t* bits_cols, const uint8_t* live, const int32_t* cu_seqlens,
                         int64_t B, int64_t H, int64_t T, int64_t hd, float scale, const int64_t* strides, const void* cos,
                         const void* sin, int64_t cs_stride_t, float dropout_p, const void* seed, uint32_t salt, void* dq, void* dk,
                         void* dv, float* delta, dalm_stream_t stream);

/* ---- the low-rank branch of a LoRA-wrapped Linear ----------------------------------------------------------------
 * The reference wraps q_proj / v_proj (key / query / value for BERT retrievers) in peft LoRA adapters, r = 8, alpha = 16,
 * dropout 0.05 (dalm/models/rag_e2e_base_model.py:145-160, retriever_only_base_model.py:92-107); peft evaluates
 *   out = W x + s * B(A(dropout(x)))                                  (peft/tuners/lora/layer.py, Linear.forward)
 * as eager ops and autograd differentiates it op by op.  With r <= 16 every tensor of the branch is [rows, r] or streams a
 * [rows, K] activation once; the three entry points below are the branch and its backward as HBM-bound kernels:
 *   dalm_lora_rowdot :  out[row, j]  = scale * sum_k m x[row, k] W(j, k)       z = dropout(x) A^T / (1-p);  dz = s g B
 *   dalm_lora_rankupd:  y[row, c]   += scale * m * sum_j z[row, j] W(j, c)     out += s z B^T;  dx += m (dz A) / (1-p)
 *   dalm_lora_colacc :  out(j, c)    = scale * sum_row m x[row, c] z[row, j]   dB = s g^T z;  dA = dz^T (m x) / (1-p)
 * x / y: [R, K|C] row-major, `dtype` DALM_F32 or DALM_BF16, 16-byte aligned, column count a multiple of 8.  W, z, out: f32.
 *   rowdot : w_kmajor != 0 reads W as [rank][K] (lora_A.weight), 0 as [K][rank] (lora_B.weight).
 *   rankupd: w_cmajor != 0 reads W as [C][rank] (lora_B.weight), 0 as [rank][C] (lora_A.weight).
 *   colacc : out_jmajor != 0 writes [rank][C] (the layout of lora_A.weight), 0 writes [C][rank] (lora_B.weight).
 * rank: 8 or 16.  m: dropout keep mask with drop probability p (p = 0: no mask), never stored - a counter-based hash of
 * (the 64-bit word at `seed`, a DEVICE pointer read when the kernel runs, NULL = 0; `salt`; the flat element index
 * row * columns + column), so the three kernels regenerate the same mask for the same (seed, salt, shape).  1 / (1 - p) is the
 * caller's business (fold it into `scale`).  All sums run in fixed orders: results are bit-reproducible. */
int dalm_lora_rowdot(const void* x, int dtype, const float* W, int w_kmajor, int64_t R, int64_t K, int rank, float scale,
                     float p, const void* seed, uint32_t salt, float* out, dalm_stream_t stream);
int dalm_lora_rankupd(void* y, int dtype, const float* z, const float* W, int w_cmajor, int64_t R, int64_t C, int rank,
                      float scale, float p, const void* seed, uint32_t salt, dalm_stream_t stream);
size_t dalm_lora_colacc_workspace_bytes(int64_t R, int64_t C, int rank);
int dalm_lora_colacc(const void* x, int dtype, const float* z, int64_t R, int64_t C, int rank, float scale, float p,
                     const void* seed, uint32_t salt, float* out, int out_jmajor, void* ws, size_t ws_bytes,
                     dalm_stream_t stream);

/* ---- round 5: the same branch for projections that SHARE their input, bf16 activations, mask stored as bits ------------
 * (q_proj / v_proj of one attention block read the same normed hidden states, query / key / value of a BERT block likewise:
 * dalm/models/rag_e2e_base_model.py:61-80.)  `mode` selects how the two operand slots are used:
 *   1  one problem (slot 0);
 *   2  two terms over ONE activation (slot 0's x / y): rowdot reads x once for z_0 and z_1, colacc reads x once for out_0 and
 *      out_1, rankupd adds both rank updates in one pass over y.  rank 8 only;
 *   3  two independent problems of one shape in one launch (slots 0 and 1).
 * Every W is [rank][K] row-major f32 (lora_A as stored; lora_B transposed - models/lora.py keeps it that way in memory), z and
 * out f32, x / y bf16 row-major, 16-byte aligned, columns a multiple of 8 (rowdot: of 32).
 * Dropout (mask v2, restated in oracle/lora_mask.py::keep_mask_v2): dalm_lora2_rowdot computes the keep mask of slot t from
 * (the 64-bit word at `seed`, salt_t, flat element index) ONCE and writes it to bits_t, [R][K / 8] bytes, bit e of byte
 * (row, c) = element 8 c + e of the row survives; rankupd / colacc take bits_t (NULL = no mask) - the backward never depends on
 * the seed word again.  1 / (1 - p) is folded into `scale` by the caller.
 *   dalm_lora2_rowdot :  out_t[row][j]  = scale * sum_k m_t x_t[row][k] W_t[j][k]
 *   dalm_lora2_rankupd:  y[row][c]     += scale * sum_t m_t sum_j z_t[row][j] W_t[j][c]        (mode 3: y_t, one term each)
 *   dalm_lora2_colacc :  out_t[j][c]    = scale * sum_row m_t x_t[row][c] z_t[row][j]          ([rank][C], finished in the launch)
 * colacc: `ws` >= dalm_lora2_colacc_workspace_bytes, 8-byte aligned; `tickets`: dalm_lora2_colacc_ticket_words 32-bit words,
 * ZERO on entry and left zero (calls that share the buffer must be ordered on one stream).  Fixed summation orders. */
int dalm_lora2_rowdot(const void* x0, const void* x1, const float* W0, const float* W1, float* out0, float* out1, void* bits0,
                      void* bits1, int64_t R, int64_t K, int rank, float scale, float p, const void* seed, uint32_t salt0,
                      uint32_t salt1, int mode, dalm_stream_t stream);
int dalm_lora2_rankupd(void* y0, void* y1, const float* z0, const float* z1, const float* W0, const float* W1,
                       const void* bits0, const void* bits1, int64_t R, int64_t C, int rank, float scale, int mode,
                       dalm_stream_t stream);
size_t dalm_lora2_colacc_workspace_bytes(int64_t R, int64_t C, int rank, int mode);
size_t dalm_lora2_colacc_ticket_words(int64_t C, int mode);
int dalm_lora2_colacc(const void* x0, const void* x1, const float* z0, const float* z1, const void* bits0, const void* bits1,
                      float* out0, float* out1, int64_t R, int64_t C, int rank, float scale, int mode, void* ws,
                      size_t ws_bytes, uint32_t* tickets, dalm_stream_t stream);

/* ---- round 5: backward of the fused lm_head + marginalised CE ---------------------------------------
 * Replaces, for a FROZEN bias-free head,   logits = lm_head(hidden); loss(logits).backward()
 *   (dalm/models/rag_e2e_base_model.py:104-106 -> dalm/training/utils/train_utils.py:113-138)
 * without the [R, V] logits or their gradient ever existing: with row_lse from dalm_lm_head_lse_fwd, per vocabulary chunk
 *   dalm_lm_head_dlogits :  dl[r][c] = bf16( coef[r] * (exp(hidden[r] . W[col_base + c] - row_lse[r]) - [labels[r] == col_base + c]) )
 *                           for c < Vc, zero for Vc <= c < pitch  (the logits tile is recomputed on the bf16 matrix cores)
 *   dalm_transpose_bf16  :  wt[k][c] = W[col_base + c][k]   (zero for c >= Vc)        [K][pitch]
 *   dalm_lm_head_dhidden :  dh[r][k] (+)= sum_c dl[r][c] * wt[k][c]                    f32 [R][K]
 * hidden [R, K] and weight_chunk (= W + col_base * K, Vc rows) bf16 row-major, K % 64 == 0; labels: global vocabulary ids
 * (anything outside [0, V) never matches); coef[r] = d loss / d log p(label_r) magnitude (mask_r / M; 0 for rows without
 * loss); pitch: a multiple of 64 in [Vc, round_up(Vc, 256)].  dalm_f32_to_bf16 rounds the finished f32 gradient.
 * Fixed summation order (chunks in call order, K tiles in order inside a launch): bit-reproducible. */
int dalm_lm_head_dlogits(const void* hidden, const void* weight_chunk, const int64_t* labels, const float* row_lse,
                         const float* coef, int64_t R, int64_t Vc, int64_t K, int64_t col_base, void* dl, int64_t pitch,
                         dalm_stream_t stream);
/* round 6, the two-contraction form of the training step (no recomputation): logits [R, pitch] bf16 = hidden [R, K] . weight [V, K]^T
 * through the same bf16 MFMA main loop (columns V .. pitch - 1 are not written).  A row chunk of logits (<= ~128 MB) stays in the
 * Infinity Cache while dalm_marg_ce_fwd turns it into d(logits) in place and dalm_lm_head_dhidden (Vp = pitch = V, wt = the
 * head's transposed copy [K, V]) contracts it. */
int dalm_lm_head_logits(const void* hidden, const void* weight, int64_t R, int64_t V, int64_t K, void* logits, int64_t pitch,
                        dalm_stream_t stream);
int dalm_lm_head_dhidden(const void* dl, const void* wt, int64_t R, int64_t Vp, int64_t K, float* dh, int accumulate,
                         dalm_stream_t stream);
int dalm_transpose_bf16(const void* src, int64_t rows, int64_t cols, int64_t ld_src, void* dst, int64_t ld_dst,
                        dalm_stream_t stream);
int dalm_f32_to_bf16(const float* src, void* dst, int64_t n, dalm_stream_t stream);

/* ---- round 5: the similarity BACKWARD on the bf16 matrix cores at f32 accuracy --------------------------------------------------
 * Same result as dalm_sim_grad (dA = scale * dS . B, dS rebuilt from S = scale * A . B^T and the row / column statistics; the
 * autograd of dalm/training/utils/train_utils.py:76-88 + the doc term :121-124), computed as two bf16x3 contractions (three bf16
 * thirds per operand, six significant products, f32 accumulation) through the lm_head core instead of the f32 MFMA pipe:
 * dA within ~2e-6 of fp64 relative to its largest entry.  D % 64 == 0, operand images below 4 GB
 * (dalm_sim_grad_bf16x3_supported); workspace: dalm_sim_grad_bf16x3_workspace_bytes.  Fixed summation order. */
int dalm_sim_grad_bf16x3_supported(int64_t m, int64_t n, int64_t D);
size_t dalm_sim_grad_bf16x3_workspace_bytes(int64_t m, int64_t n, int64_t D);
int dalm_sim_grad_bf16x3(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D, float scale, int64_t diag_offset,
                         const float* row_coef, const float* row_lse, const float* col_coef, const float* col_lse, float* dA,
                         void* ws, size_t ws_bytes, dalm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DALM_HIP_H */
