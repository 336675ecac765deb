/* libktup_hip.so -- C ABI of the MI355X (gfx950) joint KG + recommender scoring engine.
 *
 * The reference (TaoMiner/joint-kg-recommender) has no FFI layer: its hot path is the torch-op
 * compositions inside jTransUP/models/{bprmf,transE,transH,transR,transUP,jTransUP}.py,
 * jTransUP/utils/loss.py and the ranking walk in jTransUP/utils/misc.py.  Each entry point below
 * replaces one of those compositions and cites it (paths relative to the reference checkout).
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless stated; tables are row-major fp32 with a row pitch
 *     `ld*` given in ELEMENTS; index arrays are int64 (the reference's LongTensor) unless stated;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous and stream-ordered;
 *   - nothing is allocated or freed: outputs and scratch are caller-owned (see *_workspace_bytes);
 *   - gradient outputs are ACCUMULATED with atomics (caller zero-fills, like a dense .grad);
 *   - return 0 on success, <0 on error; ktup_last_error() gives the thread-local message;
 *   - callable from any host thread: the only shared state is the set of integer options below (ktup_set_option), which
 *     choose between kernels computing the same results and are never read from the environment on the launch path.
 */
#ifndef KTUP_HIP_H
#define KTUP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KTUP_OK 0
#define KTUP_ERR_INVALID_ARG (-1)
#define KTUP_ERR_LAUNCH (-2)
#define KTUP_ERR_UNSUPPORTED (-3)

/* Gumbel modes for the preference gate (transUP.py:143-170, jTransUP.py:288-315). */
#define KTUP_GUMBEL_OFF 0     /* soft: raw logits are the mixture weights (transUP.py:108-113)      */
#define KTUP_GUMBEL_INPUT 1   /* hard, uniforms supplied by the caller (parity mode)                 */
#define KTUP_GUMBEL_PHILOX 2  /* hard, uniforms drawn on device from Philox4x32-10(seed, offset)     */
#define KTUP_GUMBEL_PHILOX_DEV 3 /* as PHILOX, but (seed, offset) are read from device memory when the kernel
                                  * runs: `uniform` points at a uint64_t[2] = {seed, offset}; the `seed` /
                                  * `offset` arguments are ignored.  For launches captured in a HIP graph. */

int ktup_version(void);
const char* ktup_last_error(void);

/* Process-wide integer options for tests and A/B measurements; each is seeded once, at library load, from the environment
 * variable in brackets.  They select between kernels that return the same results.
 *   "pref_mc"     [KTUP_PREF_MC, 1]      0: the generic K5-K7 kernels even for shapes the matrix-core kernels cover
 *   "eval_mc"     [KTUP_EVAL_MC, 1]      0: VALU all-candidate kernels instead of the matrix-core ones
 *   "rank_chunk"  [KTUP_RANK_CHUNK, 0]   > 0: force the chunked ranking kernels with this chunk size
 *   "seg_bwd_min" [KTUP_SEG_BWD_MIN, 8192]  batch size from which the *_bwd entry points given a workspace reduce row
 *                                        gradients by sorted segments instead of atomics (0: never)
 *   "bwd_wide_max" [KTUP_BWD_WIDE_MAX, 4096]  K5-K7 backward, d <= 128: pairs up to which four waves share a 16-pair tile
 *   "side_sort"   [KTUP_SIDE_SORT, 1]    0: the id sorts of those segment reductions stay on the caller's stream (default: a
 *                                        library-owned side stream, forked at entry and joined before the reduction; never
 *                                        while the caller's stream is being captured into a graph)
 *   "deterministic" [KTUP_DETERMINISTIC, 0]  1: ktup_train_rec_step / ktup_train_kg_step (gradients by atomics) run on ONE
 *                                        workgroup, so every gradient cell receives its adds from one wave in program order
 *                                        and two runs of a step give the same bits (sums of float atomics issued by several
 *                                        workgroups depend on the order they land in).  For parity runs: ~50x slower.        */
int ktup_set_option(const char* name, int value);
int ktup_get_option(const char* name, int* value);

/* ------------------------------------------------------------------ K1  BPRMF  bprmf.py:46-49 */
int ktup_score_bprmf_fwd(const float* U, int64_t ldu, const float* I, int64_t ldi, int d,
                         const int64_t* u_ids, const int64_t* i_ids, int64_t n, float* score, void* stream);
int ktup_score_bprmf_bwd(const float* U, int64_t ldu, const float* I, int64_t ldi, int d,
                         const int64_t* u_ids, const int64_t* i_ids, int64_t n, const float* gscore,
                         float* gU, float* gI, void* stream);

/* ------------------------------------------------------------------ K2  TransE  transE.py:51-63 */
/* n_rel = rows of R (0 if unknown): large batches with a small relation table take the wave-tile forward.     */
int ktup_score_transe_fwd(const float* E, int64_t lde, const float* R, int64_t ldr, int64_t n_rel, int d,
                          const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                          float* score, void* stream);
int ktup_score_transe_bwd(const float* E, int64_t lde, const float* R, int64_t ldr, int d,
                          const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                          const float* gscore, float* gE, float* gR, void* stream);

/* ------------------------------------------- K3  TransH  transH.py:58-71 + utils/misc.py:18-19
 * (also the KG branch of KTUP, jTransUP.py:144-157, on its ent/rel/norm tables)                  */
/* n_rel = rows of R and Nrm (relation_total); small relation tables are staged in LDS and only the entity rows
 * are gathered (wave-tile kernel for large batches).  Pass 0 if unknown: relation rows are gathered per triple. */
int ktup_score_transh_fwd(const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                          int64_t n_rel, int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n,
                          int l1, float* score, void* stream);
int ktup_score_transh_bwd(const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                          int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                          const float* gscore, float* gE, float* gR, float* gN, void* stream);

/* ------------------------------------------- K4  TransR  transR.py:65-78 + utils/misc.py:21-26
 * M is the (n_rel x d*d) projection table, row r reshaped (d_rel=d) x (d_ent=d), row-major.       */
/* Forward with scratch `ws` of ktup_score_transr_workspace_bytes(n, n_rel) bytes (4-byte aligned): the batch is bucketed
 * by relation and each M_r is staged once per workgroup for a matrix-core projection (d in {64,100,128}, n_rel <= 4096);
 * ws == NULL (or another shape) gathers M_r per triple like the reference does.                                      */
size_t ktup_score_transr_workspace_bytes(int64_t n, int64_t n_rel);
int ktup_score_transr_fwd(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm,
                          int64_t n_rel, int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n,
                          int l1, float* score, void* ws, void* stream);
int ktup_score_transr_bwd(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm,
                          int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                          const float* gscore, float* gE, float* gR, float* gM, void* stream);
/* Backward with scratch `ws` of ktup_score_transr_bwd_workspace_bytes(n, d, n_ent, n_rel) bytes (256-byte aligned; 0 = no
 * scratch route for this shape): the batch is bucketed by relation like the forward, gq = M_r^T gy and gM_r += gy (x) (h - t)
 * run on the matrix cores (d in {64,100,128}, n_rel <= 4096, ldm >= d*d) with gM_r / gR accumulated in registers per relation
 * run, and from n >= option seg_bwd_min the entity-row gradients are summed by sorted segments instead of float atomics.
 * ws == NULL or another shape: exactly ktup_score_transr_bwd (M_r streamed and d*d atomics per triple).                 */
size_t ktup_score_transr_bwd_workspace_bytes(int64_t n, int d, int64_t n_ent, int64_t n_rel);
int ktup_score_transr_bwd_ws(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm,
                             int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                             const float* gscore, float* gE, float* gR, float* gM, int64_t n_ent, int64_t n_rel, void* ws,
                             void* stream);

/* K1-K3 backward with caller scratch, for large batches: from n >= option seg_bwd_min (default 8192) every row's gradient vector
 * is written with plain stores and summed per table row by sorted segments (ktup_segment_reduce_rows, below) -- float atomics
 * serialise when hundreds of rows of a batch share a table row -- and the relation-side tables accumulate in LDS.  Otherwise
 * identical to the *_bwd entry points above (`ws` unused; *_workspace_bytes returns 0).  n_ent / n_users / n_items / n_rel = rows of
 * the tables (key ranges of the counting sorts, size of the LDS accumulators).                                              */
size_t ktup_score_kg_bwd_workspace_bytes(int64_t n, int d, int64_t n_ent);
int ktup_score_transe_bwd_ws(const float* E, int64_t lde, const float* R, int64_t ldr, int d, const int64_t* h,
                             const int64_t* t, const int64_t* r, int64_t n, int l1, const float* gscore, float* gE,
                             float* gR, int64_t n_ent, int64_t n_rel, void* ws, void* stream);
int ktup_score_transh_bwd_ws(const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                             int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                             const float* gscore, float* gE, float* gR, float* gN, int64_t n_ent, int64_t n_rel, void* ws,
                             void* stream);
size_t ktup_score_bprmf_bwd_workspace_bytes(int64_t n, int d, int64_t n_users, int64_t n_items);
int ktup_score_bprmf_bwd_ws(const float* U, int64_t ldu, const float* I, int64_t ldi, int d, const int64_t* u_ids,
                            const int64_t* i_ids, int64_t n, const float* gscore, float* gU, float* gI, int64_t n_users,
                            int64_t n_items, void* ws, void* stream);

/* ------------------------------------------- K5/K6/K7  TUP and KTUP preference-gated translation
 * transUP.py:69-82,105-170 ; jTransUP.py:122-143,250-315.
 *
 * ktup_pref_prepare mixes and pre-scales the (tiny) preference tables once per table version into
 * caller scratch `ws` (ktup_pref_workspace_bytes): A = pref (+ rel), C = pref_norm (+ norm);
 * it stores A/2 (logit table), beta*A and beta*C with beta = 1 for TUP (rel == NULL) and 1/2 for
 * KTUP (jTransUP.py:253,257,258), zero-padded so the score kernels run without bounds checks.
 * d: any positive multiple of 4 (models/base.py:52 takes any integer; callers stage other widths with a zero tail).  Up to 256
 * columns the tile kernels run; beyond, the one-wave-per-pair kernels (ktup_score_pref_row.hip, n_pref <= 128) behind the same
 * entry points: ktup_score_{tup,ktup}_{fwd,bwd,bwd_ws} and ktup_eval_pref_scores.  0 bytes = (d, n_pref) not covered.      */
size_t ktup_pref_workspace_bytes(int d, int n_pref);
int ktup_pref_prepare(const float* pref, const float* pref_norm, const float* rel, const float* norm, int64_t ld,
                      int n_pref, int d, float* ws, void* stream);

/* TUP: E == NULL and item2ent == NULL.  KTUP: ie = I[i] + E[item2ent[i]] (int32 table; the pad row
 * of E is all-zero, jTransUP.py:46,100).  `uniform` is (n x n_pref) for KTUP_GUMBEL_INPUT.         */
int ktup_score_tup_fwd(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* pref_ws, int n_pref,
                       int d, const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                       const float* uniform, uint64_t seed, uint64_t offset, float* score, void* stream);
int ktup_score_ktup_fwd(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                        const int32_t* item2ent, const float* pref_ws, int n_pref, int d, const int64_t* u_ids,
                        const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform,
                        uint64_t seed, uint64_t offset, float* score, void* stream);

/* Backward: gU/gI/gE are dense table grads (atomics); gA and gC are (n_pref x d, pitch d) grads of the
 * MIXED tables A = pref(+rel), C = pref_norm(+norm): the caller adds gA to pref.grad (and rel.grad),
 * gC to pref_norm.grad (and norm.grad).  ent_pad >= 0 names the E row that never receives a gradient
 * (nn.Embedding(padding_idx=...), jTransUP.py:96).                                                  */
int ktup_score_tup_bwd(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* pref_ws, int n_pref,
                       int d, const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                       const float* uniform, uint64_t seed, uint64_t offset, const float* gscore, float* gU,
                       float* gI, float* gA, float* gC, void* stream);
int ktup_score_ktup_bwd(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                        const int32_t* item2ent, int64_t ent_pad, const float* pref_ws, int n_pref, int d,
                        const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                        const float* uniform, uint64_t seed, uint64_t offset, const float* gscore, float* gU,
                        float* gI, float* gE, float* gA, float* gC, void* stream);

/* Backward with caller scratch, for large batches and hot rows: from n >= option seg_bwd_min (default 8192) and
 * d in {64, 100, 128, 256} the kernel writes each pair's row gradients with plain stores (2 x n x d floats of `ws`) and they are
 * summed per table row by sorted segments (ktup_segment_reduce_rows below) -- (segments + chunks) x d atomics instead of
 * 3 n d, and no serialisation on rows that many pairs share.  Otherwise identical to ktup_score_{tup,ktup}_bwd (ws unused).
 * n_user_rows / n_item_rows = rows of U / I.  ktup_score_pref_bwd_workspace_bytes returns 0 when the atomics path applies. */
size_t ktup_score_pref_bwd_workspace_bytes(int64_t n, int d, int64_t n_user_rows, int64_t n_item_rows);
int ktup_score_tup_bwd_ws(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* pref_ws, int n_pref,
                          int d, const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                          const float* uniform, uint64_t seed, uint64_t offset, const float* gscore, float* gU,
                          float* gI, float* gA, float* gC, int64_t n_user_rows, int64_t n_item_rows, void* ws, void* stream);
int ktup_score_ktup_bwd_ws(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                           const int32_t* item2ent, int64_t ent_pad, const float* pref_ws, int n_pref, int d,
                           const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                           const float* uniform, uint64_t seed, uint64_t offset, const float* gscore, float* gU,
                           float* gI, float* gE, float* gA, float* gC, int64_t n_user_rows, int64_t n_item_rows, void* ws,
                           void* stream);

/* ------------------------------------------- row-gradient reduction by sorted segments (new; replaces n x d float atomics)
 * gT[ids[e], :] += sign(e) * G[e mod n_src, :]   for e in [0, m), sign = +1 for e < sign_split, -1 otherwise; m = n_src
 * (one id per source row) or 2 n_src (two roles per source row, e.g. TransE: ids = h ++ t, sign_split = n_src:
 * gE[h] += gz, gE[t] -= gz, transE.py:51-63 differentiated).  The ids are counting-sorted over the key range [0, n_rows)
 * and consecutive entries of a row are summed in registers; if map2 != NULL every flushed row sum is also added to
 * gT2[map2[row], :] unless map2[row] == pad2 (KTUP: gE[item2ent[i]] gets what gI[i] gets, jTransUP.py:122-130).
 * d % 4 == 0, 16-byte aligned rows, m and n_rows < 2^31; `ws`: ktup_segment_workspace_bytes(m, n_rows) bytes, 16-byte aligned. */
size_t ktup_segment_workspace_bytes(int64_t m, int64_t n_rows);
int ktup_segment_reduce_rows(const float* G, int64_t ldg, int d, int64_t n_src, const int64_t* ids, int64_t m,
                             int64_t sign_split, int64_t n_rows, float* gT, int64_t ldt, const int32_t* map2, int64_t pad2,
                             float* gT2, int64_t ldt2, void* ws, void* stream);

/* ------------------------------------------- K8/K9  pairwise losses  utils/loss.py:8-16,29-31
 * bpr   : loss = mean(-logsigmoid(target * (pos - neg)))   target = +1 (bprmf) / -1 (translation models,
 *         utils/trainer.py:15-17);   margin: loss = SUM max(pos - neg + margin, 0).
 * `loss` is one device float (overwritten); `gloss` is the upstream gradient as a DEVICE scalar.        */
int ktup_loss_bpr_fwd(const float* pos, const float* neg, int64_t n, float target, float* loss, void* stream);
int ktup_loss_bpr_bwd(const float* pos, const float* neg, int64_t n, float target, const float* gloss,
                      float* gpos, float* gneg, void* stream);
int ktup_loss_margin_fwd(const float* pos, const float* neg, int64_t n, float margin, float* loss, void* stream);
int ktup_loss_margin_bwd(const float* pos, const float* neg, int64_t n, float margin, const float* gloss,
                         float* gpos, float* gneg, void* stream);

/* ------------------------------------------- K10  regularisers  utils/loss.py:18-23
 * norm : sum over rows T[ids[i]] of max(|x|^2 - 1, 0);  orth: sum over rows of (Nrm[r].Rel[r])^2 / |Rel[r]|^2.
 * ids == NULL means rows 0..n-1 (whole table); otherwise the gather the reference does with a second
 * nn.Embedding lookup (item_recommendation.py:177-180, knowledge_representation.py:197-204) is fused.
 * Backward accumulates into table-shaped gradients (same pitch as the table).                            */
int ktup_reg_norm_fwd(const float* T, int64_t ld, int d, const int64_t* ids, int64_t n, float* loss, void* stream);
int ktup_reg_norm_bwd(const float* T, int64_t ld, int d, const int64_t* ids, int64_t n, const float* gloss,
                      float* gT, void* stream);
int ktup_reg_orth_fwd(const float* Rel, int64_t ldr, const float* Nrm, int64_t ldn, int d, const int64_t* ids,
                      int64_t n, float* loss, void* stream);
int ktup_reg_orth_bwd(const float* Rel, int64_t ldr, const float* Nrm, int64_t ldn, int d, const int64_t* ids,
                      int64_t n, const float* gloss, float* gRel, float* gNrm, void* stream);

/* Value + gradient in one launch (the GPU-resident training step issues these): same arithmetic as the _fwd / _bwd pairs
 * above, but the loss VALUE is added to *loss_acc instead of overwriting it -- the caller zeroes its loss slots once per
 * step -- and the gradients are produced in the same pass.                                                        */
int ktup_loss_bpr_fused(const float* pos, const float* neg, int64_t n, float target, const float* gloss, float* loss_acc,
                        float* gpos, float* gneg, void* stream);
int ktup_loss_margin_fused(const float* pos, const float* neg, int64_t n, float margin, const float* gloss, float* loss_acc,
                           float* gpos, float* gneg, void* stream);
int ktup_reg_norm_fused(const float* T, int64_t ld, int d, const int64_t* ids, int64_t n, const float* gloss, float* loss_acc,
                        float* gT, void* stream);
int ktup_reg_orth_fused(const float* Rel, int64_t ldr, const float* Nrm, int64_t ldn, int d, const int64_t* ids, int64_t n,
                        const float* gloss, float* loss_acc, float* gRel, float* gNrm, void* stream);

/* ------------------------------------------- K11-K16  all-candidate scores for evaluation
 * Every function writes the full (nq x n_cand) fp32 score matrix `out` (pitch ldo), which keeps the
 * reference's evaluate / evaluateRec / evaluateHead / evaluateTail drop-in; `ws` is caller scratch of the
 * size the matching *_workspace_bytes reports.  head != 0: query is (t, r), candidates are heads
 * (c = proj(t) - r); head == 0: query is (h, r), candidates are tails (c = proj(h) + r).                  */

/* K11  bprmf.py:51-54 : out = U[u] . I^T  (fp32-input MFMA). */
int ktup_eval_bprmf_scores(const float* U, int64_t ldu, const float* I, int64_t ldi, int d, const int64_t* u_ids,
                           int64_t nq, int64_t n_items, float* out, int64_t ldo, void* stream);

size_t ktup_eval_kg_workspace_bytes(int d, int64_t nq);
/* K12  transE.py:65-105.  C (n_cand x d) is the candidate table (normally E itself). */
int ktup_eval_transe_scores(const float* E, int64_t lde, const float* R, int64_t ldr, int d, const float* C,
                            int64_t ldc, int64_t n_cand, const int64_t* q, const int64_t* r, int64_t nq, int l1,
                            int head, float* out, int64_t ldo, float* ws, void* stream);
/* K13  transH.py:73-121, jTransUP.py:193-247 (there C includes the zero pad row, which is ranked). */
int ktup_eval_transh_scores(const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                            int d, const float* C, int64_t ldc, int64_t n_cand, const int64_t* q, const int64_t* r,
                            int64_t nq, int l1, int head, float* out, int64_t ldo, float* ws, void* stream);
/* K14  transR.py:80-128 + utils/misc.py:29-33 : candidates projected by the query's relation matrix. */
size_t ktup_eval_transr_workspace_bytes(int d, int64_t nq, int64_t n_ent, int n_rel);
/* The entity side of K14 does not depend on the queries: ktup_eval_transr_prepare computes it once per evaluation pass into
 * `ents_ws` (ktup_eval_transr_entities_workspace_bytes bytes, 16-byte aligned) -- |M_rho e|^2 per (relation, entity) for the
 * squared-L2 matrix-core route (d in {64,100,128}), the projected table M_rho e otherwise -- and ktup_eval_transr_scores takes
 * it as `ents_ws` (NULL: recomputed inside every call, as in round 1).  Same tables, l1 and shape on both calls.          */
size_t ktup_eval_transr_entities_workspace_bytes(int d, int64_t n_ent, int n_rel);
int ktup_eval_transr_prepare(const float* E, int64_t lde, const float* M, int64_t ldm, int d, int64_t n_ent, int n_rel, int l1,
                             float* ents_ws, void* stream);
int ktup_eval_transr_scores(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm,
                            int d, int64_t n_ent, int n_rel, const int64_t* q, const int64_t* r, int64_t nq, int l1,
                            int head, float* out, int64_t ldo, float* ws, const float* ents_ws, void* stream);
/* K15/K16  transUP.py:84-102 (E == NULL) and jTransUP.py:163-191.  item2ent has one int32 per ROW of I;
 * uniform is (nq x n_items x n_pref) for KTUP_GUMBEL_INPUT (the reference draws noise in evaluate too).   */
size_t ktup_eval_pref_workspace_bytes(int d, int n_pref, int64_t nq, int64_t n_items);
int ktup_eval_pref_scores(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                          const int32_t* item2ent, const float* pref_ws, int n_pref, int d, const int64_t* u_ids,
                          int64_t nq, int64_t n_items, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                          uint64_t offset, float* out, int64_t ldo, float* ws, void* stream);
/* The same in two halves, for an evaluation PASS (the weights are frozen while transUP.py:84-102 / jTransUP.py:163-191 are
 * called batch after batch): the item-side projections once, then every batch of users against them.  `ws` of the second
 * call needs ktup_eval_pref_workspace_bytes(d, n_pref, nq, 0) bytes (the user side only).                          */
size_t ktup_eval_pref_items_workspace_bytes(int d, int n_pref, int64_t n_items);
int ktup_eval_pref_items_prepare(const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                                 const float* pref_ws, int n_pref, int d, int64_t n_items, float* items_ws, void* stream);
int ktup_eval_pref_scores_prepared(const float* U, int64_t ldu, const float* pref_ws, int n_pref, int d, const int64_t* u_ids,
                                   int64_t nq, int64_t n_items, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                                   uint64_t offset, float* out, int64_t ldo, const float* items_ws, float* ws, void* stream);

/* K16 + K17 of a whole evaluation PASS fused (new): every user of `u_ids` against all items, filtered top-n, in one sweep that
 * never materialises the (users x items) matrix -- what jTransUP.py:163-191 / transUP.py:84-102 + utils/misc.py:186-248 compute
 * batch by batch.  Straight from the tables: items are I[row] (+ E[item2ent[row]] for KTUP; E == item2ent == NULL for TUP),
 * pref_ws the prepared preference tables (ktup_pref_prepare).  Every user x item cross term except u.v is contracted in
 * preference space (csrc/ktup_eval_pass.hip): scores agree with ktup_eval_pref_scores to fp32 rounding, the order is the
 * (score, id) order of ktup_eval_topk_filtered.  Soft gate, squared L2 (l1 == 0), d in {64, 100, 128}, n_pref <= 32,
 * topn <= 16; otherwise KTUP_ERR_UNSUPPORTED and the caller keeps the per-batch pair of calls.  filt_off / filt_ids: CSR filter
 * sets per user of u_ids (NULL = none); top_scores may be NULL.  `ws`: ktup_eval_pref_topk_workspace_bytes bytes, 16-byte aligned. */
size_t ktup_eval_pref_topk_workspace_bytes(int d, int n_pref, int64_t nq, int64_t n_items, int topn);
int ktup_eval_pref_topk(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                        const int32_t* item2ent, const float* pref_ws, int n_pref, int d, const int64_t* u_ids, int64_t nq,
                        int64_t n_items, int l1, const int64_t* filt_off, const int32_t* filt_ids, int topn, int32_t* top_ids,
                        float* top_scores, float* ws, void* stream);

/* The same pass for the HARD (ST-Gumbel) gate and for L1 (transUP.py:84-102 with use_st_gumbel; csrc/ktup_eval.hip
 * sweep_hard_kernel): the pair arithmetic and the noise stream positions of ktup_eval_pref_scores -- the scores are that call's
 * bits for the same (gumbel_mode, uniform, seed, offset) over all nq users at once, i.e. pair (b, j) draws at
 * ((b * n_items + j) * n_pref + p) + offset -- with the filtered top-n taken where the scores are made (KTUP_GUMBEL_INPUT: uniform
 * is (nq x n_items x n_pref)); any d % 4 == 0, n_pref <= 32, topn <= 16.  gumbel_mode == KTUP_GUMBEL_OFF: the SOFT gate scored pair
 * by pair (the arithmetic of ktup_eval_pref_scores, its bits) with the same epilogue -- the one-sweep pass for L1 and for widths
 * ktup_eval_pref_topk does not cover (d <= 168); KTUP_ERR_UNSUPPORTED where the stage does not fit the LDS.                    */
size_t ktup_eval_pref_topk_hard_workspace_bytes(int d, int n_pref, int64_t nq, int64_t n_items, int topn);
int ktup_eval_pref_topk_hard(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                             const int32_t* item2ent, const float* pref_ws, int n_pref, int d, const int64_t* u_ids, int64_t nq,
                             int64_t n_items, int l1, int gumbel_mode, const float* uniform, uint64_t seed, uint64_t offset,
                             const int64_t* filt_off, const int32_t* filt_ids, int topn, int32_t* top_ids, float* top_scores,
                             float* ws, void* stream);

/* ------------------------------------------- K17/K18  ranking walk  utils/misc.py:125-146,213-248
 * Order: ascending score (descending != 0 negates first, misc.py:93,180), ties -> lower id (declared rule).
 * Filter sets are CSR: filt_off[nq + 1] (int64) into filt_ids (int32); filt_off == NULL means no filter.
 * topk : the first `topn` unfiltered candidate ids per query (-1 padded), optionally their scores.
 * ranks: for every gold entry (CSR gold_off / gold_ids) its 0-based rank among unfiltered NON-gold
 *        candidates (other golds do not advance the rank); -1 if the gold id is itself filtered.
 * n_cand <= 19000 (ml1m: 3240 items / 14709 entities): the row's keys sit in LDS, one read of the score row.
 * Larger catalogues stream through LDS in 16 K-key chunks (same results; topn <= 1024 there); ids are 32-bit. */
int ktup_eval_topk_filtered(const float* scores, int64_t lds, int64_t nq, int64_t n_cand, int descending,
                            const int64_t* filt_off, const int32_t* filt_ids, int topn, int32_t* top_ids,
                            float* top_scores, void* stream);
int ktup_eval_gold_ranks(const float* scores, int64_t lds, int64_t nq, int64_t n_cand, int descending,
                         const int64_t* filt_off, const int32_t* filt_ids, const int64_t* gold_off,
                         const int32_t* gold_ids, int32_t* ranks, void* stream);

/* A whole link-prediction pass in one call (knowledge_representation.py:93-146 evaluate: evaluateHead / evaluateTail per batch
 * + utils/misc.py:61-146 per batch, here with the loop over batches under the ABI): ranks[g] = the filtered 0-based rank of gold
 * entry g (as ktup_eval_gold_ranks; -1 for a gold id that is itself filtered) for all nq keys (q[i], r[i]).  filt_off / gold_off
 * are the pass's CSR offsets (nq + 1 entries, ABSOLUTE offsets into filt_ids / gold_ids); the keys are scored `chunk` at a time
 * (512 = the reference's batch) into `ws` (ktup_eval_kg_ranks_workspace_bytes) by K12 / K13, so the integers are those of the
 * per-batch route.  model: KTUP_KG_TRANSE (Nrm ignored) or KTUP_KG_TRANSH; head / l1 / C as in ktup_eval_trans{e,h}_scores. */
#define KTUP_KG_TRANSE 0
#define KTUP_KG_TRANSH 1
/* The pass WITHOUT the (keys x candidates) score matrix (csrc/ktup_eval_kg_fused.hip), TransE / TransH: the ranks are counts formed
 * where the scores are made -- rank(g) = #{c : key(c) < key(g)} minus the filtered ids and other golds of the key ordered before g,
 * whose scores come from a small launch that runs the sweep's own instruction sequence (bit-identical keys).  Squared L2 at d in
 * {20, 36, 64, 100, 128} with at most 8 golds per key sweeps on the matrix cores; L1, other widths and larger gold sets score the
 * tiles with the pair kernels of the per-batch entry points (any d) and count in their epilogue.  Same arguments and results as
 * ktup_eval_kg_ranks, all keys in one go; n_filt / n_gold = lengths of the id arrays, max_golds = the largest gold set of a key
 * (host side).  n_rel = rows of R / Nrm: TransH's product w.e depends on (relation, candidate) only and comes from an
 * n_rel x n_cand table computed once per pass in `ws` -- by the matrix-core sweep and, filled with the pair kernels' own first-pass
 * operations (the same bits), by the pair-kernel route (n_rel = 0, option kg_wtab = 0, a table beyond 1 GiB or rows that are not
 * 16-byte aligned: the product is recomputed per pair; the same integers either way).                                           */
int ktup_eval_kg_ranks_fused_supported(int model, int d, int l1, int64_t max_golds);
size_t ktup_eval_kg_ranks_fused_workspace_bytes(int model, int d, int64_t nq, int64_t n_gold, int64_t n_filt, int64_t n_cand,
                                                int64_t n_rel);
int ktup_eval_kg_ranks_fused(int model, const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                             int64_t n_rel, int d, const float* C, int64_t ldc, int64_t n_cand, const int64_t* q, const int64_t* r, int64_t nq, int l1, int head,
                             int descending, const int64_t* filt_off, const int32_t* filt_ids, int64_t n_filt,
                             const int64_t* gold_off, const int32_t* gold_ids, int64_t n_gold, int64_t max_golds, int32_t* ranks,
                             void* ws, void* stream);
/* TransR (transR.py:80-128): the same pass over the entity table; ents_ws = ktup_eval_transr_prepare's output for these tables
 * (NULL: the entity side is recomputed per chunk).  The score matrix of a chunk is double-buffered and K18 of chunk c runs on a
 * library-owned second stream beside the score kernel of chunk c + 1 (not during graph capture).                              */
size_t ktup_eval_kg_ranks_transr_workspace_bytes(int d, int64_t n_ent, int n_rel, int64_t chunk);
int ktup_eval_kg_ranks_transr(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm, int d,
                              int64_t n_ent, int n_rel, const float* ents_ws, const int64_t* q, const int64_t* r, int64_t nq,
                              int l1, int head, int descending, const int64_t* filt_off, const int32_t* filt_ids,
                              const int64_t* gold_off, const int32_t* gold_ids, int32_t* ranks, int64_t chunk, void* ws,
                              void* stream);
size_t ktup_eval_kg_ranks_workspace_bytes(int d, int64_t n_cand, int64_t chunk);
int ktup_eval_kg_ranks(int model, const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn, int d,
                       const float* C, int64_t ldc, int64_t n_cand, const int64_t* q, const int64_t* r, int64_t nq, int l1,
                       int head, int descending, const int64_t* filt_off, const int32_t* filt_ids, const int64_t* gold_off,
                       const int32_t* gold_ids, int32_t* ranks, int64_t chunk, void* ws, void* stream);

/* ranks over a candidate SHARD (sharded-catalogue evaluation, one shard per GPU): the rank of ktup_eval_gold_ranks is a count, so
 * it is additive over disjoint shards.  `scores` (nq x n_local) covers candidates [cand_lo, cand_lo + n_local) with GLOBAL ids in
 * the filter / gold CSR lists; gold_scores[e] = the score of gold entry e (from the shard that owns it).  counts[e] = this shard's
 * unfiltered non-gold candidates ordered before gold e (same (score, global id) order), or a large negative value when the gold
 * is itself filtered: rank = all-reduce(sum) of counts, negative -> -1.                                                   */
int ktup_eval_gold_rank_counts(const float* scores, int64_t lds, int64_t nq, int64_t n_local, int64_t cand_lo,
                               int descending, const int64_t* filt_off, const int32_t* filt_ids,
                               const int64_t* gold_off, const int32_t* gold_ids, const float* gold_scores,
                               int32_t* counts, void* stream);
/* the same for a shard that is a LATTICE of the catalogue: local candidate j has the global id cand_lo + cand_stride * j (the rows
 * g % world == rank of a table sharded by row: cand_lo = rank, cand_stride = world) -- evaluation straight from the shards of -shard_tables */
int ktup_eval_gold_rank_counts_strided(const float* scores, int64_t lds, int64_t nq, int64_t n_local, int64_t cand_lo,
                                       int64_t cand_stride, int descending, const int64_t* filt_off, const int32_t* filt_ids,
                                       const int64_t* gold_off, const int32_t* gold_ids, const float* gold_scores,
                                       int32_t* counts, void* stream);

/* K18b  utils/misc.py:232-248 + utils/evaluation.py:80-110 (ndcg_at_k, method 0): per query (f1, precision, recall,
 * hit, ndcg) as 5 float64 from its ranked id list (`topn` entries, -1 padded as ktup_eval_topk_filtered writes them)
 * and its gold ids (CSR, ASCENDING within a query).  precision divides by the number of valid entries, recall by
 * |gold|, the NDCG ideal is the best ordering of the observed hits -- all as the reference computes them.        */
int ktup_eval_rec_metrics(const int32_t* top_ids, int64_t nq, int topn, const int64_t* gold_off, const int32_t* gold_ids,
                          double* out, void* stream);

/* ------------------------------------------- row-sharded tables (new: the reference is single-device)
 * Device halves of the all-to-all lookup exchange for tables partitioned by `row % world_size`:
 * pack  : out[k,:] = table[ids[k],:]         owner side, rows a peer asked for -> contiguous send buffer
 * unpack: gtable[ids[k],:] += rows[k,:]      owner side, returned row gradients -> shard gradient (atomics) */
/* Negative ids are padding (fixed-capacity id lists): pack writes a zero row, unpack and the sparse step skip them.
 * dedupe: uniq[0 .. *n_unique) = the distinct ids of the batch (no particular order), the rest of uniq[0 .. n) = -1,
 *         inverse[e] = position of ids[e] in uniq; *n_unique is one DEVICE int32 -- the host never needs the count.
 *         `ws`: ktup_shard_dedupe_workspace_bytes(n) bytes, 8-byte aligned (an open-addressing hash table).              */
size_t ktup_shard_dedupe_workspace_bytes(int64_t n);
int ktup_shard_dedupe(const int64_t* ids, int64_t n, int64_t* uniq, int64_t* inverse, int32_t* n_unique, void* ws, void* stream);
int ktup_shard_pack_rows(const float* table, int64_t ldt, int d, const int64_t* ids, int64_t n, float* out,
                         int64_t ldo, void* stream);
int ktup_shard_unpack_rows_add(const float* rows, int64_t ldr, int d, const int64_t* ids, int64_t n, float* gtable,
                               int64_t ldg, void* stream);

/* owner-side sparse optimizer: update only the `n` UNIQUE rows `ids` of a shard with the
 * combined row gradients `grows` (n x d).  g' = g * min(1, max_norm / (sqrt(*sumsq) + 1e-6)) (sumsq: one device double, the
 * job-wide sum of squared gradients; max_norm <= 0: no clip).  KTUP_OPT_SGD: p -= lr g';  KTUP_OPT_ADAGRAD: state += g'^2,
 * p -= lr g' / (sqrt(state) + eps).  Equal to the reference's dense step (utils/trainer.py:63-77) when l2_lambda = 0.   */
int ktup_shard_sparse_step(int kind, float* table, int64_t ldt, float* state, int64_t lds, int d, const int64_t* ids,
                           int64_t n, const float* grows, int64_t ldg, float lr, float eps, const double* sumsq,
                           float max_norm, void* stream);

/* ---- the same exchange with FIXED shapes: config 5's step without a host synchronisation (csrc/ktup_shard_step.hip) -----------
 * Replaces, for the KTUP rec step over row-sharded tables, the lines knowledgable_recommendation.py:335-344,394-403 (model(pos),
 * model(neg), bprLoss, backward, clip_grad_norm, optimizer.step) over jTransUP.py:122-143; the reference is single-device.
 *
 * Wire layout.  T <= KTUP_SHARD_MAX_TABLES tables, `world` owners (owner of global row g = g % world, its local row = g / world),
 * cap[t] slots per (owner, table): wire row  w = owner * capsum + toff[t] + slot  with toff = prefix sums of cap, capsum = sum of
 * cap, W = world * capsum rows (+ one all-zero row W that stands for "no row").  The id send buffer, the row buffer, the compact
 * table the scorer reads and the gradient buffer all use this layout, so every exchange is an all-to-all with EQUAL splits
 * (capsum rows per peer) and nothing is re-ordered between them.
 *
 * ktup_shard_route: `ids` = n_entries ids in blocks of `block` entries; inside a block entries [ent_off[t], ent_off[t+1]) belong to
 *   table t (ent_off: n_tables + 1 HOST values, ent_off[n_tables] = block); negative ids are padding.  Outputs, all on the device:
 *   inverse[e] = wire row of entry e (W for padding); send_ids[w] = owner-local row of wire row w, -1 where the slot is unused;
 *   pair_map (may be NULL; needs n_entries == block): for an entry of table pair_a, pair_map[its wire row] = wire row of the entry
 *   at the same position of table pair_b, or W if that entry is padding (KTUP: item row -> entity row, jTransUP.py:114-130);
 *   sort_ws (ktup_shard_route_sort_bytes(n_entries, W) bytes, int32): the entries counting-sorted by wire row for
 *   ktup_shard_reduce_rows; counters: world * n_tables + 1 int32, the last one counts ids that found no free slot (cap too small:
 *   the caller must then skip the step -- ktup_shard_apply does when handed that word); zero_doubles: n_zero_doubles device doubles
 *   cleared by the same launch (the step's accumulators).  `ws`: ktup_shard_route_workspace_bytes(n_entries), 8-byte aligned.
 *   No memset, no host read: five launches, capturable; sort_ws 16-byte aligned.  On the owner side of a multi-rank step the same call with world = 1 and
 *   block = capsum groups the rows several peers asked for.
 * ktup_shard_reduce_rows: gwire[w] += sum of G rows of the entries sorted to wire row w; entry e reads row e of G for e < n_src,
 *   row e - src_off otherwise (KTUP: entries [u | pos ; neg | entities], G = [GU ; GV] of ktup_train_rec_step_rows, 3B rows; the 2B
 *   entity entries re-read GV: n_src = 3B, src_off = 2B).  Sorted segments (csrc/ktup_segreduce.hip), no per-element atomics.
 * ktup_shard_ktup_entries: entries = [u | pos ; neg | item2ent[pos ; neg]] (5B, or 3B when item2ent is NULL), an entity of
 *   value < 0 or == ent_pad becoming padding.
 * ktup_shard_pack_wire: out[w] = table_t[ids[w]] for the n_blocks * capsum rows of a wire buffer (t from w's place in its block;
 *   rows with ids[w] < 0 are left untouched).  tables / ld / cap: HOST arrays of n_tables device pointers / pitches / capacities.
 * ktup_shard_apply: global-norm clip (coef = min(1, max_norm / (sqrt(sum of sumsq[0 .. sumsq_slots)) + 1e-6))) + row-sparse SGD / Adagrad
 *   (ktup_shard_sparse_step's rule) for every wire row with ids[w] >= 0, and for `n_small` small replicated gradients of
 *   small_rows x d each: gradient k updates small_p0[k] (and small_p1[k] if not NULL -- the two summands of a mixed table share
 *   one gradient); with small_g64 the small gradients are read from that fp64 array (the all-reduced bucket) instead.  Consumed
 *   gradient rows are zero-filled.  If *skip_count != 0 or *skip_value != 0 (either may be NULL) nothing is updated, gradients are
 *   still cleared.  The launch also closes the step's books (all four may be NULL / 0): the step kernels accumulate their loss
 *   terms into loss_step[0 .. n_loss); here loss_sum[k] += loss_step[k] unless the step is skipped, loss_step := 0, and a skipped
 *   step adds 1 to *skipped_steps -- a counter no launch ever clears, so a run can be checked once at its end.
 * ktup_shard_bucket: mode 0: bucket = [small gradients as doubles | sum of sumsq_local[0 .. sumsq_slots) | *overflow]; mode 1 (after the all-reduce):
 *   *sumsq_total = bucket[n] + small_weight * sum of squares of bucket[0 .. n) (2 when every small gradient feeds two tables).
 * ktup_zero_async: zero-fill by a kernel (16-byte aligned, multiple of 16 bytes).                                          */
#define KTUP_SHARD_MAX_TABLES 4
#define KTUP_SHARD_MAX_SMALL 4
#define KTUP_SHARD_SUMSQ_SLOTS 16
size_t ktup_shard_route_workspace_bytes(int64_t n_entries);
size_t ktup_shard_route_sort_bytes(int64_t n_entries, int64_t n_wire_rows);
int ktup_shard_route(const int64_t* ids, int64_t n_entries, int64_t block, int n_tables, const int64_t* ent_off, int world,
                     const int64_t* cap, int pair_a, int pair_b, int64_t* inverse, int64_t* send_ids, int32_t* pair_map,
                     int32_t* sort_ws, int32_t* counters, double* zero_doubles, int n_zero_doubles, void* ws, void* stream);
/* ktup_shard_route for the KTUP rec step with the entry list built by its first launch: batch (*cursor mod n_batches) of the id
 * columns u / pos_items / neg_items (n_batches x B each; cursor NULL: batch 0) -> entries = [u | pos ; neg | item2ent[pos ; neg]]
 * (5B int64, written; 3B / two tables when item2ent is NULL; a user is one entry, shared by its positive and its negative pair), tables 0 / 1 / 2 = users / items / entities, pair_map = item wire row
 * -> entity wire row.  *cursor (device) is incremented by the call: a replayed graph walks through the columns by itself.
 * phase: 0 = the whole route; 1 = its first launch only (scratch init + the entry list); 2 = the remaining four launches -- a
 * scorer that needs nothing but the entry list (global ids) can then run beside phase 2 on another stream; 3 = the whole route
 * WITHOUT moving the cursor -- for a scorer that reads the id columns itself and runs beside ALL of it (ktup_shard_reduce_norm moves
 * the cursor once both are done); 4 = the first three launches (send_ids, inverse, pair_map: what an id exchange and the scorer wait
 * for); 5 = the last two (the counting sort's scan + scatter, read by the row-gradient reduction only) -- phase 5 may then run on
 * another stream beside the id exchange and the pack launch.                                                                   */
int ktup_shard_route_ktup(const int64_t* u, const int64_t* pos_items, const int64_t* neg_items, int64_t B, int64_t n_batches,
                          int64_t* cursor, const int32_t* item2ent, int64_t ent_pad, int64_t* entries, int world,
                          const int64_t* cap, int64_t* inverse, int64_t* send_ids, int32_t* pair_map, int32_t* sort_ws,
                          int32_t* counters, double* zero_doubles, int n_zero_doubles, void* ws, int phase, void* stream);
/* The same for the KTUP kg step (knowledgable_recommendation.py:346-383): batch (*cursor mod n_batches) of the six triple columns
 * -> entries = [ph ; pt ; nh ; nt] (4B int64, written; nh / nt = -1 where it equals ph / pt), ONE table (entities; cap: one value); rels = [pr ; nr] (2B, written): the
 * relation ids ktup_train_kg_step_rows reads.  phase as above.                                                                */
int ktup_shard_route_kg(const int64_t* ph, const int64_t* pt, const int64_t* pr, const int64_t* nh, const int64_t* nt,
                        const int64_t* nr, int64_t B, int64_t n_batches, int64_t* cursor, int64_t* entries, int64_t* rels,
                        int world, const int64_t* cap, int64_t* inverse, int64_t* send_ids, int32_t* sort_ws, int32_t* counters,
                        double* zero_doubles, int n_zero_doubles, void* ws, int phase, void* stream);
int ktup_shard_reduce_rows(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                           int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, void* stream);
int ktup_shard_ktup_entries(const int64_t* u, const int64_t* pos_items, const int64_t* neg_items, int64_t B,
                            const int32_t* item2ent, int64_t ent_pad, int64_t* entries, void* stream);
int ktup_shard_pack_wire(int n_tables, float* const* tables, const int64_t* ld, const int64_t* cap, int d, const int64_t* ids,
                         int64_t n_blocks, float* out, int64_t ldo, void* stream);
/* Row-sparse ADAM (kind = KTUP_OPT_ADAM in ktup_shard_apply / ktup_shard_reduce_apply) that reproduces the reference's DENSE Adam
 * to within ~1e-6 absolute per element (the replay is cut after `replay` steps and seeds its bias corrections from fp32 exponentials;
 * plain SGD / Adagrad need no replay and are exact)
 * (utils/trainer.py:63-66: torch.optim.Adam over whole tables, weight_decay = l2_lambda = 0; the configuration of the published
 * recipe, ktup.sh:1): a dense Adam step moves every row that ever had a gradient -- m <- beta1 m, v <- beta2 v,
 * p <- p - lr / (1 - beta1^s) m / (sqrt(v) / sqrt(1 - beta2^s) + eps) -- whether the batch touches it or not.  A state row is
 * [m (d) | v (d) | last (int32) | 3 words of padding], KTUP_SHARD_ADAM_STATE_PITCH(d) floats; `last` = the step the row's state was
 * written at (0: never).  Touching a row at step t first REPLAYS the zero-gradient steps last + 1 .. t - 1 in registers (the dense
 * recurrence, step by step -- or, for eight steps or more of a state old enough that its bias corrections move slowly, as ONE series
 * per row in eps / (sqrt(v) + eps) whose remainder is bounded below 1e-6 of the replayed displacement; after `replay` steps, when the
 * increments have fallen below ~1e-5 of the first, only m and v keep decaying in closed form), then applies step t.  `step` points at TWO device int64: step[0] = the number of the step being applied,
 * step[1] = that step's bias corrections {1 - beta1^t, sqrt(1 - beta2^t)} as two floats; ktup_shard_step_count -- the launch before
 * the apply launch, and the only writer of both -- moves the counter (+1 unless the step is skipped) and refreshes them.  ktup_shard_adam_flush replays every row of a shard up to
 * *step (before an evaluation or a checkpoint reads the table).
 * WEIGHT DECAY (the reference builds every optimizer with weight_decay = l2_lambda, 1e-5 by default: utils/trainer.py:63-77, base.py:51): the
 * dense step adds weight_decay * p to the gradient of EVERY row at EVERY step, touched or not.  The same state rows and the same three
 * launches carry it: `rule` = 0 Adam, 1 Adagrad (its sum in the `v` half of the state row), 2 plain SGD; with weight_decay > 0 a touched row
 * first takes the steps last + 1 .. t - 1 on g = weight_decay * p one by one (from step 1 on: `last` = 0 is a row never written), then step t
 * on g + weight_decay * p; catch-up and flush likewise.  (Adagrad / SGD without weight decay keep their plain row-sparse forms, kind =
 * KTUP_OPT_ADAGRAD / KTUP_OPT_SGD; kind = KTUP_OPT_ADAM means "state rows [m | v | last] and this rule".)                      */
typedef struct ktup_adam_t { float beta1, beta2; int32_t replay; int32_t rule; const int64_t* step; float weight_decay; float reserved; } ktup_adam_t;
#define KTUP_SHARD_ADAM_STATE_PITCH(d) (2 * (d) + 4)
int ktup_shard_step_count(int64_t* step, const int32_t* skip_count, const double* skip_value, float beta1, float beta2, void* stream);
int ktup_shard_adam_flush(float* table, int64_t ldt, float* state, int64_t lds, int d, int64_t n_rows, float lr, float eps,
                          const ktup_adam_t* adam_rule, void* stream);
/* The catch-up has to come BEFORE a step reads a row (the pending zero-gradient moves are part of the weights the reference's forward
 * sees): ktup_shard_adam_catchup replays, in ONE launch, the rows ids[k][0 .. n_rows[k]) of table k for up to KTUP_SHARD_ADAM_MAX_SEG
 * tables (owner-local row numbers, DISTINCT within a segment, negative = padding; ids == NULL or ids[k] == NULL: rows 0 .. n_rows[k] - 1)
 * up to *step -- after the route has named the step's distinct rows, before the pack launch / the step kernel gathers them.  The apply
 * launch of the step then finds last == step - 1 on every row it touches.                                                       */
#define KTUP_SHARD_ADAM_MAX_SEG 8
int ktup_shard_adam_catchup(int n_seg, float* const* tables, const int64_t* ld, float* const* states, const int64_t* lds,
                            const int64_t* const* ids, const int64_t* n_rows, int d, float lr, float eps,
                            const ktup_adam_t* adam_rule, void* stream);
int ktup_shard_apply(int kind, int n_tables, float* const* tables, const int64_t* ld, float* const* states, const int64_t* lds,
                     const int64_t* cap, int d, const int64_t* ids, int64_t n_blocks, float* grads, int64_t ldg, int n_small,
                     int small_rows, float* const* small_grads, float* const* small_p0, float* const* small_s0,
                     float* const* small_p1, float* const* small_s1, const double* small_g64, float lr, float eps,
                     const double* sumsq, int sumsq_slots, float max_norm, const int32_t* skip_count, const double* skip_value,
                     float* loss_step, int n_loss, float* loss_sum, int32_t* skipped_steps, const ktup_adam_t* adam_rule, void* stream);
int ktup_shard_bucket(int mode, int n_small, float* const* small_grads, int64_t small_elems, double* bucket,
                      const double* sumsq_local, int sumsq_slots, const int32_t* overflow, double* sumsq_total, double small_weight,
                      void* stream);
int ktup_zero_async(void* ptr, int64_t nbytes, void* stream);
/* reduce -> norm -> apply WITHOUT a W x d gradient buffer: the segment reduction of ktup_shard_reduce_rows run twice over G.
 * ktup_shard_reduce_norm: the squared norm of every reduced row (and of the n_small small gradients, weighted by small_weight) is
 *   ADDED to sumsq[0 .. n_slots) (n_slots = 1 or KTUP_SHARD_SUMSQ_SLOTS); nothing else is written, except that the few rows whose
 *   entries straddle two workgroups are summed into gwire (all-zero before the call) and listed in xkeys
 *   (ktup_shard_reduce_list_len(n_entries, d) int32) -- their squared norm is accumulated by the adds themselves (an add of v onto
 *   `old` raises the square by 2 old v + v^2, and the atomic returns old).  One launch (the small gradients, `fold` and `cursor` ride in
 *   extra workgroups).  dup_only != 0: the kernel that wrote G has already added
 *   |G row|^2 for every ENTRY (ktup_train_rec_step_rows / ktup_train_kg_step_rows with `sumsq`), so the walk adds only what rows
 *   shared by several entries change -- |sum of the rows|^2 - sum of |row|^2 -- and never reads an entry that is alone on its row.
 *   fold / n_fold (may be NULL / 0): accumulators that a kernel on ANOTHER branch of the step's graph filled while sumsq was being
 *   cleared (the step kernel beside the route): their sum is added to sumsq and they are left zero.  cursor (may be NULL): the batch
 *   position of ktup_shard_route_ktup / _kg with phase 3, moved on here, after every reader of it.
 * ktup_shard_reduce_apply: the same walk again; every reduced row goes straight from registers through the clipped row-sparse
 *   SGD / Adagrad rule of ktup_shard_apply into table_t[ids[w]]; the listed rows are applied from gwire, which is left all-zero
 *   again; small tables as in ktup_shard_apply -- both as extra workgroups of the ONE launch.  Same arguments, same G, same sort_ws as the norm call. */
int64_t ktup_shard_reduce_list_len(int64_t n_entries, int d);
/* The requester's side of the gradient exchange WITHOUT a zero-filled buffer and without a read-modify-write:
 * ktup_shard_reduce_store: gwire[w] = sum of the G rows of the entries sorted to wire row w (arguments of ktup_shard_reduce_rows).  A row whose
 *   entries lie inside one workgroup's stretch of the sorted order -- all rows with one entry, nearly all others -- is STORED; a row cut by a
 *   workgroup's edge is summed by float atomics and must be zero beforehand, which is what
 * ktup_shard_zero_shared_rows does: it zero-fills every wire row that has two or more entries (named by the entry of rank 1 in the row; any
 *   time after ktup_shard_route's finish launch, phase 4).  Wire rows without an entry are not written at all (the owner never reads them:
 *   their id is negative). */
int ktup_shard_zero_shared_rows(const int32_t* sort_ws, int64_t n_entries, int64_t n_wire_rows, const int64_t* inverse, float* gwire,
                                int64_t ldw, int d, void* stream);
int ktup_shard_reduce_store(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                            int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, void* stream);
int ktup_shard_reduce_store_fold(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                                 int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, float* rep, int n_rep,
                                 int64_t rep_elems, float* const* rep_dst, void* stream);
int ktup_shard_reduce_norm(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                           int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, int32_t* xkeys, int n_small,
                           float* const* small_grads, int64_t small_elems, float small_weight, double* sumsq, int n_slots,
                           int dup_only, double* fold, int n_fold, int64_t* cursor, void* stream);
int ktup_shard_reduce_norm_fold(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                                int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, int32_t* xkeys, int n_small,
                                float* const* small_grads, int64_t small_elems, float small_weight, double* sumsq, int n_slots,
                                int dup_only, double* fold, int n_fold, int64_t* cursor, float* rep, int n_rep,
                                float* const* rep_dst, void* stream);
int ktup_shard_reduce_apply(int kind, int n_tables, float* const* tables, const int64_t* ld, float* const* states,
                            const int64_t* lds, const int64_t* cap, const int64_t* ids, int64_t n_blocks, const float* G,
                            int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws, int64_t n_entries,
                            float* gwire, int64_t ldw, const int32_t* xkeys, int n_small, int small_rows,
                            float* const* small_grads, float* const* small_p0, float* const* small_s0, float* const* small_p1,
                            float* const* small_s1, const double* small_g64, float lr, float eps, const double* sumsq,
                            int sumsq_slots, float max_norm, const int32_t* skip_count, const double* skip_value,
                            float* loss_step, int n_loss, float* loss_sum, int32_t* skipped_steps, const ktup_adam_t* adam_rule,
                            void* stream);

/* ------------------------------------------- K19  negative sampling on the device  utils/data.py:12-85
 * rec: one uniform negative item per (u, positive): != positive, bit not set in the user's row of
 *      `user_item_bitmap` (n_users x words_per_user uint32, train + eval items; NULL = no filter), and -- when
 *      unique_in_batch -- not drawn by another row of this batch (data.py:77-82).  Batch uniqueness is deterministic:
 *      (seed, offset, inputs) fix the batch (earlier try rounds win, then lower rows), so replicas that draw the same
 *      global batch agree.  `ws`: ktup_negsample_rec_workspace_bytes(n_items) bytes, 8-byte aligned (unique_in_batch only).
 * kg : per triple a fair coin corrupts head or tail with a uniform entity != original whose triple is not in
 *      `sorted_keys` (ascending uint64 ((h * n_rel + r) * n_ent + t) of every known triple; NULL = no filter).
 * Draws are Philox4x32-10(seed, offset + row * 4096 + attempt): advance `offset` by n * 4096 per call.
 * Outputs are ALWAYS valid row indices: after 4096 failed tries a deterministic scan finds an admissible candidate; if
 * none exists the row gets an in-range stand-in and the device counter *fail_count (int32, caller-zeroed, may be NULL)
 * is incremented -- check it before trusting a run (jTransUP/utils/device_sampler.py DeviceSampler.check).            */
size_t ktup_negsample_rec_workspace_bytes(int64_t n_items);
int ktup_negsample_rec(const int64_t* u_ids, const int64_t* pos_items, int64_t n, int64_t n_items,
                       const uint32_t* user_item_bitmap, int64_t words_per_user, uint64_t seed, uint64_t offset,
                       int unique_in_batch, int64_t* neg_items, void* ws, int32_t* fail_count, void* stream);
int ktup_negsample_kg(const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int64_t n_ent, int64_t n_rel,
                      const uint64_t* sorted_keys, int64_t n_keys, uint64_t seed, uint64_t offset, int64_t* neg_h,
                      int64_t* neg_t, int32_t* fail_count, void* stream);
/* Feed (new): one launch builds a whole training batch in the steppers' layout from DEVICE-resident state, so that a step
 * replays from a HIP graph with no host work in between (the reference assembles each batch in python: utils/data.py:87-110
 * MakeTrainIterator, then :64-85 / :12-56).  `col_*`: this epoch's shuffled example columns (n_rows each; the host reshuffles
 * them in place once per epoch and resets *cursor); rows [*cursor, *cursor + B) are the batch.  Negatives exactly as
 * ktup_negsample_rec / _kg draw them for (seed, *offset_dev).  Outputs, 2B ids each: u2 = [u ; u], i2 = [pos ; neg] resp.
 * h2 = [h ; neg_h], t2 = [t ; neg_t], r2 = [r ; r] -- the id arrays of ktup_train_rec_step / ktup_train_kg_step.  Afterwards
 * *cursor += B and *offset_dev += B * 4096.  A cursor outside [0, n_rows - B] counts as a failure and restarts at row 0.
 * ktup_feed_rec's `ws` (unique_in_batch; ktup_negsample_rec_workspace_bytes) must be all-ones (0xff bytes) before the FIRST call;
 * every call leaves it all-ones again -- no memset inside the launch (a memset node captured in a HIP graph was observed to stop
 * taking effect after eager memsets ran between replays).                                                              */
int ktup_feed_rec(const int64_t* col_u, const int64_t* col_i, int64_t n_rows, int64_t B, int64_t* cursor, uint64_t* offset_dev,
                  int64_t n_items, const uint32_t* user_item_bitmap, int64_t words_per_user, uint64_t seed, int unique_in_batch,
                  int64_t* u2, int64_t* i2, void* ws, int32_t* fail_count, void* stream);
int ktup_feed_kg(const int64_t* col_h, const int64_t* col_t, const int64_t* col_r, int64_t n_rows, int64_t B, int64_t* cursor,
                 uint64_t* offset_dev, int64_t n_ent, int64_t n_rel, const uint64_t* sorted_keys, int64_t n_keys, uint64_t seed,
                 int64_t* h2, int64_t* t2, int64_t* r2, int32_t* fail_count, void* stream);

/* ------------------------------------------- K20  global-norm clip + dense optimizer step
 * Replaces   torch.nn.utils.clip_grad_norm(params, max_norm); optimizer.step()
 * (item_recommendation.py:189-192, knowledge_representation.py:209-211, knowledgable_recommendation.py:399-401) for the
 * optimizers of utils/trainer.py:63-77 with weight_decay = l2_lambda, in two launches over a list of tables.
 * `params/grads/state1/state2/sizes/steps/first` are HOST arrays of n_tensors entries (device pointers, element counts,
 * each tensor's own 1-based step count for Adam, 1 where a momentum buffer is still uninitialised).
 * state1 / state2:  SGD momentum_buffer / -;  Adagrad sum / -;  Adam exp_avg / exp_avg_sq;  RMSprop square_avg /
 * momentum_buffer.  The clip factor min(1, max_norm / (||g|| + 1e-6)) is applied to the gradients in place, like
 * clip_grad_norm_; max_norm <= 0 disables clipping (sumsq, one device double, may then be NULL).  zero_grads != 0
 * stores zeros instead (optimizer.zero_grad() of the NEXT step folded into this pass).  `steps_dev` (Adam; may be
 * NULL): the same step counts as n_tensors int64 in DEVICE memory, read when the kernel runs -- it overrides `steps`
 * and makes the launch replayable from a HIP graph (the caller increments the counters with a captured launch).      */
#define KTUP_OPTIM_MAX_TENSORS 12
#define KTUP_OPT_SGD 0
#define KTUP_OPT_ADAGRAD 1
#define KTUP_OPT_ADAM 2
#define KTUP_OPT_RMSPROP 3
int ktup_optim_gradnorm(int n_tensors, float* const* grads, const int64_t* sizes, double* sumsq, void* stream);
/* the same sum ADDED to sumsq[0 .. n_slots) -- workgroup b adds to slot b mod n_slots, the result is the sum of the slots (one double
 * atomic per workgroup costs ~20 ns on ONE address) -- with no clearing memset inside: for HIP-graph callers that keep their
 * accumulators clean themselves */
int ktup_optim_gradnorm_acc(int n_tensors, float* const* grads, const int64_t* sizes, double* sumsq, int n_slots, void* stream);
int ktup_optim_step(int kind, int n_tensors, float* const* params, float* const* grads, float* const* state1,
                    float* const* state2, const int64_t* sizes, const int64_t* steps, const int64_t* steps_dev,
                    const int32_t* first, float lr, float weight_decay, float momentum, float beta1, float beta2, float eps, float alpha,
                    const double* sumsq, float max_norm, int zero_grads, void* stream);

/* ------------------------------------------- the B = 512 training step in two launches (new; ktup_train_step.hip)
 * One fused launch per step kind does everything between the sampled ids and the dense gradients; ktup_optim_clip_step follows.
 *
 * ktup_train_rec_step  (knowledgable_recommendation.py:335-344 / item_recommendation.py:160-182):
 *   rows k and k + B of (u_ids, i_ids) are the positive and the negative pair of example k.  Mixes the RAW preference-side
 *   tables (what ktup_pref_prepare does), scores both pairs (jTransUP.py:122-143 / transUP.py:69-82), adds
 *   mean_k -logsigmoid(target (pos_k - neg_k)) to loss[0] and, if orth != 0, orthogonalLoss(pref, pref_norm) to loss[1], and
 *   ACCUMULATES the gradients of (batch-mean + orth) x gscale into gU / gI / gE (pad entity row skipped) and gP, gPn (+ gR, gRn:
 *   the mixed-table gradient goes to both summands).  TUP: E == item2ent == rel == norm == gE == gR == gRn == NULL.
 *   d in {64, 100, 128} with n_pref <= 32, or d = 256 with n_pref <= 20; the preference-side tables and every gradient are
 *   contiguous (pitch d).
 * ktup_train_kg_step   (knowledgable_recommendation.py:345-382 / knowledge_representation.py:176-204), TransH (transh != 0) or
 *   TransE: rows k and k + B of (h, t, r) are the positive triple and its corrupted twin.  loss[0] += sum_k max(pos - neg +
 *   margin, 0); regs bit 0: loss[1] += orthogonalLoss(R[r], Nrm[r]) over the 2B relation ids, bit 1: loss[2] += normLoss over
 *   the 4B entity rows, bit 2: loss[3] += normLoss over the 2B relation rows; gradients x gscale accumulated into gE / gR / gN.
 * ktup_train_step_supported(kind, d, n_pref): 1 if the fused kernel exists (kind 0 rec, 1 kg TransH, 2 kg TransE).
 *
 * ktup_optim_clip_step = ktup_optim_gradnorm + ktup_optim_step (same arguments, same arithmetic) as ONE launch: the gradients
 *   stay in registers across a grid-wide barrier between the norm and the update (at most 512 workgroups, all resident).  `ws`:
 *   KTUP_OPTIM_WS_DOUBLES doubles, zero-filled ONCE by the caller; every launch leaves them consistent for the next, ws[0] =
 *   the squared gradient norm of the last clipped step.  max_norm <= 0: no clipping, no barrier.  loss_slots (may be NULL):
 *   *loss_out = loss_scale * sum(loss_slots[0..n_slots)) (also added to *loss_acc, a running sum, if not NULL) and
 *   loss_slots := 0 for the next step -- the step kernels above accumulate into them.  The barrier's poll is bounded; after a timeout (never observed; a lost workgroup would otherwise
 *   hang the GPU) ws[KTUP_OPTIM_WS_DOUBLES - 1] reads non-zero (as a uint64) and that step was applied unclipped.
 *
 * gnorm (may be NULL; KTUP_GNORM_WS_DOUBLES doubles, zero-filled ONCE by the caller, the same pointer to the step launch and to the
 *   ktup_optim_clip_step that follows it): the gradient norm WITHOUT a pass over the gradients.  ktup_train_rec_step /
 *   ktup_train_kg_step then issue every gradient add as a returning atomic and track the squared norm of the buffers they build
 *   (adding v onto a cell that held `old` raises it by (2 old + v) v), one double atomic per workgroup into the workspace;
 *   ktup_optim_clip_step reads the sum instead of running its norm pass and its grid barrier (utils/trainer.py:63-81
 *   clip_grad_norm_ + step: same norm up to fp32 rounding of the tracked terms).  Requires that the gradient buffers were zero-filled
 *   before the step launch (zero_grads of the previous ktup_optim_clip_step), that nothing else adds to them in between (one
 *   process: an all-reduce of the gradients changes their norm) and that max_norm > 0.                                       */
#define KTUP_GNORM_WS_DOUBLES 64
int ktup_train_step_supported(int kind, int d, int n_pref);
int ktup_train_rec_step(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                        const int32_t* item2ent, int64_t ent_pad, const float* pref, const float* pref_norm,
                        const float* rel, const float* norm, int64_t ldp, int n_pref, int d, const int64_t* u_ids,
                        const int64_t* i_ids, int64_t B, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                        uint64_t offset, float target, float gscale, int orth, float* loss, float* gU, float* gI, float* gE,
                        float* gP, float* gPn, float* gR, float* gRn, double* gnorm, void* stream);
/* ktup_train_rec_step with the row gradients STORED instead of accumulated by atomics -- for ktup_shard_reduce_rows / large
 * batches: u_ids holds B ids (example k's user: a BPR example's positive and negative pair share it), i_ids 2B (positives then
 * negatives); row k of GU (B x d) = the user-row gradient of example k from BOTH pairs, row k of GV (2B x d) = the item-row
 * gradient of pair k (which is also its entity row's).  sumsq (may be NULL): n_slots doubles to which the launch ADDS
 * sum_k |GU row k|^2 + sum_k |GV row k|^2 x (1 + [pair k's item has an entity row]) -- the squared norm of the row gradients as if
 * every entry of the route had its own table row (ktup_shard_reduce_norm with dup_only corrects for shared rows).  neg_ids (may be
 * NULL): u_ids / i_ids / neg_ids are then the id COLUMNS of ktup_shard_route_ktup (n_batches x B each) and the kernel reads batch
 * (*cursor mod n_batches) of them itself (cursor NULL: batch 0): it does not wait for an entry list.  Soft gate only.  gR / gRn may be NULL although rel / norm are given (then orth must be 0): gP / gPn are the gradients
 * of both summands of the mixed tables.                                                                                     gumbel_mode / gumbel: the preference gate -- KTUP_GUMBEL_OFF (soft, gumbel NULL), KTUP_GUMBEL_INPUT (gumbel = 2B x n_pref
 * uniforms, positives then negatives, as transUP.py:159-162 draws them: the parity mode) or KTUP_GUMBEL_PHILOX_DEV (gumbel = device uint64
 * {seed, offset}: draw (row k of [pos ; neg], preference p) is Philox number offset + k n_pref + p; the caller moves the offset on). */
int ktup_train_rec_step_rows(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                             const int32_t* item2ent, int64_t ent_pad, const float* pref, const float* pref_norm,
                             const float* rel, const float* norm, int64_t ldp, int n_pref, int d, const int64_t* u_ids,
                             const int64_t* i_ids, int64_t B, int l1, float target, float gscale, int orth, float* loss,
                             float* GU, float* GV, float* gP, float* gPn, float* gR, float* gRn, double* sumsq, int n_slots,
                             const int64_t* neg_ids, const int64_t* cursor, int64_t n_batches, int gumbel_mode, const void* gumbel,
                             void* stream);
/* The same with REPLICAS of the small tables' gradients (small_ws: ktup_train_rec_step_rows_ws_bytes(B, n_pref, d) bytes =
 * KTUP_TRAIN_SMALL_REPLICAS x [A | C] x n_pref x d floats, 16-byte aligned, ZERO before the call; NULL / too small = the call above).
 * Without them every tile workgroup of the launch adds its 2 x n_pref x d partial sums of the preference-table gradients to gP / gPn
 * (/ gR / gRn) by float atomics as the workgroups end together: 256 adds per address at config 5, 10 us of a 117 us step (twice that when pref
 * and rel have gradients of their own).  With them workgroup b adds to replica b mod KTUP_TRAIN_SMALL_REPLICAS -- 4 us -- and gP / gPn /
 * gR / gRn receive NOTHING from the tile workgroups (orthogonalLoss's gradient still goes straight to gP / gPn): the caller's next
 * launch must fold the replicas in -- ktup_shard_reduce_norm_fold (one rank: the norm walk's extra workgroups sum the replicas, add the sum
 * to the gradients, take the folded values' squares for the norm) or ktup_shard_reduce_store_fold (several ranks: the same as extra
 * workgroups of the requester's reduction, before the bucket launch reads the gradients); both leave the replicas zero.
 * rep_dst = {gP, gPn, gR, gRn} (the last two NULL when rel / norm share gP / gPn); rep_elems = n_pref x d.                        */
#define KTUP_TRAIN_SMALL_REPLICAS 8
size_t ktup_train_rec_step_rows_ws_bytes(int64_t B, int n_pref, int d);
int ktup_train_rec_step_rows_ws(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                const int32_t* item2ent, int64_t ent_pad, const float* pref, const float* pref_norm,
                                const float* rel, const float* norm, int64_t ldp, int n_pref, int d, const int64_t* u_ids,
                                const int64_t* i_ids, int64_t B, int l1, float target, float gscale, int orth, float* loss,
                                float* GU, float* GV, float* gP, float* gPn, float* gR, float* gRn, double* sumsq, int n_slots,
                                const int64_t* neg_ids, const int64_t* cursor, int64_t n_batches, int gumbel_mode, const void* gumbel,
                                void* small_ws, size_t small_ws_bytes, void* stream);
/* TUP's row regularisers (item_recommendation.py:177-180: normLoss(user rows of the batch) + normLoss(item rows of [pos ; neg]) +
 * normLoss(pref); normLoss(x) = sum_rows max(|x|^2 - 1, 0), utils/loss.py:21-23) for a step with STORED row gradients: after
 * ktup_train_rec_step_rows and before the reduction, row k of GU (example k's user, id u_ids[k]) and row k of GV (pair k's item, id i_ids[k],
 * k < 2B) take 2 scale_rows x for rows with |x|^2 > 1, gP likewise with scale_pref for the n_pref rows of `pref` (contiguous, pitch d);
 * loss[0] += scale_rows x (the two row terms), loss[1] += scale_pref x normLoss(pref).  GU / GV: pitch d.                            */
int ktup_train_rec_reg_rows(const float* U, int64_t ldu, const float* I, int64_t ldi, int d, const int64_t* u_ids, const int64_t* i_ids,
                            int64_t B, float* GU, float* GV, const float* pref, int n_pref, float* gP, float scale_rows, float scale_pref,
                            float* loss, void* stream);
int ktup_train_kg_step(int transh, const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                       int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t B, int l1, float margin,
                       float gscale, int regs, float* loss, float* gE, float* gR, float* gN, double* gnorm, void* stream);
/* ktup_train_kg_step for row-sharded entity tables (config 5's kg steps; csrc/ktup_shard_kg.hip): ent_ids = [ph ; pt ; nh ; nt]
 * (4B rows of E: a shard's rows, or wire rows of a compact table), rel_ids = [pr ; nr] (2B); the entity-row gradients of triple k
 * are STORED as rows k, B + k, 2B + k, 3B + k of GE (4B x d, pitch d) for ktup_shard_reduce_norm / _apply instead of accumulated
 * into a table-shaped gradient; gR / gN (relation-side, replicated) are accumulated.  An nh / nt that is negative or == ent_pad
 * says "the same entry as ph / pt" (a corrupted triple keeps its head or its tail; ktup_shard_route_kg writes -1 there): the twin's
 * gradient is added to the positive's stored row and the twin's own row is not written.  `order` (may be NULL): the triples' indices
 * sorted by relation (ktup_shard_kg_rel_order) -- consecutive triples of one relation then share ONE flush of its gradient rows.
 * sumsq / n_slots (may be NULL): += the sum of |row|^2 over the 4B stored rows, as in ktup_train_rec_step_rows.
 * ktup_shard_kg_rel_order: order[0 .. B) = a counting sort of rel[0 .. B) over [0, n_rel) in one launch (n_rel > 16384: identity). */
int ktup_shard_kg_rel_order(const int64_t* rel, int64_t B, int64_t n_rel, int32_t* order, void* stream);
int ktup_train_kg_step_rows(int transh, const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                            int d, const int64_t* ent_ids, int64_t ent_pad, const int64_t* rel_ids, const int32_t* order, int64_t B, int l1,
                            float margin, float gscale, int regs, float* loss, float* GE, float* gR, float* gN, double* sumsq,
                            int n_slots, void* stream);
#define KTUP_OPTIM_WS_DOUBLES 784
/* largest grid of ktup_optim_clip_step on the current device: its grid barrier needs every workgroup resident at once, so the
 * launch is sized from the occupancy query x the CU count (one workgroup per CU short of it), at most 512; 0 = unknown (the
 * one-launch entry point then refuses and the caller uses ktup_optim_gradnorm + ktup_optim_step).                           */
int ktup_optim_clip_step_capacity(int kind);
int ktup_optim_clip_step(int kind, int n_tensors, float* const* params, float* const* grads, float* const* state1,
                         float* const* state2, const int64_t* sizes, const int64_t* steps, const int64_t* steps_dev,
                         const int32_t* first, float lr, float weight_decay, float momentum, float beta1, float beta2, float eps,
                         float alpha, double* ws, double* gnorm, float max_norm, int zero_grads, float* loss_slots, int n_slots,
                         float loss_scale, float* loss_out, float* loss_acc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KTUP_HIP_H */
