// Geometry of the prepared preference tables (ktup_pref_prepare workspace), shared by the training-path
// kernels (ktup_score_pref.hip) and the all-candidate evaluation kernels (ktup_eval.hip).
#pragma once
#include "ktup_common.h"

namespace ktup {

constexpr int TR = 64;  // pairs per tile (= lanes per wave)
constexpr int PB = 5;   // preferences handled per stage-1 pass by one wave

// Row pitch (in floats) of the prepared tables, and rows of the zero-padded logit table.
struct PrefGeom {
  int CH, NW;   // float4 chunks per lane in stage 2, waves per workgroup
  int dp;       // padded row length in floats = 4 * NW * CH
  int ppad;     // logit-table rows: multiple of NW * PB
  bool ok;
};
inline PrefGeom pref_geom(int d, int n_pref) {
  PrefGeom g{0, 0, 0, 0, false};
  if (d <= 0 || d % 4 || n_pref <= 0) return g;
  const int nch = d / 4;
  if (nch <= 16) { g.CH = 4; g.NW = 4; }
  else if (nch <= 28) { g.CH = 7; g.NW = 4; }
  else if (nch <= 32) { g.CH = 8; g.NW = 4; }
  else if (nch <= 64) { g.CH = 8; g.NW = 8; }
  else return g;
  g.dp = 4 * g.NW * g.CH;
  const int step = g.NW * PB;
  g.ppad = ((n_pref + step - 1) / step) * step;
  g.ok = true;
  return g;
}
// workspace layout (floats): Alog[ppad][dp] | Ar[P][dp] | Cn[P][dp]   (row-major, zero padded)
inline size_t ws_floats(const PrefGeom& g, int n_pref) { return ((size_t)(g.ppad + 2 * n_pref) * g.dp + 15) & ~(size_t)15; }


// Constant-address-space view of a read-only table: a wave-uniform index selects s_load_dwordx4 and the
// value lives in SGPRs (one scalar operand per v_fmac), not in VGPRs / LDS.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f* sptr4;
KTUP_DEV sptr4 as_scalar(const float4* p) { return (sptr4)(uintptr_t)p; }
KTUP_DEV sptr4 as_scalar(const float* p) { return (sptr4)(uintptr_t)p; }
KTUP_DEV float4 sld(sptr4 p, int idx) { const v4f v = p[idx]; return make_float4(v.x, v.y, v.z, v.w); }
KTUP_DEV float4 sldp(sptr4 p) { const v4f v = *p; return make_float4(v.x, v.y, v.z, v.w); }
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) v16f* sptr16;  // 64-byte scalar loads (s_load_dwordx16)
KTUP_DEV sptr16 as_scalar16(const float* p) { return (sptr16)(uintptr_t)p; }

// ktup_score_pref_row.hip: rows wider than 256 columns (any multiple of 4): one wave per pair, nothing held per coordinate.  The
// prepared tables of such a width are plain [P][d] blocks: Alog | Ar | Cn (ktup_pref_prepare with ppad = P, dp = d).  i_ids == nullptr:
// the pairs are (u_ids[b], j), j in [0, n_items), scores to score[b * ldo + j] (the evaluation's all-item form; forward only).
bool pref_row_covers(int d, int n_pref);
inline size_t pref_row_ws_floats(int d, int n_pref) { return ((size_t)3 * n_pref * d + 15) & ~(size_t)15; }
int pref_row(bool bwd, const char* name, const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
             const int32_t* item2ent, int64_t ent_pad, const float* pref_ws, int n_pref, int d, const int64_t* u_ids, const int64_t* i_ids,
             int64_t n, int64_t n_items, int64_t ldo, int l1, int gumbel_mode, const float* uniform, uint64_t seed, uint64_t offset,
             float* score, const float* gscore, float* gU, float* gI, float* gE, float* gA, float* gC, hipStream_t st, int ppad = 0,
             int dp = 0);      // (ppad, dp): the tile kernels' padded table geometry when d <= 256 comes here (0: the plain blocks)

// ktup_score_pref_mc.hip: compile-time-geometry matrix-core forward (soft gate and ST-Gumbel gate).  Returns 1 when (d, n_pref) is not an
// instantiated geometry.
int pref_fwd_mc(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                const float* Alog, const float* Ar, const float* Cn, int dp, int n_pref, int d, const int64_t* u_ids,
                const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform, uint64_t seed, uint64_t offset,
                float* score, hipStream_t st, const char* name);

// ktup_score_transr_mc.hip: relation-bucketed matrix-core TransR forward.  Returns 1 when the shape is not covered.
size_t transr_mc_workspace_bytes(int64_t n, int64_t n_rel);
int transr_fwd_mc(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm, int64_t n_rel, int d,
                  const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1, float* score, void* ws, hipStream_t st,
                  const char* name);
// relation-bucketed matrix-core TransR backward (same scratch as the forward); G != NULL: entity-row gradients stored per triple
int transr_bwd_mc(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm, int64_t n_rel, int d,
                  const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1, const float* gscore, float* gE, float* gR,
                  float* gM, float* G, void* ws, hipStream_t st, const char* name);

// ktup_eval_mc.hip: squared-L2 soft-gate all-item scores as six matrix-core GEMMs.  Returns 1 for sizes it does not cover.
int pairs_l2_mc(const float* QW, const float* C0, const float* C1, const float* C2, int d, int64_t nq, int64_t n_items, float* out,
                int64_t ldo, hipStream_t st, const char* name);

// ktup_eval.hip: query-side vectors of the KG evaluation kernels (model 0 TransE, 1 TransH, 2 TransR) for all nq keys
int kg_query_prep(int model, const float* E, int64_t lde, const float* R, int64_t ldr, const float* X, int64_t ldx, int d,
                  const int64_t* q, const int64_t* r, int64_t nq, int head, float* QW, hipStream_t st, const char* name);

// ktup_eval_pass.hip: scores + filtered top-n of a whole evaluation pass in one launch (+ a merge launch).  1 = not covered.
size_t eval_pass_pspace_bytes(int d, int n_pref, int64_t nq, int64_t n_items, int topn);
int kg_valu_counts(int model, const float* QW, int d, const float* C, int64_t ldc, int64_t n_cand, int64_t nq, int l1, int descending,
                   const int64_t* gold_off, const int32_t* gold_ids, const int64_t* filt_off, const int32_t* filt_ids, float* gscore,
                   float* fscore, int32_t* counts, const int64_t* rel, const float* Nrm, int64_t ldn, int64_t n_rel, float* wtab, int64_t ldw,
                   hipStream_t st, const char* name);
int launch_topk_merge(const uint64_t* part, int64_t nq, int nsplit, int topn, int32_t* top_ids, float* top_scores, hipStream_t st, const char* name);
int eval_pass_pspace(const float* U, int64_t ldu, const int64_t* u_ids, int64_t nq, const float* I, int64_t ldi, const float* E, int64_t lde,
                     const int32_t* item2ent, int64_t n_items, const float* pref_ws,
                     int ppad, int dp, int n_pref, int d, const int64_t* filt_off, const int32_t* filt_ids, int topn, void* scratch,
                     int32_t* top_ids, float* top_scores, hipStream_t st, const char* name);
int pairs_kg_l2_mc(int model, const float* QW, int dq, const float* C, int64_t ldc, int d, int64_t nq, int64_t n_cand, float* out,
                   int64_t ldo, hipStream_t st, const char* name, const float* qcc = nullptr,
                   const float* cnorm = nullptr, const int32_t* qrel = nullptr);

// ktup_score_pref_bwd_mc.hip: matrix-core backward of the TUP / KTUP score (soft and ST-Gumbel gate).  Returns 1 for shapes it does not cover.
int pref_bwd_mc(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                int64_t ent_pad, const float* Alog, const float* Ar, const float* Cn, int dp, float beta, int n_pref, int d,
                const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                uint64_t offset, const float* gscore, float* gU, float* gI, float* gE, float* gA, float* gC, hipStream_t st,
                const char* name, float* GU = nullptr, float* GV = nullptr);
int pref_step_mc(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                 int64_t ent_pad, const float* pref, const float* pnorm, const float* rel, const float* norm, int64_t ldp, int n_pref,
                 int d, const int64_t* u_ids, const int64_t* i_ids, int64_t B, int l1, int gumbel_mode, const float* uniform,
                 uint64_t seed, uint64_t offset, float target, float gscale, int orth, float* loss, float* gU, float* gI, float* gE,
                 float* gP, float* gPn, float* gR, float* gRn, hipStream_t st,
                 const char* name, float* GU = nullptr, float* GV = nullptr, double* sumsq = nullptr, int sumsq_slots = 0,
                 const int64_t* neg_ids = nullptr, const int64_t* cursor = nullptr, int64_t n_batches = 1, double* gnorm = nullptr,
                 void* small_ws = nullptr, size_t small_ws_bytes = 0);
size_t pref_step_small_ws_bytes(int64_t B, int n_pref, int d);
// ktup_score_pref_bwd_wide.hip: the same backward for d = 256 (config 5): four waves share a 16-pair tile, 64 coordinates each.
int pref_bwd_mc_wide(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                     int64_t ent_pad, const float* Alog, const float* Ar, const float* Cn, int dp, float beta, int n_pref, int d,
                     const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                     uint64_t offset, const float* gscore, float* gU, float* gI, float* gE, float* gA, float* gC, hipStream_t st,
                     const char* name, float* GU, float* GV);

// ktup_segreduce.hip: gT[ids[e]] += (+/-) G[e mod n_src] by sorted segments (1 = shape not covered, caller keeps atomics)
size_t seg_ws_bytes(int64_t m, int64_t n_rows);
int seg_reduce(const float* G, int64_t ldg, int d, int64_t n_src, const int64_t* ids, int64_t m, int64_t sign_split, int64_t n_rows,
               float* gT, int64_t ldt, const int32_t* map2, int64_t pad2, float* gT2, int64_t ldt2, void* ws, hipStream_t st,
               const char* name);
// the same in two steps, so that the sort (a function of the ids alone) can run beside the kernel that writes G:
// seg_sort fills ws from ids (entries [0, n_src)) and ids2 (entries [n_src, m); may be null when m == n_src), seg_apply reduces.
bool seg_covers(const float* G, int64_t ldg, int d, int64_t n_src, int64_t m, int64_t n_rows, const float* gT, int64_t ldt,
                const int32_t* map2, const float* gT2, int64_t ldt2, const void* ws);
int seg_sort(const int64_t* ids, const int64_t* ids2, int64_t n_src, int64_t m, int64_t n_rows, void* ws, hipStream_t st, const char* name);
int seg_apply(const float* G, int64_t ldg, int d, int64_t n_src, int64_t m, int64_t sign_split, int64_t n_rows, float* gT, int64_t ldt,
              const int32_t* map2, int64_t pad2, float* gT2, int64_t ldt2, const void* ws, hipStream_t st, const char* name);
// for callers that produce the histogram themselves (ktup_shard_step.hip): the scan with a 1024-thread workgroup, and the
// reduction over a given sorted order whose entry count lives on the device (entry e >= n_src reads row e - src_off of G)
int seg_scan_wide(int32_t* start, int64_t K, hipStream_t st, const char* name);
int seg_apply_sorted(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* perm, const int32_t* skey,
                     int64_t m_max, const int32_t* m_dev, float* gT, int64_t ldt, hipStream_t st, const char* name);

}  // namespace ktup
