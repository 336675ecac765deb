// A whole link-prediction pass behind one call: for every (query entity, relation) key the filtered 0-based ranks of its gold
// entities among ALL candidates (knowledge_representation.py:93-146 evaluate -> model.evaluateHead / evaluateTail per batch of
// 512 keys -> utils/misc.py:61-146 evalProcess / getKGPerformance).
//
// The reference walks the batches in python: score matrix (B x N) -> host -> argsort -> walk.  The per-batch route of this
// library keeps that shape on the device (K12 / K13 + K18), but at 14,709 entities one batch is ~55-70 us of kernels behind
// ~200 us of python (iterator, RankIndex slices, two ctypes calls, scratch allocation): the pass is host-bound 3x over.  Here the
// loop over batches moves under the C ABI: the keys of the pass are handed over once, the score matrix of ONE chunk of keys
// lives in the workspace and is overwritten by the next chunk (same stream: the rank kernel of chunk c has read it before the
// score kernel of chunk c + 1 starts), and the CSR filter / gold offsets are ABSOLUTE offsets into the pass's id arrays, so a
// chunk is served by `off + c0` with the id arrays untouched.  The kernels are exactly those of the per-batch route: the ranks
// are the same integers.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ktup_common.h"

// Round 3: the score matrix of a chunk is DOUBLE-BUFFERED and the rank kernel of chunk c runs on the library's side stream while
// the score kernel of chunk c + 1 fills the other buffer: K12 / K13 is bound by writing the matrix, K18 by reading it, and their
// sum (50 + 19 us per 512 keys at ml1m-kg size) becomes their maximum.  Never while the caller's stream is being captured
// (fork_side then declines and the chunks run back to back on one stream, as before).  TransR joins the entry point family
// (ktup_eval_kg_ranks_transr: entity side prepared once per pass by the caller, K14 per chunk).
namespace {

size_t score_bytes(int64_t chunk, int64_t n_cand) { return (((size_t)chunk * (size_t)n_cand * sizeof(float)) + 255) & ~(size_t)255; }

struct Ev {
  hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
  bool tried = false, ok = false;
} g_ev;

bool events_ok() {
  if (!g_ev.tried) {
    g_ev.tried = true;
    g_ev.ok = true;
    for (int i = 0; i < 2; ++i)
      if (hipEventCreateWithFlags(&g_ev.ready[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&g_ev.done[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        g_ev.ok = false;
      }
  }
  return g_ev.ok;
}

// score(c0, nb, out) fills `out` (nb x n_cand) for keys [c0, c0 + nb) on stream st.
template <typename ScoreFn>
int kg_pass(const char* name, ScoreFn score, int64_t n_cand, int64_t nq, int descending, const int64_t* filt_off, const int32_t* filt_ids,
            const int64_t* gold_off, const int32_t* gold_ids, int32_t* ranks, int64_t chunk, float* scores2, hipStream_t st) {
  const size_t sb = score_bytes(chunk, n_cand) / sizeof(float);
  hipStream_t side = (nq > chunk && events_ok()) ? ktup::fork_side(st) : nullptr;
  int64_t c = 0;
  for (int64_t c0 = 0; c0 < nq; c0 += chunk, ++c) {
    const int64_t nb = nq - c0 < chunk ? nq - c0 : chunk;
    const int b = (int)(c & 1);
    float* out = scores2 + (side ? b * sb : 0);
    if (side && c >= 2 && hipStreamWaitEvent(st, g_ev.done[b], 0) != hipSuccess) return ktup::check_launch(name);   // ranks(c - 2) has read this buffer
    int rc = score(c0, nb, out, st);
    if (rc != KTUP_OK) { ktup::join_side(st, side); return rc; }
    hipStream_t rs = st;
    if (side) {
      if (hipEventRecord(g_ev.ready[b], st) != hipSuccess || hipStreamWaitEvent(side, g_ev.ready[b], 0) != hipSuccess) return ktup::check_launch(name);
      rs = side;
    }
    rc = ktup_eval_gold_ranks(out, n_cand, nb, n_cand, descending, filt_off ? filt_off + c0 : nullptr, filt_ids, gold_off + c0, gold_ids, ranks,
                              rs);
    if (rc != KTUP_OK) { ktup::join_side(st, side); return rc; }
    if (side && hipEventRecord(g_ev.done[b], side) != hipSuccess) return ktup::check_launch(name);
  }
  ktup::join_side(st, side);                   // the caller's stream continues after the last rank kernel
  return KTUP_OK;
}

}  // namespace

extern "C" size_t ktup_eval_kg_ranks_workspace_bytes(int d, int64_t n_cand, int64_t chunk) {
  if (d <= 0 || n_cand <= 0 || chunk <= 0) return 0;
  return 2 * score_bytes(chunk, n_cand) + ktup_eval_kg_workspace_bytes(d, chunk);
}

extern "C" int ktup_eval_kg_ranks(int model, const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                                  int d, const float* C, int64_t ldc, int64_t n_cand, const int64_t* q, const int64_t* r, int64_t nq,
                                  int l1, int head, int descending, const int64_t* filt_off, const int32_t* filt_ids,
                                  const int64_t* gold_off, const int32_t* gold_ids, int32_t* ranks, int64_t chunk, void* ws,
                                  void* stream) {
  const char* name = "ktup_eval_kg_ranks";
  KTUP_REQUIRE(model == KTUP_KG_TRANSE || model == KTUP_KG_TRANSH, "%s: model must be KTUP_KG_TRANSE or KTUP_KG_TRANSH", name);
  KTUP_REQUIRE(nq >= 0 && n_cand > 0 && chunk > 0 && d > 0, "%s: bad sizes", name);
  if (nq == 0) return KTUP_OK;
  KTUP_REQUIRE(E && R && C && q && r && gold_off && gold_ids && ranks && ws && (model == KTUP_KG_TRANSE || Nrm), "%s: null pointer argument", name);
  KTUP_REQUIRE((filt_off == nullptr) || filt_ids, "%s: filter offsets without ids", name);
  float* scores = reinterpret_cast<float*>(ws);
  float* qws = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + 2 * score_bytes(chunk, n_cand));
  auto score = [&](int64_t c0, int64_t nb, float* out, hipStream_t st) {
    return model == KTUP_KG_TRANSE
               ? ktup_eval_transe_scores(E, lde, R, ldr, d, C, ldc, n_cand, q + c0, r + c0, nb, l1, head, out, n_cand, qws, st)
               : ktup_eval_transh_scores(E, lde, R, ldr, Nrm, ldn, d, C, ldc, n_cand, q + c0, r + c0, nb, l1, head, out, n_cand, qws, st);
  };
  return kg_pass(name, score, n_cand, nq, descending, filt_off, filt_ids, gold_off, gold_ids, ranks, chunk, scores, (hipStream_t)stream);
}

// TransR (transR.py:80-128 per batch + utils/misc.py:61-146): the same pass; the candidates are the entity table itself, whose
// projected side `ents_ws` (ktup_eval_transr_prepare, once per pass) is handed to K14 for every chunk.
extern "C" size_t ktup_eval_kg_ranks_transr_workspace_bytes(int d, int64_t n_ent, int n_rel, int64_t chunk) {
  if (d <= 0 || n_ent <= 0 || chunk <= 0 || n_rel <= 0) return 0;
  return 2 * score_bytes(chunk, n_ent) + ktup_eval_transr_workspace_bytes(d, chunk, n_ent, n_rel);
}

extern "C" int ktup_eval_kg_ranks_transr(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm, int d,
                                         int64_t n_ent, int n_rel, const float* ents_ws, const int64_t* q, const int64_t* r, int64_t nq,
                                         int l1, int head, int descending, const int64_t* filt_off, const int32_t* filt_ids,
                                         const int64_t* gold_off, const int32_t* gold_ids, int32_t* ranks, int64_t chunk, void* ws,
                                         void* stream) {
  const char* name = "ktup_eval_kg_ranks_transr";
  KTUP_REQUIRE(nq >= 0 && n_ent > 0 && n_rel > 0 && chunk > 0 && d > 0, "%s: bad sizes", name);
  if (nq == 0) return KTUP_OK;
  KTUP_REQUIRE(E && R && M && q && r && gold_off && gold_ids && ranks && ws, "%s: null pointer argument", name);
  KTUP_REQUIRE((filt_off == nullptr) || filt_ids, "%s: filter offsets without ids", name);
  float* scores = reinterpret_cast<float*>(ws);
  float* qws = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + 2 * score_bytes(chunk, n_ent));
  auto score = [&](int64_t c0, int64_t nb, float* out, hipStream_t st) {
    return ktup_eval_transr_scores(E, lde, R, ldr, M, ldm, d, n_ent, n_rel, q + c0, r + c0, nb, l1, head, out, n_ent, qws, ents_ws, st);
  };
  return kg_pass(name, score, n_ent, nq, descending, filt_off, filt_ids, gold_off, gold_ids, ranks, chunk, scores, (hipStream_t)stream);
}
