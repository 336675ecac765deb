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

namespace {
size_t score_bytes(int64_t chunk, int64_t n_cand) { return (((size_t)chunk * (size_t)n_cand * sizeof(float)) + 255) & ~(size_t)255; }
}  // namespace

extern "C" size_t ktup_eval_kg_ranks_workspace_bytes(int d, int64_t n_cand, int64_t chunk) {
  if (d <= 0 || n_cand <= 0 || chunk <= 0) return 0;
  return score_bytes(chunk, n_cand) + ktup_eval_kg_workspace_bytes(d, chunk);
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
  float* qws = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + score_bytes(chunk, n_cand));
  for (int64_t c0 = 0; c0 < nq; c0 += chunk) {
    const int64_t nb = nq - c0 < chunk ? nq - c0 : chunk;
    int rc = model == KTUP_KG_TRANSE
                 ? ktup_eval_transe_scores(E, lde, R, ldr, d, C, ldc, n_cand, q + c0, r + c0, nb, l1, head, scores, n_cand, qws, stream)
                 : ktup_eval_transh_scores(E, lde, R, ldr, Nrm, ldn, d, C, ldc, n_cand, q + c0, r + c0, nb, l1, head, scores, n_cand, qws, stream);
    if (rc != KTUP_OK) return rc;
    rc = ktup_eval_gold_ranks(scores, n_cand, nb, n_cand, descending, filt_off ? filt_off + c0 : nullptr, filt_ids, gold_off + c0, gold_ids,
                              ranks, stream);
    if (rc != KTUP_OK) return rc;
  }
  return KTUP_OK;
}
