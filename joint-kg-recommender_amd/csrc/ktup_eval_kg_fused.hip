// A link-prediction pass WITHOUT the (keys x entities) score matrix: squared-L2 TransE / TransH (transE.py:65-105, transH.py:73-121,
// jTransUP.py:193-247) scored on the matrix cores exactly as ktup_eval_mc.hip's K12 / K13 do, with the filtered gold ranks of
// utils/misc.py:125-146 taken from COUNTS formed in the score kernel's epilogue instead of from a matrix that is written and read
// back (30 MB per 512 keys at ml1m-kg size: K12 / K13 + K18 were 59-74 us per chunk, most of it that round trip).
//
//   rank(g) = #{c : key(c) < key(g)} - #{c in filter(q) U gold(q), c != g : key(c) < key(g)},   key = (score image << 32 | id)
//
// (the reference's walk skips filtered ids and the other golds, so both are subtracted; a gold that is itself filtered is never
// reached: rank -1).  Three launches after the query preparation:
//   kg_list_scores  the scores of every (key, gold) and (key, filtered id) pair -- a few dozen per key -- computed with the SAME
//                   instruction sequence as the sweep (per wave: 16 keys x 16 gathered candidates per MFMA tile, of which the
//                   diagonal is kept), so that a candidate's key is bit-identical in both places;
//   kg_count_mc     the sweep: a workgroup keeps 64 keys' query vectors in LDS (registers where they fit) and walks one eighth of
//                   the candidate table (one band per XCD: its L2 then serves all the workgroups that walk it), 64 candidates per
//                   stage, double-buffered: ONE barrier per stage; each wave owns a 16 x 16 tile per stage and compares its four
//                   scores per lane with the (<= 8) gold scores of its keys -- as FLOATS, two VALU instructions per (score, gold);
//                   only when some lane sees a tie or a NaN (v_cmp_nlg: the gold itself, once per band) does the wave build the
//                   64-bit keys, so the order is that of the keys in every case; counts in registers, one flush of int atomics
//                   per workgroup at the end;
//   kg_rank_finalize  the subtraction above, per gold entry, from the list scores.
// TransH's second product w.e (w = the relation's hyperplane normal) depends on (relation, candidate) only: kg_wtab computes the
// (relations x candidates) table once per pass with the same MFMA sequence (mode 2), and the sweep is TransE's plus four table
// look-ups per lane and stage, fetched one stage ahead -- half the matrix instructions of mode 1 (both products in the sweep), which
// stays for relation counts whose table would not fit.
// EXACT ORDER NEAR THE GOLDS.  The sweep's score is |c|^2 - 2 c.e + |e|^2 (+ the TransH terms) from dot products, the reference's
// sum_k (c_k - e_k)^2 (transE.py:65-105, utils/misc.py:125-146 ranks by it): the two round differently, ~1e-6 on scores of O(1), and a
// candidate that close to a gold could land on the other side of it (8 of 3,478 ranks off by one at ml1m size).  So a comparison only
// COUNTS when it is safe: score < gold - T or score > gold + T, with T = 2 kappa (|c|^2 + max_e |e|^2), kappa = 2 (d + 8) 2^-24
// (1 + |w|^2)^2: the forward error of an n-term fp32 dot product is <= gamma_n |x||y| in any summation order, so a score is off by
// <= gamma_d (|c|^2 + |e|^2 + 2 |c||e|) <= 2 gamma_d (|c|^2 + |e|^2) (+ 8 for the additions and the rounding of c itself), and a
// comparison involves two scores.  A candidate inside the window goes to a list (one segment per workgroup, so the
// slot counters are uncontended; ~1e-4 of the pairs) -- and so do the filtered ids and other golds of a key that the finalize step
// finds inside the window, with the opposite sign.  kg_unc_resolve_kernel, the pass's LAST launch, decides every listed comparison
// in fp64 from the tables themselves -- sum_k ((q_k -+ r_k) - e_k)^2, TransH with both projections, ids break exact ties -- and
// adds +-1 to the rank.  The ranks are then those of the exact scores of the fp32 tables; where the fp32 sum of the reference's own
// formula disagrees, fp64 is the referee (tests/test_hip_eval.py).  A segment that overflows (an entity table made of equal rows:
// everything ties) switches the WHOLE pass back to the keys of the sweep's own scores -- sweep, finalize and list alike, so the
// counts stay consistent.
// The sweep covers squared L2 at d in {20, 36, 64, 100, 128}; L1 and every other width take the pair kernels' COUNT form
// (ktup_eval.hip kg_valu_counts) behind the same entry point, with this file's finalize step.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "ktup_common.h"
#include "ktup_lane_swap.h"
#include "ktup_pref_geom.h"

namespace ktup {
namespace {

constexpr int IB = 64, UB = 64, NW = 16, GS = 4, NBAND = 8;   // GS: golds a sweep launch counts for (further launches take golds
                                                              // 4.., 8.. of the few keys that have them)

// MODE 0: TransE; 1: TransH, both products in the sweep; 2: TransH, w.e from the (relations x candidates) table
template <int NCH_, int MODE_>
struct FGeom {
  static constexpr int NCH = NCH_, D = 4 * NCH, MODE = MODE_;
  static constexpr bool TRANSH = MODE == 1, WTAB = MODE == 2;
  static constexpr int KG = (D + 15) / 16;
  static constexpr bool TAIL1 = NCH - 4 * (KG - 1) == 1;
  static constexpr int KGF = TAIL1 ? KG - 1 : KG;
  static_assert(TAIL1 || NCH % 4 == 0, "k groups must be whole (d % 16 in {0, 4})");
  static constexpr int P4 = NCH | 1;
  static constexpr int QV = TRANSH ? 2 : 1;
  // Q [UB][QV][P4] v4 | C [2][IB][P4] v4 | qs [UB][4] | gth [UB][GS] | gkey [UB][GS] u64
  static constexpr size_t LDS = (size_t)(UB * QV + 2 * IB) * P4 * 16 + (size_t)UB * 4 * 4 + (size_t)UB * GS * 4 + (size_t)UB * GS * 8;
};

KTUP_DEV uint64_t kg_key(float s, bool descending, uint32_t id) {      // = make_key of ktup_rank.hip
  if (descending) s = -s;
  if (s == 0.f) s = 0.f;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | id;
}

struct Unc;
struct FArgs {
  const float* QW; int dq;
  const float* C; int64_t ldc;
  int64_t nq, n_cand;
  int descending;
  const int64_t* gold_off; const int32_t* gold_ids; const float* gscore;
  const int64_t* filt_off; const int32_t* filt_ids; float* fscore;
  float* gscore_out;
  int32_t* counts; int32_t* ranks;
  int tiles_per_band;
  int gbase;                    // the sweep launch counts for golds gbase .. gbase + GS - 1 of every key
  int dbg;                      // MEASUREMENT ONLY (option dbg_eval): 1 no compares, 2 no candidate loads, 4 no barrier per stage
  float* cnorm;                 // |e|^2 of every candidate, computed ONCE per pass (kg_pass_init_kernel) and read by both kernels
  const int64_t* rel;           // mode 2: the keys' relation ids, the normals' table and w.e of every (relation, candidate)
  const float* Nrm; int64_t ldn; int n_rel;
  float* wtab; int64_t ldw;
  // exact order near the golds: the raw inputs of the fp64 re-score, the per-key window and the list of undecided comparisons
  int model, d, head;
  const float* E; int64_t lde; const float* R; int64_t ldr; const int64_t* q;
  float* ktol;                  // [nq] T of every key (kg_list_scores writes it; nullptr: no window, the keys of the scores decide)
  uint32_t* enmax;              // [1] max |e|^2 over the candidates, as float bits
  float kappa0;
  Unc* unc; int32_t* unc_count; int unc_cap, unc_nseg;   // [unc_nseg][unc_cap] entries; [unc_nseg] counts, [unc_nseg] = 1 once a segment overflowed
  int unc_sweep_segs;                                    // segments of the sweep's workgroups (blockIdx.y * NBAND + band); the finalize blocks' follow
};

// key index; the gold: its position in the key's list (a candidate the sweep did not count: +1 if it is before the gold) or ~gi, gi its
// entry (a filtered id / other gold the finalize step did not subtract: -1 if it is); candidate id; its fp32 score (unflipped)
struct Unc { int32_t key, gi, cand; float s; };

// sum_k (c_k - e_k)^2 of key `key` against candidate `cand` in fp64 from the tables (GL lanes share the row; all of them get the sum)
template <int GL>
KTUP_DEV double precise_score(const FArgs& a, int64_t key, int32_t cand, int sub) {
  const float* e = a.E + a.q[key] * a.lde;
  const float* rl = a.R + a.rel[key] * a.ldr;
  const float* c = a.C + (int64_t)cand * a.ldc;
  const double sgn = a.head ? -1.0 : 1.0;                       // head: c = proj(t) - r; tail: c = proj(h) + r  (kg_query_prep_kernel)
  double de = 0.0, dc = 0.0;
  const float* w = nullptr;
  if (a.model == KTUP_KG_TRANSH) {
    w = a.Nrm + a.rel[key] * a.ldn;
    for (int k = sub; k < a.d; k += GL) { de += (double)e[k] * (double)w[k]; dc += (double)c[k] * (double)w[k]; }
#pragma unroll
    for (int m = 1; m < GL; m <<= 1) { de += __shfl_xor(de, m, 64); dc += __shfl_xor(dc, m, 64); }
  }
  double acc = 0.0;
  for (int k = sub; k < a.d; k += GL) {
    double qv = (double)e[k], cv = (double)c[k];
    if (w) { qv -= de * (double)w[k]; cv -= dc * (double)w[k]; }
    const double z = (qv + sgn * (double)rl[k]) - cv;
    acc += z * z;
  }
#pragma unroll
  for (int m = 1; m < GL; m <<= 1) acc += __shfl_xor(acc, m, 64);
  return acc;
}
// the same for TWO candidates of one key (a listed candidate and its gold): the key's rows are read once
template <int GL>
KTUP_DEV void precise_pair(const FArgs& a, int64_t key, int32_t c0, int32_t c1, int sub, double& s0, double& s1) {
  const float* e = a.E + a.q[key] * a.lde;
  const float* rl = a.R + a.rel[key] * a.ldr;
  const float* x0 = a.C + (int64_t)c0 * a.ldc;
  const float* x1 = a.C + (int64_t)c1 * a.ldc;
  const double sgn = a.head ? -1.0 : 1.0;
  double de = 0.0, d0 = 0.0, d1 = 0.0;
  const float* w = nullptr;
  if (a.model == KTUP_KG_TRANSH) {
    w = a.Nrm + a.rel[key] * a.ldn;
    for (int k = sub; k < a.d; k += GL) {
      const double wk = (double)w[k];
      de += (double)e[k] * wk; d0 += (double)x0[k] * wk; d1 += (double)x1[k] * wk;
    }
#pragma unroll
    for (int m = 1; m < GL; m <<= 1) { de += __shfl_xor(de, m, 64); d0 += __shfl_xor(d0, m, 64); d1 += __shfl_xor(d1, m, 64); }
  }
  double a0 = 0.0, a1 = 0.0;
  for (int k = sub; k < a.d; k += GL) {
    double qv = (double)e[k], v0 = (double)x0[k], v1 = (double)x1[k];
    if (w) { const double wk = (double)w[k]; qv -= de * wk; v0 -= d0 * wk; v1 -= d1 * wk; }
    qv += sgn * (double)rl[k];
    const double z0 = qv - v0, z1 = qv - v1;
    a0 += z0 * z0; a1 += z1 * z1;
  }
#pragma unroll
  for (int m = 1; m < GL; m <<= 1) { a0 += __shfl_xor(a0, m, 64); a1 += __shfl_xor(a1, m, 64); }
  s0 = a0; s1 = a1;
}
// is candidate (sc, cid) ordered before gold (sg, gid)?  exact scores first, ids on exact ties; an unordered pair (NaN) -> the keys
KTUP_DEV bool precise_before(double sc, int32_t cid, double sg, int32_t gid, bool desc, float fsc, float fsg) {
  if (sc != sc || sg != sg) return kg_key(fsc, desc, (uint32_t)cid) < kg_key(fsg, desc, (uint32_t)gid);
  if (desc) { sc = -sc; sg = -sg; }
  return sc < sg || (sc == sg && (uint32_t)cid < (uint32_t)gid);
}

// ---- pieces shared by the list kernel and the sweep: identical code => identical bits
template <typename G>
KTUP_DEV void stage_queries(const FArgs& a, v4* Q, int64_t u0, int nthreads) {
  constexpr int NCH = G::NCH, QV = G::QV, P4 = G::P4;
  for (int idx = threadIdx.x; idx < UB * QV * NCH; idx += nthreads) {
    const int row = idx / (QV * NCH), rem = idx - row * (QV * NCH), vec = rem / NCH, c = rem - vec * NCH;
    v4 val = (v4){0.f, 0.f, 0.f, 0.f};
    if (u0 + row < a.nq) val = *reinterpret_cast<const v4*>(a.QW + ((u0 + row) * 3 + 2 * vec) * a.dq + 4 * c);
    Q[(row * QV + vec) * P4 + c] = val;
  }
}

// |c|^2, c.w, |w|^2 of a query row (w0: its normal, LDS in mode 1, the prepared query rows in global memory in mode 2): 8 lanes per
// row, chunks sub, sub + 8, ...
template <typename G>
KTUP_DEV void row_scalars(const v4* c0, const v4* w0, int sub, float& f0, float& f1, float& f2) {
  constexpr int NCH = G::NCH;
  v4 s0 = (v4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0;
  for (int c = sub; c < NCH; c += 8) {
    const v4 x0 = c0[c];
    s0 += x0 * x0;
    if (G::MODE != 0 && w0) { const v4 x1 = w0[c]; s1 += x0 * x1; s2 += x1 * x1; }
  }
  f0 = (s0[0] + s0[1]) + (s0[2] + s0[3]); f1 = (s1[0] + s1[1]) + (s1[2] + s1[3]); f2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
#pragma unroll
  for (int m = 1; m < 8; m <<= 1) { f0 += __shfl_xor(f0, m, 64); f1 += __shfl_xor(f1, m, 64); f2 += __shfl_xor(f2, m, 64); }
}

// qs[row] = the three scalars of the workgroup's 64 keys (Q staged and visible)
template <typename G>
KTUP_DEV void query_scalars(const FArgs& a, const v4* Q, float* qs, int64_t u0, int nthreads) {
  constexpr int P4 = G::P4, QV = G::QV;
  for (int row = threadIdx.x >> 3; row < UB; row += nthreads >> 3) {
    const v4* c0 = Q + row * QV * P4;
    const v4* w0 = G::TRANSH ? c0 + P4 : nullptr;
    if (G::WTAB && u0 + row < a.nq) w0 = reinterpret_cast<const v4*>(a.QW + ((u0 + row) * 3 + 2) * a.dq);
    float f0, f1, f2;
    row_scalars<G>(c0, w0, threadIdx.x & 7, f0, f1, f2);
    if ((threadIdx.x & 7) == 0) { qs[row * 4 + 0] = f0; qs[row * 4 + 1] = f1; qs[row * 4 + 2] = f2; }
  }
}

// one 16 x 16 (keys x candidates) tile: qa = the wave's query rows (+ kq), cb = its candidate rows (+ kq)
template <typename G>
KTUP_DEV void tile_dots(const v4* qa, const v4* cb, v4& ce, v4& we) {
  constexpr int KGF = G::KGF, P4 = G::P4;
  ce = (v4){0.f, 0.f, 0.f, 0.f}; we = ce;
#pragma unroll
  for (int g = 0; g < KGF; ++g) {
    const v4 ac = qa[4 * g], be = cb[4 * g];
    v4 aw = ac;
    if (G::TRANSH) aw = qa[P4 + 4 * g];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ce = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[c], be[c], ce, 0, 0, 0);
      if (G::TRANSH) we = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[c], be[c], we, 0, 0, 0);
    }
  }
  if (G::TAIL1) {
    const float* qf = reinterpret_cast<const float*>(qa - (threadIdx.x & 63) / 16 + 4 * KGF) + ((threadIdx.x & 63) >> 4);
    const float be = (reinterpret_cast<const float*>(cb - (threadIdx.x & 63) / 16 + 4 * KGF) + ((threadIdx.x & 63) >> 4))[0];
    ce = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[0], be, ce, 0, 0, 0);
    if (G::TRANSH) we = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[4 * P4], be, we, 0, 0, 0);
  }
}

// one product of the tile: the `ce` chain of tile_dots alone (the same instructions on the same operands: the same bits)
template <typename G>
KTUP_DEV v4 tile_dot1(const v4* qa, const v4* cb) {
  constexpr int KGF = G::KGF;
  v4 ce = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < KGF; ++g) {
    const v4 ac = qa[4 * g], be = cb[4 * g];
#pragma unroll
    for (int c = 0; c < 4; ++c) ce = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[c], be[c], ce, 0, 0, 0);
  }
  if (G::TAIL1) {
    const int kq = (threadIdx.x & 63) >> 4;
    const float qf = (reinterpret_cast<const float*>(qa - kq + 4 * KGF) + kq)[0];
    const float be = (reinterpret_cast<const float*>(cb - kq + 4 * KGF) + kq)[0];
    ce = __builtin_amdgcn_mfma_f32_16x16x4f32(qf, be, ce, 0, 0, 0);
  }
  return ce;
}

// the same tile with the query-side operands held in registers (the sweep re-uses them for every candidate stage): the MFMA
// sequence and its operand VALUES are those of tile_dots, so the results are the same bits
template <typename G>
struct QRegs {
  v4 ac[G::KGF], aw[G::KGF];
  float ta, tw;
  KTUP_DEV void load(const v4* qa) {
    const int kq = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int g = 0; g < G::KGF; ++g) { ac[g] = qa[4 * g]; aw[g] = G::TRANSH ? qa[G::P4 + 4 * g] : ac[g]; }
    ta = tw = 0.f;
    if (G::TAIL1) {
      const float* qf = reinterpret_cast<const float*>(qa - kq + 4 * G::KGF) + kq;
      ta = qf[0];
      if (G::TRANSH) tw = qf[4 * G::P4];
    }
  }
};

template <typename G>
KTUP_DEV void tile_dots_q(const QRegs<G>& q, const v4* cb, v4& ce, v4& we) {
  constexpr int KGF = G::KGF;
  ce = (v4){0.f, 0.f, 0.f, 0.f}; we = ce;
#pragma unroll
  for (int g = 0; g < KGF; ++g) {
    const v4 be = cb[4 * g];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ce = __builtin_amdgcn_mfma_f32_16x16x4f32(q.ac[g][c], be[c], ce, 0, 0, 0);
      if (G::TRANSH) we = __builtin_amdgcn_mfma_f32_16x16x4f32(q.aw[g][c], be[c], we, 0, 0, 0);
    }
  }
  if (G::TAIL1) {
    const int kq = (threadIdx.x & 63) >> 4;
    const float be = (reinterpret_cast<const float*>(cb - kq + 4 * KGF) + kq)[0];
    ce = __builtin_amdgcn_mfma_f32_16x16x4f32(q.ta, be, ce, 0, 0, 0);
    if (G::TRANSH) we = __builtin_amdgcn_mfma_f32_16x16x4f32(q.tw, be, we, 0, 0, 0);
  }
}

template <typename G>
KTUP_DEV float pair_score(float ce, float we, float cc, float en, float cw, float ww) {
  float score = fmaf(-2.f, ce, cc + en);
  if (G::MODE != 0) score = fmaf(we, fmaf(we, ww - 2.f, 2.f * cw), score);
  return score;
}

// ---- list scores: one wave per 16 keys; round s scores every key of the wave against the s-th entry of ITS OWN list (golds first,
// then filtered ids): a 16 x 16 tile of which only the diagonal is wanted.  4 waves = 64 keys per workgroup share the query stage.
template <typename G>
__global__ __launch_bounds__(256) void kg_list_scores_kernel(FArgs a) {
  constexpr int NCH = G::NCH, P4 = G::P4, QV = G::QV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* Q = reinterpret_cast<v4*>(smem);                         // [UB][QV][P4]
  v4* Cd = Q + UB * QV * P4;                                   // [4 waves][16][P4]
  float* qs = reinterpret_cast<float*>(Cd + 2 * IB * P4);      // [UB][4]
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t u0 = (int64_t)blockIdx.x * UB;
  stage_queries<G>(a, Q, u0, 256);
  __syncthreads();
  query_scalars<G>(a, Q, qs, u0, 256);
  __syncthreads();
  if (a.ktol && tid < UB && u0 + tid < a.nq) {                  // T = 2 kappa (|c|^2 + max |e|^2): see the file header
    const float ww = G::MODE != 0 ? qs[tid * 4 + 2] : 0.f;
    a.ktol[u0 + tid] = 2.f * a.kappa0 * (1.f + ww) * (1.f + ww) * (qs[tid * 4 + 0] + __uint_as_float(*a.enmax));
  }
  // this lane's key (for gathering: lane = (row j of the wave, chunk group))
  const int64_t key_j = u0 + 16 * w + j;
  const bool key_on = key_j < a.nq;
  const int64_t g0 = key_on ? a.gold_off[key_j] : 0, ng = key_on ? a.gold_off[key_j + 1] - g0 : 0;
  const int64_t f0_ = (key_on && a.filt_off) ? a.filt_off[key_j] : 0, nf = (key_on && a.filt_off) ? a.filt_off[key_j + 1] - f0_ : 0;
  const int64_t wrow = (G::WTAB && key_on) ? a.rel[key_j] * a.ldw : 0;
  int64_t len = ng + nf, maxlen = len;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { const int64_t o = __shfl_xor(maxlen, m, 64); maxlen = o > maxlen ? o : maxlen; }
  v4* myC = Cd + w * 16 * P4;
  const v4* qa = Q + ((16 * w + j) * QV) * P4 + kq;
  const v4* cb = myC + j * P4 + kq;
  // Round s: row j of the tile = the s-th list entry of key j (zeros when the list is shorter); 4 lanes (kq) share a row.  The ids
  // are fetched two rounds ahead and the rows one round ahead (registers), so a round does not wait out two dependent round trips
  // to memory (id, then row) before its 25 MFMAs: 84 -> 48 us per 20,480-key pass.
  constexpr int RN = (NCH + 3) / 4;
  auto list_id = [&](int64_t s) -> int32_t {
    if (s >= len) return -1;
    return s < ng ? a.gold_ids[g0 + s] : a.filt_ids[f0_ + (s - ng)];
  };
  v4 rw[RN];
  float en_n = 0.f, wt_n = 0.f;
  auto list_rows = [&](int32_t cid) {
    const bool valid = cid >= 0 && cid < a.n_cand;
#pragma unroll
    for (int k = 0; k < RN; ++k) {
      const int c = kq + 4 * k;
      rw[k] = (valid && c < NCH) ? *reinterpret_cast<const v4*>(a.C + (int64_t)cid * a.ldc + 4 * c) : (v4){0.f, 0.f, 0.f, 0.f};
    }
    en_n = valid ? a.cnorm[cid] : 0.f;                           // the same |e|^2 the sweep reads
    wt_n = (G::WTAB && valid) ? a.wtab[wrow + cid] : 0.f;         // and the same w.e
  };
  int32_t cid_cur = list_id(0), cid_nxt = list_id(1);
  list_rows(cid_cur);
  for (int64_t s = 0; s < maxlen; ++s) {
    const bool valid = cid_cur >= 0 && cid_cur < a.n_cand;
#pragma unroll
    for (int k = 0; k < RN; ++k)
      if (kq + 4 * k < NCH) myC[j * P4 + kq + 4 * k] = rw[k];
    const float en = en_n, wt = wt_n;
    const int32_t cid_nn = list_id(s + 2);
    list_rows(cid_nxt);                                          // in flight under this round's MFMAs
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    v4 ce, we;
    if constexpr (G::TRANSH) tile_dots<G>(qa, cb, ce, we);
    else { ce = tile_dot1<G>(qa, cb); we = (v4){wt, wt, wt, wt}; }
    // diagonal: key (4 kq + reg) against candidate column j  <=>  j == 4 kq + reg
    const int reg = j - 4 * kq;
    if (reg >= 0 && reg < 4 && valid) {
      const int ur = 16 * w + j;
      const float sc = pair_score<G>(ce[reg], we[reg], qs[ur * 4 + 0], en, qs[ur * 4 + 1], qs[ur * 4 + 2]);
      if (s < ng) a.gscore_out[g0 + s] = sc; else a.fscore[f0_ + (s - ng)] = sc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    cid_cur = cid_nxt; cid_nxt = cid_nn;
  }
}

// ---- mode 2: w.e of every (relation, candidate): a wave = 16 relations x 16 candidates, the sweep's `we` chain
template <typename G>
__global__ __launch_bounds__(256) void kg_wtab_kernel(FArgs a) {
  constexpr int NCH = G::NCH, P4 = G::P4;
  __shared__ __attribute__((aligned(16))) v4 W[16 * P4];
  __shared__ __attribute__((aligned(16))) v4 Cs[IB * P4];
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t i0 = (int64_t)blockIdx.x * IB;
  const int r0 = blockIdx.y * 16;
  for (int idx = tid; idx < 16 * NCH; idx += 256) {
    const int row = idx / NCH, c = idx - row * NCH;
    v4 val = (v4){0.f, 0.f, 0.f, 0.f};
    if (r0 + row < a.n_rel) {                                   // the zero padding of kg_query_prep's slot 2 (d % 4 != 0 never gets here: d = 4 NCH)
      const float* src = a.Nrm + (int64_t)(r0 + row) * a.ldn + 4 * c;
      val = (v4){src[0], src[1], src[2], src[3]};
    }
    W[row * P4 + c] = val;
  }
  for (int idx = tid; idx < IB * NCH; idx += 256) {
    const int row = idx / NCH, c = idx - row * NCH;
    Cs[row * P4 + c] = i0 + row < a.n_cand ? *reinterpret_cast<const v4*>(a.C + (i0 + row) * a.ldc + 4 * c) : (v4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const v4 we = tile_dot1<G>(W + j * P4 + kq, Cs + (16 * w + j) * P4 + kq);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg)
    if (r0 + 4 * kq + reg < a.n_rel) a.wtab[(int64_t)(r0 + 4 * kq + reg) * a.ldw + i0 + 16 * w + j] = we[reg];
}

// ---- the sweep
// KMAX: the most golds per key (in this launch's window of GS) the instantiation carries a stage loop for.  The four-gold loop keeps 16
// counters and 32 window bounds per lane beside the operands and does not fit the 128 registers a wave has at 16 waves per CU (38-96
// spilled VGPRs, depending on the width); compiled into the same kernel it sets the register count and the scratch set-up of the loops
// that DO fit.  The drivers' passes carry one to three golds per key: the host picks KMAX = 3 from the pass's largest gold list
// (run_fused) and only a pass with longer lists launches the instantiation that holds the four-gold loop.
template <typename G, int KMAX>
__global__ __launch_bounds__(NW * 64) void kg_count_mc_kernel(FArgs a) {
  constexpr int NCH = G::NCH, P4 = G::P4, QV = G::QV, GMX = GS;
  // what a lane keeps in registers across the stages (128 VGPRs per wave at 16 waves per CU): the gold scores and the three scalars
  // of its four keys, and -- at the narrow widths, one vector per key -- the query operands of its MFMAs (see QR below)
  constexpr bool QREGS = !G::TRANSH && !(G::WTAB && NCH >= 25);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* Q = reinterpret_cast<v4*>(smem);
  v4* Cd = Q + UB * QV * P4;                                    // [2][IB][P4]
  float* qs = reinterpret_cast<float*>(Cd + 2 * IB * P4);
  float* gth = qs + UB * 4;                                     // [UB][GMX] gold scores as compared (sign-flipped when descending)
  uint64_t* gkey = reinterpret_cast<uint64_t*>(gth + UB * GMX); // [UB][GMX]  (8-byte aligned: every part before is a multiple of 16)
  __shared__ int unc_n;                                         // entries of this workgroup's list segment (earlier launches' included)
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int band = blockIdx.x;                                  // consecutive workgroup ids go round the 8 XCDs: band == XCD
  const int64_t u0 = (int64_t)blockIdx.y * UB;
  const bool desc = a.descending != 0;
  const uint32_t flip = desc ? 0x80000000u : 0u;                // scores are compared as s ^ flip: ascending in every case
  if (a.gbase > 0) {                                            // a later launch: only for workgroups with a key that has such golds
    int more = 0;
    if (tid < UB && u0 + tid < a.nq) more = a.gold_off[u0 + tid + 1] - a.gold_off[u0 + tid] > a.gbase;
    if (!__syncthreads_or(more)) return;
  }
  int most = 0;
  if (tid < UB && u0 + tid < a.nq) most = (int)min((int64_t)GMX, a.gold_off[u0 + tid + 1] - a.gold_off[u0 + tid] - a.gbase);
  const bool wg_one = !__syncthreads_or(most > 1), wg_two = !__syncthreads_or(most > 2), wg_three = !__syncthreads_or(most > 3);
  stage_queries<G>(a, Q, u0, NW * 64);
  const int seg = (int)blockIdx.y * NBAND + band;
  if (tid == 0) unc_n = a.ktol ? a.unc_count[seg] : 0;
  for (int idx = tid; idx < UB * GMX; idx += NW * 64) {          // gold keys of the 64 keys (0 = no gold: no key is below it)
    const int row = idx / GMX, g = a.gbase + idx - row * GMX;
    uint64_t k = 0;
    float th = -__builtin_inff();                               // nothing is below it; a tie with it goes to the key compare (false)
    if (u0 + row < a.nq) {
      const int64_t g0 = a.gold_off[u0 + row], n = a.gold_off[u0 + row + 1] - g0;
      if (g < n) {
        k = kg_key(a.gscore[g0 + g], desc, (uint32_t)a.gold_ids[g0 + g]);
        th = __uint_as_float(__float_as_uint(a.gscore[g0 + g]) ^ flip);
      }
    }
    gkey[idx] = k; gth[idx] = th;
  }
  __syncthreads();
  query_scalars<G>(a, Q, qs, u0, NW * 64);
  __syncthreads();
  const int ut = w >> 2, it = w & 3;
  const v4* qa = Q + ((16 * ut + j) * QV) * P4 + kq;
  // Most keys of a link-prediction pass have one or two golds: a workgroup whose 64 keys have at most one / two in this launch's window
  // runs the loop with one / two compares per score instead of four (the compares are a quarter of the sweep: 921 -> 829 us with two)
  auto run = [&](auto gn_c) {
  constexpr int GN = decltype(gn_c)::value;
  // the query operands stay in registers only where they fit beside the counters and window bounds: at d >= 100 they do not for any
  // number of golds (measured on the 1-3-gold pass of the drivers: 106-122 spilled registers, the TransE sweep 2.7 ms; 0.95 ms with
  // the operands read from LDS like modes 1 / 2 do)
  constexpr bool QR = QREGS && NCH < 25 && GN <= 2;
  int cnt[4][GN];
  float lo[4][GN], hi[4][GN];                                   // a score counts below lo, is out above hi; in between: the list
  v4 qsr[4];
  int32_t wo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ur = 16 * ut + 4 * kq + r;
    const float T = (a.ktol && u0 + ur < a.nq) ? a.ktol[u0 + ur] : 0.f;
#pragma unroll
    for (int g = 0; g < GN; ++g) { cnt[r][g] = 0; lo[r][g] = gth[ur * GMX + g] - T; hi[r][g] = gth[ur * GMX + g] + T; }
    qsr[r] = *reinterpret_cast<const v4*>(qs + ur * 4);
    wo[r] = (G::WTAB && u0 + ur < a.nq) ? (int32_t)(a.rel[u0 + ur] * a.ldw) : 0;
  }
  QRegs<G> qr;
  if constexpr (QR) qr.load(qa);
  const int t0 = band * a.tiles_per_band, t1 = t0 + a.tiles_per_band;
  // a thread's share of a 64-candidate stage (IB * NCH float4 over 1024 threads: at most NLD each) and its candidate's |e|^2 (mode 2:
  // and the four w.e of its keys), fetched one stage AHEAD into registers: the loads of stage t + 1 are in flight under the matrix
  // work of stage t
  constexpr int NLD = (IB * NCH + NW * 64 - 1) / (NW * 64);
  v4 nx[NLD];
  float en_nx = 0.f, we_nx[4] = {0.f, 0.f, 0.f, 0.f};
  // BUFFER loads: a per-thread byte offset fixed for the whole sweep + the stage's offset (one 32-bit add per load), bounds checked by
  // the descriptor (rows past the band's end and stages past the sweep's read as zeros) -- no 64-bit address arithmetic or range tests
  // per stage (they were ~47 of the ~130 VALU instructions a wave issued per stage beside its 25 MFMAs)
  const int64_t band_row0 = (int64_t)t0 * IB;
  const int64_t band_rows = max((int64_t)0, min((int64_t)a.tiles_per_band * IB, a.n_cand - band_row0));
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.C + band_row0 * a.ldc), 0,
                                                                         (int)(band_rows * a.ldc * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc(a.cnorm + band_row0, 0, (int)(band_rows * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(G::WTAB ? a.wtab : a.cnorm, 0, G::WTAB ? (int)((int64_t)a.n_rel * a.ldw * 4) : 0,
                                                                         0x00020000);
  int voC[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int idx = tid + k * NW * 64;
    const int row = idx / NCH, c = idx - row * NCH;
    voC[k] = idx < IB * NCH ? (int)((row * a.ldc + 4 * c) * 4) : -16;             // (as unsigned: out of range for every stage -> zeros)
  }
  const int voN = (16 * it + j) * 4;
  int voW[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) voW[r] = (wo[r] + (int)band_row0 + 16 * it + j) * 4;
  const int stepC = (int)(IB * a.ldc * 4);
  auto fetch = [&](int s) {                                     // s = stage index inside the band (uniform)
    const int soC = s * stepC, soN = s * IB * 4;
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      nx[k] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rsC, voC[k] < 0 ? voC[k] : voC[k] + soC, 0, 0));
    en_nx = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsN, voN + soN, 0, 0));
    if constexpr (G::WTAB) {
#pragma unroll
      for (int r = 0; r < 4; ++r) we_nx[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsW, voW[r] + soN, 0, 0));
    }
  };
  fetch(0);
  for (int t = t0; t < t1; ++t) {
    const int64_t i0 = (int64_t)t * IB;
    if (i0 >= a.n_cand) break;
    v4* Cs = Cd + ((t - t0) & 1) * IB * P4;                     // the other half was read in stage t - 1: every wave is past that barrier
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + k * NW * 64;
      if (idx < IB * NCH) Cs[(idx / NCH) * P4 + (idx % NCH)] = nx[k];
    }
    const float ee = en_nx;
    float wl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wl[r] = we_nx[r];
    if (!(a.dbg & 2)) fetch(t + 1 - t0);
    if (!(a.dbg & 4)) __syncthreads();
    const v4* cb = Cs + (16 * it + j) * P4 + kq;
    v4 ce, we;
    if constexpr (G::TRANSH) tile_dots<G>(qa, cb, ce, we);      // two query vectors per key do not fit the registers (measured: spills)
    else {
      if constexpr (QR) tile_dots_q<G>(qr, cb, ce, we); else ce = tile_dot1<G>(qa, cb);
      we = (v4){wl[0], wl[1], wl[2], wl[3]};
    }
    const int64_t cand = i0 + 16 * it + j;
    if (a.dbg & 1) { if (ce[0] + we[1] == 12345.f) cnt[0][0] += 1; continue; }
    if (cand < a.n_cand) {
      float s[4];
      bool tie = false;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const v4 qv = qsr[reg];
        s[reg] = __uint_as_float(__float_as_uint(pair_score<G>(ce[reg], we[reg], qv[0], ee, qv[1], qv[2])) ^ flip);
#pragma unroll
        for (int g = 0; g < GN; ++g) {
          const bool lt = s[reg] < lo[reg][g];
          cnt[reg][g] += lt ? 1 : 0;
          tie |= !lt && !(s[reg] > hi[reg][g]);                 // inside the window, or unordered: decided elsewhere
        }
      }
      if (__builtin_amdgcn_ballot_w64(tie) != 0) {               // rare: the gold itself (once per band), near ties, NaNs
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ur = 16 * ut + 4 * kq + reg;
#pragma unroll
          for (int g = 0; g < GN; ++g) {
            if (s[reg] < lo[reg][g] || s[reg] > hi[reg][g]) continue;
            const uint64_t gk = gkey[ur * GMX + g];
            if (gk == 0 || (uint32_t)gk == (uint32_t)cand) continue;          // no such gold; the gold itself is not before itself
            const int slot = a.ktol ? atomicAdd(&unc_n, 1) : a.unc_cap;      // an LDS counter: the segment is this workgroup's alone
            if (slot < a.unc_cap) {
              a.unc[(int64_t)seg * a.unc_cap + slot] = Unc{(int32_t)(u0 + ur), a.gbase + g, (int32_t)cand, __uint_as_float(__float_as_uint(s[reg]) ^ flip)};
            } else {                                             // no window / the segment is full: the keys of the sweep's own scores decide
              if (a.ktol) a.unc_count[a.unc_nseg] = 1;
              cnt[reg][g] += kg_key(s[reg], false, (uint32_t)cand) < gk ? 1 : 0;      // s is already flipped
            }
          }
        }
      }
    }
  }
  // 16 candidate lanes (j) of a key -> one sum; the lanes with j == 0 flush
#pragma unroll
  for (int reg = 0; reg < 4; ++reg)
#pragma unroll
    for (int g = 0; g < GN; ++g) {
      int c = cnt[reg][g];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) c += __shfl_xor(c, m, 64);
      cnt[reg][g] = c;
    }
  if (j == 0) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t key = u0 + 16 * ut + 4 * kq + reg;
      if (key < a.nq) {
        const int64_t g0 = a.gold_off[key], n = a.gold_off[key + 1] - g0;
#pragma unroll
        for (int g = 0; g < GN; ++g)
          if (a.gbase + g < n && cnt[reg][g] != 0) atomicAdd(a.counts + g0 + a.gbase + g, cnt[reg][g]);
      }
    }
  }
  };
  if (wg_one) run(std::integral_constant<int, 1>{});
  else if (wg_two) run(std::integral_constant<int, 2>{});
  else if (KMAX <= 3 || wg_three) run(std::integral_constant<int, 3>{});
  else {
    if constexpr (KMAX > 3) run(std::integral_constant<int, GMX>{});
  }
  if (a.ktol) {
    __syncthreads();
    if (tid == 0) a.unc_count[seg] = unc_n < a.unc_cap ? unc_n : a.unc_cap;
  }
}

// ---- per gold entry: rank = count - (filtered ids and other golds of the key that are ordered before it); -1 if itself filtered.
// The drivers' filter and gold lists are sorted sets; a key whose lists are strictly increasing (checked in one pass) needs no
// duplicate tests -- membership of a gold in the filter list is a binary search, "another gold that is also filtered" likewise.
// Lists in any other order take the quadratic tests (exact for multisets): that form was 63 us per 20,480-key pass for every key.
KTUP_DEV bool in_sorted(const int32_t* __restrict__ a, int64_t lo, int64_t hi, int32_t x) {
  const int64_t end = hi;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo < end && a[lo] == x;
}
// the listed comparisons, 16 lanes per entry, a workgroup per segment: fp64 from the tables; after an overflow the keys of the fp32
// scores for all of them.  counts[gi] < 0 marks a gold that is itself filtered (rank -1: nothing to adjust).
__global__ __launch_bounds__(256) void kg_unc_resolve_kernel(FArgs a) {
  const bool desc = a.descending != 0;
  const bool approx = a.unc_count[a.unc_nseg] != 0;
  const int sub = threadIdx.x & 15;
  for (int seg = blockIdx.x; seg < a.unc_nseg; seg += gridDim.x) {
    const int n = min(a.unc_count[seg], a.unc_cap);
    for (int i = threadIdx.x >> 4; i < n; i += 16) {
      const Unc u = a.unc[(int64_t)seg * a.unc_cap + i];
      const int32_t gi = u.gi < 0 ? ~u.gi : (int32_t)a.gold_off[u.key] + u.gi;
      if (a.counts[gi] < 0) continue;
      const int32_t gid = a.gold_ids[gi];
      bool before;
      if (approx) {
        before = kg_key(u.s, desc, (uint32_t)u.cand) < kg_key(a.gscore[gi], desc, (uint32_t)gid);
      } else {
        double sc, sg;
        precise_pair<16>(a, u.key, u.cand, gid, sub, sc, sg);
        before = precise_before(sc, u.cand, sg, gid, desc, u.s, a.gscore[gi]);
      }
      if (sub == 0 && before) atomicAdd(a.ranks + gi, u.gi < 0 ? -1 : 1);
    }
  }
}

__global__ __launch_bounds__(256) void kg_rank_finalize_kernel(FArgs a, int64_t n_gold_total) {
  const bool desc = a.descending != 0;
  const bool windowed = a.ktol != nullptr;
  const int seg = a.unc_sweep_segs + (int)blockIdx.x;
  for (int64_t key = (int64_t)blockIdx.x * 256 + threadIdx.x; key < a.nq; key += (int64_t)gridDim.x * 256) {
    const int64_t g0 = a.gold_off[key], g1 = a.gold_off[key + 1];
    const int64_t f0 = a.filt_off ? a.filt_off[key] : 0, f1 = a.filt_off ? a.filt_off[key + 1] : 0;
    bool sorted = true;
    for (int64_t f = f0 + 1; f < f1 && sorted; ++f) sorted = a.filt_ids[f - 1] < a.filt_ids[f];
    for (int64_t o = g0 + 1; o < g1 && sorted; ++o) sorted = a.gold_ids[o - 1] < a.gold_ids[o];
    for (int64_t gi = g0; gi < g1; ++gi) {
      const int32_t gid = a.gold_ids[gi];
      bool filtered = gid < 0 || gid >= a.n_cand;
      if (!filtered) {
        if (sorted) filtered = in_sorted(a.filt_ids, f0, f1, gid);
        else for (int64_t f = f0; f < f1 && !filtered; ++f) filtered = a.filt_ids[f] == gid;
      }
      if (filtered) { a.ranks[gi] = -1; a.counts[gi] = -1; continue; }
      const uint64_t gk = kg_key(a.gscore[gi], desc, (uint32_t)gid);
      const float gs = a.gscore[gi], T = windowed ? a.ktol[key] : 0.f;
      // list entry (score fs, id c) before this gold?  outside the window the float scores decide; inside it the comparison joins the
      // sweep's list (sign -1) and kg_unc_resolve_kernel decides it
      auto before = [&](float fs, int32_t c) {
        if (!windowed || fs < gs - T || fs > gs + T) return kg_key(fs, desc, (uint32_t)c) < gk;
        const int slot = atomicAdd(a.unc_count + seg, 1);
        if (slot < a.unc_cap) { a.unc[(int64_t)seg * a.unc_cap + slot] = Unc{(int32_t)key, ~(int32_t)gi, c, fs}; return false; }
        a.unc_count[a.unc_nseg] = 1;                           // full: the whole pass falls back to the keys (the resolve step reads this)
        return kg_key(fs, desc, (uint32_t)c) < gk;
      };
      int sub = 0;
      if (sorted) {
#pragma unroll 8
        for (int64_t f = f0; f < f1; ++f) {                   // independent loads: eight in flight
          const int32_t c = a.filt_ids[f];
          sub += (c >= 0 && c < a.n_cand && before(a.fscore[f], c)) ? 1 : 0;
        }
      } else {
        for (int64_t f = f0; f < f1; ++f) {
          const int32_t c = a.filt_ids[f];
          if (c < 0 || c >= a.n_cand) continue;
          bool dup = false;                                    // a filter list is a set, but stay exact if it is not
          for (int64_t e = f0; e < f && !dup; ++e) dup = a.filt_ids[e] == c;
          if (!dup && before(a.fscore[f], c)) ++sub;
        }
      }
      for (int64_t o = g0; o < g1; ++o) {
        const int32_t c = a.gold_ids[o];
        if (o == gi || c < 0 || c >= a.n_cand) continue;
        bool dup = false;                                      // another gold that is also filtered was counted above
        if (sorted) dup = in_sorted(a.filt_ids, f0, f1, c);
        else {
          for (int64_t f = f0; f < f1 && !dup; ++f) dup = a.filt_ids[f] == c;
          for (int64_t e = g0; e < o && !dup; ++e) dup = a.gold_ids[e] == c;
        }
        if (!dup && before(a.gscore[o], c)) ++sub;
      }
      a.ranks[gi] = a.counts[gi] - sub;
      a.counts[gi] = 0;
    }
  }
}

// counts <- 0 and |e|^2 of every candidate (8 lanes per row; ONE value per candidate for the whole pass)
__global__ __launch_bounds__(256) void kg_unc_reset_kernel(uint32_t* enmax, int32_t* unc_count, int nseg) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *enmax = 0u;
  for (int i = blockIdx.x * 256 + threadIdx.x; i <= nseg; i += gridDim.x * 256) unc_count[i] = 0;
}

__global__ __launch_bounds__(256) void kg_pass_init_kernel(int32_t* counts, int64_t n, const float* __restrict__ C, int64_t ldc, int nch,
                                                           int64_t n_cand, float* __restrict__ cnorm, uint32_t* enmax) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) counts[i] = 0;
  float mx = 0.f;
  for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; row < n_cand; row += ((int64_t)gridDim.x * 256) >> 3) {
    const v4* r0 = reinterpret_cast<const v4*>(C + row * ldc);
    v4 s0 = (v4){0.f, 0.f, 0.f, 0.f};
    for (int c = threadIdx.x & 7; c < nch; c += 8) { const v4 x0 = r0[c]; s0 += x0 * x0; }
    float f0 = (s0[0] + s0[1]) + (s0[2] + s0[3]);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) f0 += __shfl_xor(f0, m, 64);
    if ((threadIdx.x & 7) == 0) cnorm[row] = f0;
    mx = f0 > mx ? f0 : mx;                                     // (a NaN norm does not raise the window: its rows tie by keys anyway)
  }
  if (enmax) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const float o = __shfl_xor(mx, m, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(enmax, __float_as_uint(mx));     // non-negative floats order like their bits
  }
}

// undecided comparisons: one list segment per sweep workgroup (64 keys x one band of candidates: ~1e-4 of its 120 k pairs per gold
// land in a window) and per finalize block (256 keys' filter lists), UNC_CAP entries each
constexpr int UNC_CAP = 1024;
int fin_blocks(int64_t nq) { return grid_for((nq + 255) / 256, 1024); }
int64_t unc_sweep_segs(int64_t nq) { return ((nq + UB - 1) / UB) * NBAND; }
int64_t unc_segs(int64_t nq) { return unc_sweep_segs(nq) + fin_blocks(nq); }

template <typename G>
int run_fused(FArgs a, int64_t n_gold, int64_t max_golds, hipStream_t st, const char* name) {
  (void)hipFuncSetAttribute((const void*)kg_list_scores_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
  const unsigned qblocks = (unsigned)((a.nq + UB - 1) / UB);
  const int64_t init_work = n_gold > (a.n_cand * 8) ? n_gold : a.n_cand * 8;
  hipLaunchKernelGGL(kg_unc_reset_kernel, dim3(grid_for((a.unc_nseg + 256) / 256, 64)), dim3(256), 0, st, a.enmax, a.unc_count, a.unc_nseg);
  hipLaunchKernelGGL(kg_pass_init_kernel, dim3(grid_for((init_work + 255) / 256, 2048)), dim3(256), 0, st, a.counts, n_gold, a.C, a.ldc, G::NCH,
                     a.n_cand, a.cnorm, a.enmax);
  const int64_t ntiles = (a.n_cand + IB - 1) / IB;
  if constexpr (G::WTAB) hipLaunchKernelGGL((kg_wtab_kernel<G>), dim3((unsigned)ntiles, (unsigned)((a.n_rel + 15) / 16)), dim3(256), 0, st, a);
  a.gscore_out = const_cast<float*>(a.gscore);
  hipLaunchKernelGGL((kg_list_scores_kernel<G>), dim3(qblocks), dim3(256), G::LDS, st, a);
  a.tiles_per_band = (int)((ntiles + NBAND - 1) / NBAND);
  (void)hipFuncSetAttribute((const void*)kg_count_mc_kernel<G, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
  (void)hipFuncSetAttribute((const void*)kg_count_mc_kernel<G, GS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
  for (a.gbase = 0; a.gbase < max_golds; a.gbase += GS) {   // typical link-prediction keys have one to three golds: one launch; the
    if (max_golds - a.gbase <= 3)                           // workgroups of a later one whose keys have no such golds exit
      hipLaunchKernelGGL((kg_count_mc_kernel<G, 3>), dim3(NBAND, qblocks), dim3(NW * 64), G::LDS, st, a);
    else
      hipLaunchKernelGGL((kg_count_mc_kernel<G, GS>), dim3(NBAND, qblocks), dim3(NW * 64), G::LDS, st, a);
  }
  a.gbase = 0;
  hipLaunchKernelGGL(kg_rank_finalize_kernel, dim3(fin_blocks(a.nq)), dim3(256), 0, st, a, n_gold);
  if (a.ktol) hipLaunchKernelGGL(kg_unc_resolve_kernel, dim3(grid_for(a.unc_nseg, 4096)), dim3(256), 0, st, a);
  return check_launch(name);
}

template <int MODE>
int dispatch_fused(const FArgs& a, int d, int64_t n_gold, int64_t max_golds, hipStream_t st, const char* name) {
  switch (d) {
    case 20: return run_fused<FGeom<5, MODE>>(a, n_gold, max_golds, st, name);
    case 36: return run_fused<FGeom<9, MODE>>(a, n_gold, max_golds, st, name);
    case 64: return run_fused<FGeom<16, MODE>>(a, n_gold, max_golds, st, name);
    case 100: return run_fused<FGeom<25, MODE>>(a, n_gold, max_golds, st, name);
    case 128: return run_fused<FGeom<32, MODE>>(a, n_gold, max_golds, st, name);
    default: return 1;
  }
}

size_t pad256(size_t n) { return (n + 255) & ~(size_t)255; }


// mode 2's table: pitch (whole candidate stages) and whether it is used at all -- offsets stay 32-bit, the table below 1 GiB
int64_t wtab_pitch(int64_t n_cand) { return ((n_cand + IB - 1) / IB) * IB; }
bool wtab_on(int model, int64_t n_cand, int64_t n_rel) {
  return model == KTUP_KG_TRANSH && n_rel > 0 && opt_kg_wtab() && n_rel * wtab_pitch(n_cand) <= (int64_t)(1ll << 28);
}

}  // namespace
}  // namespace ktup

// the matrix-core sweep applies; everything else TransE / TransH takes the VALU count route (ktup_eval.hip kg_valu_counts)
static bool fused_mfma(int d, int l1, int64_t max_golds) {
  (void)max_golds;      // (golds beyond the first four of a key: further sweep launches whose other workgroups exit at once)
  return !l1 && (d == 20 || d == 36 || d == 64 || d == 100 || d == 128) && ktup::opt_eval_mc();
}

extern "C" int ktup_eval_kg_ranks_fused_supported(int model, int d, int l1, int64_t max_golds) {
  (void)l1; (void)max_golds;
  return (model == KTUP_KG_TRANSE || model == KTUP_KG_TRANSH) && d > 0 && d <= 1024;
}

extern "C" size_t ktup_eval_kg_ranks_fused_workspace_bytes(int model, int d, int64_t nq, int64_t n_gold, int64_t n_filt, int64_t n_cand,
                                                           int64_t n_rel) {
  if (d <= 0 || nq <= 0 || n_cand <= 0) return 0;
  return ktup::pad256(ktup_eval_kg_workspace_bytes(d, nq)) + ktup::pad256((size_t)(n_gold > 0 ? n_gold : 1) * 4) * 2 +
         ktup::pad256((size_t)(n_filt > 0 ? n_filt : 1) * 4) + ktup::pad256((size_t)n_cand * 4) + ktup::pad256((size_t)nq * 4) + 256 +
         ktup::pad256((size_t)(ktup::unc_segs(nq) + 1) * 4) + ktup::pad256((size_t)ktup::unc_segs(nq) * ktup::UNC_CAP * sizeof(ktup::Unc)) +
         (ktup::wtab_on(model, n_cand, n_rel) ? ktup::pad256((size_t)n_rel * ktup::wtab_pitch(n_cand) * 4) : 0);
}

extern "C" int ktup_eval_kg_ranks_fused(int model, const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                                        int64_t n_rel, int d, const float* C, int64_t ldc, int64_t n_cand, const int64_t* q, const int64_t* r,
                                        int64_t nq, int l1, int head, int descending, const int64_t* filt_off, const int32_t* filt_ids,
                                        int64_t n_filt, const int64_t* gold_off, const int32_t* gold_ids, int64_t n_gold, int64_t max_golds,
                                        int32_t* ranks, void* ws, void* stream) {
  const char* name = "ktup_eval_kg_ranks_fused";
  using namespace ktup;
  if (!ktup_eval_kg_ranks_fused_supported(model, d, l1, max_golds))
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: TransE / TransH (TransR: ktup_eval_kg_ranks_transr)", name);
  KTUP_REQUIRE(nq >= 0 && n_cand > 0 && n_gold >= 0 && n_filt >= 0 && n_rel >= 0, "%s: bad sizes", name);
  if (nq == 0 || n_gold == 0) return KTUP_OK;
  KTUP_REQUIRE(E && R && C && q && r && gold_off && gold_ids && ranks && ws && (model == KTUP_KG_TRANSE || Nrm), "%s: null pointer argument", name);
  KTUP_REQUIRE((filt_off == nullptr) || filt_ids, "%s: filter offsets without ids", name);
  KTUP_REQUIRE(n_cand < (1ll << 31) && (nq + UB - 1) / UB <= 65535, "%s: sizes out of range", name);
  const bool mfma = fused_mfma(d, l1, max_golds) && aligned16(C) && (ldc & 3) == 0 &&
                    ((n_cand + IB * NBAND - 1) / (IB * NBAND) + 1) * IB * ldc * 4 < (1ll << 31);   // a band goes through one 32-bit buffer descriptor
  hipStream_t st = (hipStream_t)stream;
  char* p = reinterpret_cast<char*>(ws);
  float* QW = reinterpret_cast<float*>(p); p += pad256(ktup_eval_kg_workspace_bytes(d, nq));
  float* gscore = reinterpret_cast<float*>(p); p += pad256((size_t)n_gold * 4);
  int32_t* counts = reinterpret_cast<int32_t*>(p); p += pad256((size_t)n_gold * 4);
  float* fscore = reinterpret_cast<float*>(p); p += pad256((size_t)(n_filt > 0 ? n_filt : 1) * 4);
  float* cnorm = reinterpret_cast<float*>(p); p += pad256((size_t)n_cand * 4);
  float* ktol = reinterpret_cast<float*>(p); p += pad256((size_t)nq * 4);
  uint32_t* enmax = reinterpret_cast<uint32_t*>(p); p += 256;
  int32_t* unc_count = reinterpret_cast<int32_t*>(p); p += pad256((size_t)(unc_segs(nq) + 1) * 4);
  Unc* unc = reinterpret_cast<Unc*>(p); p += pad256((size_t)unc_segs(nq) * UNC_CAP * sizeof(Unc));
  if (int e = kg_query_prep(model, E, lde, R, ldr, Nrm, ldn, d, q, r, nq, head, QW, st, name)) return e;
  FArgs a{};
  a.QW = QW; a.dq = (d + 3) & ~3; a.C = C; a.ldc = ldc; a.nq = nq; a.n_cand = n_cand; a.descending = descending;
  a.gold_off = gold_off; a.gold_ids = gold_ids; a.gscore = gscore; a.filt_off = filt_off; a.filt_ids = filt_ids; a.fscore = fscore;
  a.counts = counts; a.ranks = ranks; a.cnorm = cnorm; a.dbg = opt_dbg_eval();
  a.model = model; a.d = d; a.head = head; a.E = E; a.lde = lde; a.R = R; a.ldr = ldr; a.q = q; a.rel = r; a.Nrm = Nrm; a.ldn = ldn;
  a.enmax = enmax; a.unc = unc; a.unc_count = unc_count; a.unc_cap = UNC_CAP; a.unc_nseg = (int)unc_segs(nq);
  a.unc_sweep_segs = (int)unc_sweep_segs(nq);
  a.kappa0 = 2.f * (float)(d + 8) * 5.9604645e-8f;             // 2 (d + 8) 2^-24
  a.ktol = (mfma && opt_kg_exact()) ? ktol : nullptr;          // the VALU routes score by the reference's own formula: no window
  if (!mfma) {
    // L1, widths without a matrix-core instantiation, keys with more than 8 golds: the pair kernels of ktup_eval.hip score the tiles
    // on the VALU and count where the scores are made (no score matrix either); list scores by the same function
    hipLaunchKernelGGL(kg_pass_init_kernel, dim3(grid_for((n_gold + 255) / 256, 2048)), dim3(256), 0, st, counts, n_gold, C, ldc, 0, (int64_t)0, cnorm, (uint32_t*)nullptr);
    const bool tab = wtab_on(model, n_cand, n_rel);   // (the workspace holds the table exactly then)
    if (int e = kg_valu_counts(model, QW, d, C, ldc, n_cand, nq, l1, descending, gold_off, gold_ids, filt_off, filt_ids, gscore, fscore, counts,
                               tab ? r : nullptr, Nrm, ldn, n_rel, tab ? reinterpret_cast<float*>(p) : nullptr, wtab_pitch(n_cand), st, name))
      return e;
    hipLaunchKernelGGL(kg_rank_finalize_kernel, dim3(grid_for((nq + 255) / 256, 1024)), dim3(256), 0, st, a, n_gold);
    return check_launch(name);
  }
  int rc;
  if (model == KTUP_KG_TRANSE) rc = dispatch_fused<0>(a, d, n_gold, max_golds, st, name);
  else if (wtab_on(model, n_cand, n_rel)) {      // relation ids are bounds-checked by the caller's tables (r indexes R and Nrm already)
    a.rel = r; a.Nrm = Nrm; a.ldn = ldn; a.n_rel = (int)n_rel; a.wtab = reinterpret_cast<float*>(p); a.ldw = wtab_pitch(n_cand);
    rc = dispatch_fused<2>(a, d, n_gold, max_golds, st, name);
  } else rc = dispatch_fused<1>(a, d, n_gold, max_golds, st, name);
  if (rc == 1) return set_error(KTUP_ERR_UNSUPPORTED, "%s: d=%d is not an instantiated width", name, d);
  return rc;
}
