// K11-K16: all-candidate scoring for evaluation (every item for a batch of users / every entity for a batch of
// (h,r) or (t,r) queries).  reference: bprmf.py:51-54, transE.py:65-105, transH.py:73-121, transR.py:80-128,
// transUP.py:84-102, jTransUP.py:163-247.
//
// The reference materialises several (B x N x d) tensors per call (663 MB each at B=512, N=3240, d=100) and runs
// the per-pair P x d contractions B*N times.  Here:
//   * the preference gate is decomposed: logits(b,j) = LU_b + LV_j, and in the soft gate r and n are linear in the
//     logits, so r(b,j) = RU_b + RV_j and n(b,j) = NU_b + NV_j.  The P x d contractions run once per user and once
//     per item (pref_project_kernel); the pair kernel is purely elementwise:
//         s = (u_b - v_j).(NU_b + NV_j),   z = (u_b + RU_b) - (v_j - RV_j) - s (NU_b + NV_j)
//   * TransH: z = c_b - e_j + (e_j.w_b) w_b   and TransE: z = c_b - e_j   use the same pair kernel;
//   * BPRMF's U[u].I^T is the one real GEMM: fp32-input MFMA (v_mfma_f32_32x32x2_f32), exact fp32 fma chains.
// Pair-kernel mapping: lane <-> candidate (64-candidate tile staged once in LDS as [k/4][lane] float4, conflict-free
// ds_read_b128); the 4 waves of a workgroup take different queries, QB = 4 queries at a time, whose vectors are
// wave-uniform and therefore come through scalar loads into SGPRs.  The (B x N) score row is written coalesced.
// These kernels are bound by the vector pipe's issue rate, so the arithmetic is written on float pairs (v_pk_add_f32 / v_pk_fma_f32).
// Whole-pass forms of the same arithmetic (no score matrix): the COUNT form of the pair kernel + pairs_list_kernel (link prediction:
// counts of candidates ordered before each gold; TransH reads e . w from a per-pass table, pairs_wtab_kernel), sweep_soft_kernel
// (soft gate, L1) and sweep_hard_kernel (ST-Gumbel gate) with the filtered top-n in their epilogues.
#include <cstdlib>

#include "ktup_pref_geom.h"

using namespace ktup;

namespace {

constexpr int CT = 64;  // candidates per workgroup tile
constexpr int QB = 4;   // queries a wave scores together (amortises each LDS read over QB queries)

KTUP_DEV float wave_sum(float v) { return group_sum<64>(v); }

// ---------------------------------------------------------------------------------------------------------
// Query-side vectors for the KG models.  QW[b][3][dq]: slot 0 = c_b (translated query), slot 2 = w_b (TransH).
// model: 0 TransE, 1 TransH, 2 TransR.   One wave per query.
__global__ __launch_bounds__(256) void kg_query_prep_kernel(int model, const float* __restrict__ E, int64_t lde,
                                                            const float* __restrict__ R, int64_t ldr,
                                                            const float* __restrict__ X, int64_t ldx, int d, int dq,
                                                            const int64_t* __restrict__ q, const int64_t* __restrict__ r,
                                                            int64_t nq, int head, float* __restrict__ QW) {
  const int lane = threadIdx.x & 63;
  const float sgn = head ? -1.f : 1.f;  // head: c = proj(t) - r ; tail: c = proj(h) + r
  for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < nq; b += (int64_t)gridDim.x * 4) {
    const float* e = E + q[b] * lde;
    const float* rel = R + r[b] * ldr;
    float* out = QW + b * 3 * dq;
    if (model == 0) {
      for (int k = lane; k < dq; k += 64) out[k] = k < d ? e[k] + sgn * rel[k] : 0.f;
    } else if (model == 1) {
      const float* w = X + r[b] * ldx;
      float dot = 0.f;
      for (int k = lane; k < d; k += 64) dot = fmaf(e[k], w[k], dot);
      dot = wave_sum(dot);
      for (int k = lane; k < dq; k += 64) {
        out[k] = k < d ? (e[k] - dot * w[k]) + sgn * rel[k] : 0.f;
        out[2 * dq + k] = k < d ? w[k] : 0.f;
      }
    } else {
      const float* M = X + r[b] * ldx;  // (d x d) row-major: out_i = sum_k M[i][k] e[k]   (misc.py:21-26)
      for (int i = lane; i < dq; i += 64) {
        float acc = 0.f;
        if (i < d) {
          for (int k = 0; k < d; ++k) acc = fmaf(M[(int64_t)i * d + k], e[k], acc);
          acc += sgn * rel[i];
        }
        out[i] = acc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Preference projections of one table (users or items): logits L = (A x)/2, R = beta A^T L, N = beta C^T L.
// Outputs (pitch d): O0 = x + sign*R, O1 = x, O2 = N, OL = L (n_pref per row), ON = x . Cn_p (n_pref per row; NULL = not wanted: the
// hard gate's squared-L2 score reads it).  One wave per row, lane = 16-B chunk.
__global__ __launch_bounds__(256) void pref_project_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ E,
                                                           int64_t lde, const int32_t* __restrict__ item2ent,
                                                           const int64_t* __restrict__ ids, int64_t nrows, int d, int P,
                                                           const float* __restrict__ ws, int ppad, int dp, float sign,
                                                           int64_t opitch, float* __restrict__ O0, float* __restrict__ O1,
                                                           float* __restrict__ O2, float* __restrict__ OL, float* __restrict__ ON) {
  const int lane = threadIdx.x & 63;
  const int nch = d / 4;
  const float4* Alog = reinterpret_cast<const float4*>(ws);
  const float4* Ar = reinterpret_cast<const float4*>(ws + (size_t)ppad * dp);
  const float4* Cn = reinterpret_cast<const float4*>(ws + (size_t)(ppad + P) * dp);
  const int dp4 = dp / 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += (int64_t)gridDim.x * 4) {
    const int64_t src = ids ? ids[row] : row;
    float4 x = f4zero();
    if (lane < nch) {
      x = reinterpret_cast<const float4*>(X + src * ldx)[lane];
      if (E) x = x + reinterpret_cast<const float4*>(E + (int64_t)item2ent[src] * lde)[lane];
    }
    float4 racc = f4zero(), nacc = f4zero();
    for (int pb = 0; pb < P; pb += 64) {          // up to 64 logits per pass: lane l keeps logit pb + l
      const int pe = min(P, pb + 64);
      float mylog = 0.f, mydn = 0.f;
      // phase 1: the logits.  Four preferences per trip = four INDEPENDENT cross-lane reductions in flight (one dependent
      // reduction per preference, as a single loop had it, is a chain of ~20 x 6 shuffle latencies per row)
      for (int p = pb; p < pe; p += 4) {
        float part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) part[k] = (lane < nch && p + k < pe) ? dot4(x, Alog[min(p + k, pe - 1) * dp4 + lane]) : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) part[k] = wave_sum(part[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (lane == p + k - pb) mylog = part[k];
        if (ON) {
#pragma unroll
          for (int k = 0; k < 4; ++k) part[k] = (lane < nch && p + k < pe) ? dot4(x, Cn[min(p + k, pe - 1) * dp4 + lane]) : 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) part[k] = wave_sum(part[k]);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (lane == p + k - pb) mydn = part[k];
        }
      }
      if (pb + lane < pe) OL[row * P + pb + lane] = mylog;
      if (ON && pb + lane < pe) ON[row * P + pb + lane] = mydn;
      // phase 2: R and N accumulate over the preferences; logit p comes from lane p - pb (uniform index -> v_readlane)
      for (int p = pb; p < pe; ++p) {
        const float lp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mylog), p - pb));
        if (lane < nch) {
          racc = fma4(lp, Ar[p * dp4 + lane], racc);
          nacc = fma4(lp, Cn[p * dp4 + lane], nacc);
        }
      }
    }
    if (lane < nch) {
      reinterpret_cast<float4*>(O0 + row * opitch)[lane] = fma4(sign, racc, x);
      reinterpret_cast<float4*>(O1 + row * opitch)[lane] = x;
      reinterpret_cast<float4*>(O2 + row * opitch)[lane] = nacc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
struct PairsArgs {
  const float *C0, *C1, *C2;  // candidate-side vectors: C0 enters z, C1 enters s, C2 = candidate part of the normal
  int64_t ldc0, ldc1, ldc2;
  const float* QW;            // [nq][3][dq] : A, Q1, NQ
  int64_t n_cand, nq;
  int d, dq, l1;
  float* out;
  int64_t ldo;
  int cvec;                   // candidate rows readable as aligned float4
  // TransR only: candidates depend on the query's relation -> grid.z = relation, queries bucketed by relation
  const int32_t* qperm;       // query order sorted by relation (NULL otherwise)
  const int32_t* rel_off;     // [n_rel + 1] bucket offsets into qperm
  int64_t rel_stride;         // elements between the projected-candidate tables of consecutive relations
  // COUNT kernels (the link-prediction pass without the score matrix, any distance): per gold entry of a query the number of
  // candidates ordered before it, from the scores where they are made; LIST kernel: the scores of the queries' own list entries
  const int64_t *gold_off, *filt_off;
  const int32_t *gold_ids, *filt_ids;
  float *gscore, *fscore;
  int32_t* counts;
  int descending;
  // MODE 3 (TransH with the pass's table of e . w): the keys' relation ids, the table [n_rel][ldw] of s = -(e . w_rel) in
  // pair_group_scores' own operation order (transh_dots), and -- for the kernel that fills it -- the relation normals
  const int64_t* rel;
  float* wtab;
  int64_t ldw;
  const float* Nrm;
  int64_t ldn;
  int n_rel;
};

KTUP_DEV float4 load_cand4(const float* base, int64_t ld, int64_t row, int c, int d, bool vec) {
  const float* p = base + row * ld + 4 * c;
  if (vec) return *reinterpret_cast<const float4*>(p);
  float4 v;
  v.x = 4 * c + 0 < d ? p[0] : 0.f;
  v.y = 4 * c + 1 < d ? p[1] : 0.f;
  v.z = 4 * c + 2 < d ? p[2] : 0.f;
  v.w = 4 * c + 3 < d ? p[3] : 0.f;
  return v;
}

// MODE 0: z = A - C0 (TransE / TransR-projected)   MODE 1: TransH   MODE 2: TUP / KTUP soft gate   MODE 3: TransH, e . w from the pass's table
// L1: the distance kind is compile-time -- a run-time flag makes the compiler evaluate |z| AND z^2 per element and select.
// MODE 2 keeps three candidate vectors in LDS (77 KB at d = 100: two workgroups per CU), so its workgroups are 8 waves.
template <int MODE>
struct PairsWG { static constexpr int NWV = MODE == 2 ? 8 : 4, NT = NWV * 64; };

// the scores of NQ queries (wave-uniform vectors through scalar loads) against the lane's candidate of the staged tile.  ONE function
// for the score kernel, its COUNT form and the list kernel: the same instructions on the same operands, so a (query, candidate) pair
// has the same bits wherever it is scored
// The arithmetic is written on float PAIRS (v_pk_add_f32 / v_pk_fma_f32: two elements per ~5-clock issue against one per 4): these
// kernels are bound by the vector pipe's issue rate, and every operand pair is naturally aligned (float4 chunks from s_load_dwordx4 /
// ds_read_b128).  Dot products keep one partial sum per pair half; sum |z| has no packed form (no |.| modifier on packed operands).
typedef float v2f __attribute__((ext_vector_type(2)));
KTUP_DEV v2f lo2(float4 a) { return v2f{a.x, a.y}; }
KTUP_DEV v2f hi2(float4 a) { return v2f{a.z, a.w}; }
KTUP_DEV v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// CF: the candidate's chunk c of staged vector v -- the LDS tile (StagedCand) or the row itself (the list kernel's RowCand)
struct StagedCand {
  static constexpr int UNROLL = 2;
  const float4* cand; int nch4, lane;
  KTUP_DEV float4 operator()(int v, uint32_t c) const { return cand[(v * nch4 + c) * CT + lane]; }
};
// MODE 2 beyond 212 columns: three staged vectors pass the LDS (3 d x 256 B), two do not up to d = 320 -- v and NV stay staged, C0
// (read once per chunk in the second pass, shared by the QB queries of a group) comes from the lane's own row in memory: a lane
// streams its row, so a 128-byte line serves eight of its chunks
struct HybridCand {
  static constexpr int UNROLL = 2;
  const float4* cand; const float4* row0; int nch4, lane;
  KTUP_DEV float4 operator()(int v, uint32_t c) const { return v == 0 ? row0[c] : cand[((v - 1) * nch4 + c) * CT + lane]; }
};
// TransH's s = -(e . w) of the lane's candidate against NQ wave-uniform normals.  ONE function for the pair kernels' first pass and for
// the kernel that tabulates s per (relation, candidate) once per pass: the same operations in the same order, so a table entry has
// the bits the two-pass kernels compute
template <int NQ, class CF>
KTUP_DEV void transh_dots(const CF& cf, int nch4, const sptr4 (&qn)[NQ], float (&s)[NQ]) {
  v2f s2[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) s2[qi] = v2f{0.f, 0.f};
#pragma unroll CF::UNROLL
  for (uint32_t c = 0; c < (uint32_t)nch4; ++c) {
    const float4 c1 = cf(0, c);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const float4 nqv = sldp(qn[qi] + c);
      s2[qi] = fma2(-lo2(c1), lo2(nqv), s2[qi]);
      s2[qi] = fma2(-hi2(c1), hi2(nqv), s2[qi]);
    }
  }
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) s[qi] = s2[qi].x + s2[qi].y;
}

// MODE 3 = MODE 1 with s read from the pass's table (stab: this lane's candidate against each query's relation)
template <int MODE, bool L1, int NQ, class CF>
KTUP_DEV void pair_group_scores(const CF& cf, int nch4, const sptr4 (&qa)[NQ], const sptr4 (&qn)[NQ], const sptr4 (&q1p)[NQ], float (&acc)[NQ],
                                const float* stab = nullptr) {
  constexpr bool TRANSH = MODE == 1 || MODE == 3;
  v2f s2[NQ], a2[NQ];
  float s[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) { s2[qi] = v2f{0.f, 0.f}; a2[qi] = v2f{0.f, 0.f}; acc[qi] = 0.f; s[qi] = 0.f; }
  if constexpr (MODE == 1) transh_dots<NQ>(cf, nch4, qn, s);
  if constexpr (MODE == 3) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) s[qi] = stab[qi];
  }
  if constexpr (MODE == 2) {
#pragma unroll CF::UNROLL
    for (uint32_t c = 0; c < (uint32_t)nch4; ++c) {
      const float4 c1 = cf(1, c);
      const float4 nc = cf(2, c);
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const float4 nqv = sldp(qn[qi] + c);
        const float4 q1 = sldp(q1p[qi] + c);
        s2[qi] = fma2(lo2(q1) - lo2(c1), lo2(nqv) + lo2(nc), s2[qi]);
        s2[qi] = fma2(hi2(q1) - hi2(c1), hi2(nqv) + hi2(nc), s2[qi]);
      }
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) s[qi] = s2[qi].x + s2[qi].y;
  }
  v2f ms[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) ms[qi] = v2f{-s[qi], -s[qi]};
#pragma unroll CF::UNROLL
  for (uint32_t c = 0; c < (uint32_t)nch4; ++c) {
    const float4 c0 = cf(0, c);
    const float4 nc = MODE == 2 ? cf(2, c) : f4zero();
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const float4 av = sldp(qa[qi] + c);
      v2f zl = lo2(av) - lo2(c0), zh = hi2(av) - hi2(c0);
      if constexpr (MODE == 2) {
        const float4 nqv = sldp(qn[qi] + c);
        zl = fma2(ms[qi], lo2(nqv) + lo2(nc), zl);
        zh = fma2(ms[qi], hi2(nqv) + hi2(nc), zh);
      }
      if constexpr (TRANSH) {
        const float4 nqv = sldp(qn[qi] + c);
        zl = fma2(ms[qi], lo2(nqv), zl);
        zh = fma2(ms[qi], hi2(nqv), zh);
      }
      if constexpr (L1) {
        acc[qi] += (fabsf(zl.x) + fabsf(zl.y)) + (fabsf(zh.x) + fabsf(zh.y));
      } else {
        a2[qi] = fma2(zl, zl, a2[qi]);
        a2[qi] = fma2(zh, zh, a2[qi]);
      }
    }
  }
  if constexpr (!L1) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) acc[qi] = a2[qi].x + a2[qi].y;
  }
}

// a wave-uniform address read through the scalar cache (data written by an earlier kernel)
template <class T>
KTUP_DEV T uload(const T* p) { return *(const __attribute__((address_space(4))) T*)(uintptr_t)p; }

KTUP_DEV uint64_t count_key(float s, uint32_t id) {    // ktup_rank.hip make_key (ascending)
  if (s == 0.f) s = 0.f;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | id;
}

template <int MODE, bool L1, bool COUNT = false, bool HYB = false>
__global__ __launch_bounds__(PairsWG<MODE>::NT) void pairs_kernel(PairsArgs a) {
  static_assert(!HYB || (MODE == 2 && !COUNT), "the hybrid stage is the soft gate's score kernel's");
  constexpr int NWV = PairsWG<MODE>::NWV, NT = PairsWG<MODE>::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* cand = reinterpret_cast<float4*>(smem);  // [NCV][nch4][CT]  (HYB: [2][nch4][CT], vectors 1 and 2)
  constexpr int NCV = MODE == 2 ? 3 : 1;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nch4 = a.dq / 4;
  const int64_t j0 = (int64_t)blockIdx.x * CT;
  int64_t range_lo = 0, range_hi = a.nq;
  const float* C0 = a.C0;
  if (a.qperm) {  // (uniform per workgroup)
    range_lo = a.rel_off[blockIdx.z];
    range_hi = a.rel_off[blockIdx.z + 1];
    if (range_lo >= range_hi) return;
    C0 += (int64_t)blockIdx.z * a.rel_stride;
  }
  for (int idx = t; idx < nch4 * CT; idx += NT) {
    const int j = idx & (CT - 1), c = idx >> 6;
    const int64_t gj = min(j0 + j, a.n_cand - 1);
    if (!HYB) cand[(0 * nch4 + c) * CT + j] = load_cand4(C0, a.ldc0, gj, c, a.d, a.cvec);
    if (NCV == 3) {
      cand[((HYB ? 0 : 1) * nch4 + c) * CT + j] = load_cand4(a.C1, a.ldc1, gj, c, a.d, a.cvec);
      cand[((HYB ? 1 : 2) * nch4 + c) * CT + j] = load_cand4(a.C2, a.ldc2, gj, c, a.d, a.cvec);
    }
  }
  __syncthreads();
  const sptr4 QW = as_scalar(a.QW);
  const int dq4 = nch4;
  // this workgroup's slice of the queries (grid.y splits them), QB at a time per wave
  const int64_t nrange = range_hi - range_lo;
  const int64_t per = ((nrange + gridDim.y - 1) / gridDim.y + NWV * QB - 1) / (NWV * QB) * (NWV * QB);
  const int64_t qlo = range_lo + (int64_t)blockIdx.y * per, qhi = min(range_hi, qlo + per);
  for (int64_t b0 = qlo + w * QB; b0 < qhi; b0 += NWV * QB) {
    // per-query scalar base pointers (A, Q1, NQ vectors), hoisted: inside the chunk loops every scalar load is then
    // base + one shared 32-bit offset -- with an index recomputed per load, 64-bit scalar address arithmetic (5 SALU per
    // s_load) costs as many issue slots as the VALU work it feeds
    sptr4 qa[QB], qn[QB], q1p[QB];
    int64_t qid[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
      const int64_t pos = min(b0 + qi, range_hi - 1);
      qid[qi] = a.qperm ? (int64_t)a.qperm[pos] : pos;
      qa[qi] = QW + qid[qi] * 3 * dq4;
      q1p[qi] = qa[qi] + dq4;
      qn[qi] = qa[qi] + 2 * dq4;
    }
    // COUNT: the queries' gold lists (offsets through scalar loads, then the first TH scores and ids of all QB queries with ONE vector
    // load each: lane qi * TH + k takes slot k of query qi) are requested BEFORE the scores are computed -- after them they were
    // three dependent round trips per query with nothing to hide behind
    constexpr int TH = 4;
    static_assert(QB == 4 && TH == 4, "lane qi * TH + k <-> (query of the group, gold slot)");
    int64_t cg0[QB];
    int cng[QB];
    float thv = 0.f;            // lane qi * TH + k: the k-th gold's score / id of the group's query qi -- two registers for the whole
    uint32_t gidv = 0;          // group, read back lane by lane (v_readlane) where a threshold is needed
    int64_t slot_g = 0;
    if constexpr (COUNT) {
#pragma unroll
      for (int qi = 0; qi < QB; ++qi) {
        cg0[qi] = uload(a.gold_off + qid[qi]);
        cng[qi] = b0 + qi < qhi ? (int)(uload(a.gold_off + qid[qi] + 1) - cg0[qi]) : 0;
      }
      const int sq = (lane >> 2) & 3, sk = lane & 3;
      slot_g = (sq == 0 ? cg0[0] : sq == 1 ? cg0[1] : sq == 2 ? cg0[2] : cg0[3]) + sk;
      if (lane < QB * TH && sk < (sq == 0 ? cng[0] : sq == 1 ? cng[1] : sq == 2 ? cng[2] : cng[3])) {
        thv = a.gscore[slot_g];
        gidv = (uint32_t)a.gold_ids[slot_g];
      }
    }
    float stab[QB];
    if constexpr (MODE == 3) {            // s of (this lane's candidate, each query's relation): one coalesced table read per query
#pragma unroll
      for (int qi = 0; qi < QB; ++qi) stab[qi] = a.wtab[uload(a.rel + qid[qi]) * a.ldw + j0 + lane];
    }
    float acc[QB];
    if constexpr (HYB)
      pair_group_scores<MODE, L1, QB>(HybridCand{cand, reinterpret_cast<const float4*>(C0 + min(j0 + lane, a.n_cand - 1) * a.ldc0), nch4, lane},
                                      nch4, qa, qn, q1p, acc, stab);
    else
      pair_group_scores<MODE, L1, QB>(StagedCand{cand, nch4, lane}, nch4, qa, qn, q1p, acc, stab);
    if constexpr (COUNT) {
      // per gold entry of each query: the candidates of this tile ordered before it -- (score, id) order of ktup_rank.hip's keys: a
      // lower score, or the same score and a lower id; NaNs on either side (uniform tests) go through the keys themselves
      // Branch-free per gold: three compares into lane masks, mask arithmetic and a population count on the scalar unit; the counts
      // of the group's first TH golds per query collect in one register (lane qi * TH + k) and leave as ONE atomic instruction
      // (written with short-circuit tests and an atomic per gold, the epilogue was ~45 instructions and 8 branches per gold:
      // a third of the kernel at d = 100)
      const uint64_t inm = __builtin_amdgcn_ballot_w64(j0 + lane < a.n_cand);
      const uint32_t cid = (uint32_t)(j0 + lane);
      const bool desc = a.descending != 0;
      int cntv = 0;
#pragma unroll
      for (int qi = 0; qi < QB; ++qi) {
        const float sc = desc ? -acc[qi] : acc[qi];
        // a threshold read back from its lane (v_readlane) is a scalar: its sign flip and NaN test run on the scalar unit, and a
        // query with a NaN on either side (rare) takes the key compares for all its golds -- nothing of that path is computed otherwise
        bool slow = __builtin_amdgcn_ballot_w64(sc != sc) != 0;
        uint32_t thb[TH], gid[TH];
#pragma unroll
        for (int k = 0; k < TH; ++k) {
          thb[k] = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(thv), qi * TH + k) ^ (desc ? 0x80000000u : 0u);
          gid[k] = (uint32_t)__builtin_amdgcn_readlane((int)gidv, qi * TH + k);
          slow |= k < cng[qi] && (thb[k] & 0x7fffffffu) > 0x7f800000u;
        }
        if (!slow) {
#pragma unroll
          for (int k = 0; k < TH; ++k)
            if (k < cng[qi]) {
              const float th = __uint_as_float(thb[k]);
              const uint64_t lt = __builtin_amdgcn_ballot_w64(sc < th), eq = __builtin_amdgcn_ballot_w64(sc == th);
              const int n = (int)__popcll((lt | (eq & __builtin_amdgcn_ballot_w64(cid < gid[k]))) & inm);
              asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(n), "i"(qi * TH + k));   // (no builtin for it in this compiler)
            }
        } else {
          const uint64_t key = count_key(sc, cid);
#pragma unroll
          for (int k = 0; k < TH; ++k)
            if (k < cng[qi]) {
              const int n = (int)__popcll(__builtin_amdgcn_ballot_w64(key < count_key(__uint_as_float(thb[k]), gid[k])) & inm);
              asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(n), "i"(qi * TH + k));
            }
        }
        for (int k = TH; k < cng[qi]; ++k) {    // beyond TH golds: one at a time
          const float th0 = uload(a.gscore + cg0[qi] + k);
          const float th = desc ? -th0 : th0;
          const uint32_t gid = (uint32_t)uload(a.gold_ids + cg0[qi] + k);
          uint64_t m;
          if (slow || th != th) {
            m = __builtin_amdgcn_ballot_w64(count_key(sc, cid) < count_key(th, gid));
          } else {
            const uint64_t lt = __builtin_amdgcn_ballot_w64(sc < th), eq = __builtin_amdgcn_ballot_w64(sc == th);
            m = lt | (eq & __builtin_amdgcn_ballot_w64(cid < gid));
          }
          const int n = (int)__popcll(m & inm);
          if (n != 0 && lane == 0) atomicAdd(a.counts + cg0[qi] + k, n);
        }
      }
      if (lane < QB * TH && cntv != 0) atomicAdd(a.counts + slot_g, cntv);
      continue;
    }
    if (j0 + lane < a.n_cand) {
#pragma unroll
      for (int qi = 0; qi < QB; ++qi)
        if (b0 + qi < qhi) a.out[qid[qi] * a.ldo + j0 + lane] = acc[qi];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Hard (ST-Gumbel) gate, K15/K16 with use_st_gumbel: w(b,j) = onehot(argmax_p LU_b[p] + LV_j[p] + g(b,j,p)),
// r = beta A[p*], n = beta C[p*]; z = q + r - (q.n) n with q = u_b - v_j.  Noise is per (user, item, preference)
// exactly like the reference, which draws a (B x N x P) uniform tensor inside evaluate (transUP.py:92).
struct HardArgs {
  const float *V, *LV;  // candidate vectors v_j [N][d], logits [N][P]
  const float *QW, *QL; // query vectors (slot 1 = u_b), logits [nq][P]
  const float *QN, *VN; // u_b . Cn_p [nq][P], v_j . Cn_p [N][P]   (squared L2)
  const float* consts;  // [4][32]: |Ar_p|^2, Ar_p . Cn_p, |Cn_p|^2 - 2, {4 beta}   (squared L2)
  const float* ws;      // prepared tables
  int ppad, dp, P, d;
  int64_t n_cand, nq;
  int l1, gumbel;
  const float* uniform;
  uint64_t seed, offset;
  float* out;
  int64_t ldo;
};

constexpr int HARD_NT = 512;   // 8 waves share a candidate tile + tables (47 KB at d = 100, P = 20): 3 workgroups per CU

// The stage both hard-gate kernels score against: 64 candidates' vectors [nch4][CT] and logits [P][CT], then
//   MODE 1 (L1) and MODE 2 (squared L2 with more than 32 preferences): the tables Ar, Cn [P][dp4] (the chosen preference's rows are
//                read per lane), two passes over d;
//   MODE 0 (squared L2):  v_j . Cn_p [P][CT] and the per-preference constants [4][32] -- no table rows at all:
//       |q + r - (q.n) n|^2 = |q|^2 + 2 q.r + |r|^2 - 2 s (s + r.n) + s^2 |n|^2,   q = u - v, s = q.n = u.n - v.n,
//     and q.r = 4 beta (LU - LV)/2... precisely r = Ar_p = 2 beta Alog_p (a power of two: exact), so u.r - v.r = 2 beta (LU_p - LV_p)
//     comes from the logits the gate has already read.  One table-free pass over d (25 LDS reads and 8 VALU instructions per
//     16-byte chunk instead of 125 reads -- 75 of them rows picked per lane, colliding in the banks -- and 24 instructions).
struct HardStage {
  float4* cand; float* lv; float4 *tabA, *tabC; float* vn; float* cst;
};
constexpr int HARD_PMAX = 64;   // hard_user_row: lane p holds the user's logit p (beyond it pref_scores_tail scores wave per pair)
template <int MODE>
__host__ __device__ inline size_t hard_stage_floats(int nch4, int P, int dp4) {
  return (size_t)nch4 * CT * 4 + (size_t)P * CT + (MODE != 0 ? (size_t)2 * P * dp4 * 4 : (size_t)P * CT + 128);
}
template <int MODE>
KTUP_DEV HardStage hard_stage_carve(char* smem, int nch4, int P, int dp4) {
  HardStage h{};
  h.cand = reinterpret_cast<float4*>(smem);
  h.lv = reinterpret_cast<float*>(h.cand + nch4 * CT);
  if (MODE != 0) { h.tabA = reinterpret_cast<float4*>(h.lv + P * CT); h.tabC = h.tabA + P * dp4; }
  else { h.vn = h.lv + P * CT; h.cst = h.vn + P * CT; }
  return h;
}
// what does not change with the candidate tile
template <int MODE>
KTUP_DEV void hard_stage_tables(const HardArgs& a, const HardStage& h, int t, int nt) {
  const int dp4 = a.dp / 4;
  if (MODE != 0) {
    const float4* Ar = reinterpret_cast<const float4*>(a.ws + (size_t)a.ppad * a.dp);
    for (int idx = t; idx < 2 * a.P * dp4; idx += nt) h.tabA[idx] = Ar[idx];  // Ar then Cn are adjacent in ws
  } else {
    for (int idx = t; idx < 128; idx += nt) h.cst[idx] = a.consts[idx];
  }
}
template <int MODE>
KTUP_DEV void hard_stage_tile(const HardArgs& a, const HardStage& h, int64_t j0, int t, int nt) {
  const int nch4 = a.d / 4;
  for (int idx = t; idx < nch4 * CT; idx += nt) {
    const int j = idx & (CT - 1), c = idx >> 6;
    h.cand[c * CT + j] = reinterpret_cast<const float4*>(a.V + min(j0 + j, a.n_cand - 1) * a.d)[c];
  }
  for (int idx = t; idx < a.P * CT; idx += nt) {
    const int j = idx & (CT - 1), p = idx >> 6;
    const int64_t row = min(j0 + j, a.n_cand - 1);
    h.lv[p * CT + j] = a.LV[row * a.P + p];
    if (MODE == 0) h.vn[p * CT + j] = a.VN[row * a.P + p];
  }
}
// the score of (user b, this lane's candidate) under the gate's choice ps
// the user's logits (and normal products): a wave-uniform row of P floats, lane p holds entry p -- one vector load per user; the gate
// reads entry p through v_readlane (a load inside its loop is a round trip per preference: the waves then wait on memory 57 % of
// the time), the score picks the chosen entry across lanes
struct UserRow { float ql, qn; };
template <int MODE>
KTUP_DEV UserRow hard_user_row(const HardArgs& a, int64_t b, int lane) {
  const int pl = lane < a.P ? lane : a.P - 1;
  UserRow u;
  u.ql = a.QL[b * a.P + pl];
  u.qn = MODE != 0 ? 0.f : a.QN[b * a.P + pl];
  return u;
}
KTUP_DEV float lane_entry(float v, int p) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), p)); }

template <int MODE>
KTUP_DEV float hard_pair_score(const HardArgs& a, const HardStage& h, sptr4 QW, int64_t b, const UserRow& ur, int ps, int lane) {
  const int nch4 = a.d / 4, dp4 = a.dp / 4;
  const sptr4 ub = QW + (b * 3 + 1) * nch4;   // slot 1 = u_b; one scalar base, the chunk index is the only offset
  if (MODE != 0) {
    const float4* cn = h.tabC + ps * dp4;
    const float4* ar = h.tabA + ps * dp4;
    v2f s2 = v2f{0.f, 0.f};                         // float pairs, as in pair_group_scores
    for (uint32_t c = 0; c < (uint32_t)nch4; ++c) {
      const float4 u = sldp(ub + c), e = h.cand[c * CT + lane], n = cn[c];
      s2 = fma2(lo2(u) - lo2(e), lo2(n), s2);
      s2 = fma2(hi2(u) - hi2(e), hi2(n), s2);
    }
    const float s = s2.x + s2.y;
    const v2f ms = v2f{-s, -s};
    v2f a2 = v2f{0.f, 0.f};
    float acc = 0.f;
    for (uint32_t c = 0; c < (uint32_t)nch4; ++c) {
      const float4 u = sldp(ub + c), e = h.cand[c * CT + lane], n = cn[c], r = ar[c];
      const v2f zl = fma2(ms, lo2(n), (lo2(u) - lo2(e)) + lo2(r)), zh = fma2(ms, hi2(n), (hi2(u) - hi2(e)) + hi2(r));
      if constexpr (MODE == 1) {
        acc += (fabsf(zl.x) + fabsf(zl.y)) + (fabsf(zh.x) + fabsf(zh.y));
      } else {
        a2 = fma2(zl, zl, a2);
        a2 = fma2(zh, zh, a2);
      }
    }
    return MODE == 1 ? acc : a2.x + a2.y;
  }
  const float qls = __shfl(ur.ql, ps, 64), qns = __shfl(ur.qn, ps, 64);
  v2f q2 = v2f{0.f, 0.f};
  for (uint32_t c = 0; c < (uint32_t)nch4; ++c) {
    const float4 u = sldp(ub + c), e = h.cand[c * CT + lane];
    const v2f ql = lo2(u) - lo2(e), qh = hi2(u) - hi2(e);
    q2 = fma2(ql, ql, q2);
    q2 = fma2(qh, qh, q2);
  }
  const float qq = q2.x + q2.y;
  const float s = qns - h.vn[ps * CT + lane];
  const float lin = fmaf(h.cst[96], qls - h.lv[ps * CT + lane], h.cst[ps]);          // 2 q.r + |r|^2
  return fmaf(s, fmaf(s, h.cst[64 + ps], -2.f * h.cst[32 + ps]), qq + lin);           // + s (s (|n|^2 - 2) - 2 r.n)
}

template <int MODE>
__global__ __launch_bounds__(HARD_NT) void pairs_hard_kernel(HardArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const HardStage h = hard_stage_carve<MODE>(smem, a.d / 4, a.P, a.dp / 4);
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t j0 = (int64_t)blockIdx.x * CT;
  const int64_t gj = min(j0 + lane, a.n_cand - 1);
  hard_stage_tables<MODE>(a, h, t, HARD_NT);
  hard_stage_tile<MODE>(a, h, j0, t, HARD_NT);
  __syncthreads();
  const sptr4 QW = as_scalar(a.QW);
  const int64_t per = (a.nq + gridDim.y - 1) / gridDim.y;
  const int64_t qlo = (int64_t)blockIdx.y * per, qhi = min(a.nq, qlo + per);
  for (int64_t b = qlo + w; b < qhi; b += HARD_NT / 64) {
    const UserRow ur = hard_user_row<MODE>(a, b, lane);
    const uint64_t base = ((uint64_t)b * (uint64_t)a.n_cand + (uint64_t)gj) * (uint64_t)a.P;
    const int ps = gate_argmax(a.P, base, a.gumbel == KTUP_GUMBEL_INPUT, a.uniform, a.seed, a.offset,
                               [&](int p) { return lane_entry(ur.ql, p) + h.lv[p * CT + lane]; });
    const float acc = hard_pair_score<MODE>(a, h, QW, b, ur, ps, lane);
    if (j0 + lane < a.n_cand) a.out[b * a.ldo + j0 + lane] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The hard gate's evaluation PASS in one sweep (transUP.py:84-102 / jTransUP.py:163-191 with use_st_gumbel + utils/misc.py:186-248):
// pairs_hard_kernel's pair arithmetic, instruction for instruction (the scores are the batched route's bits, the noise its stream
// positions), with the filtered top-n taken where the scores are made.  A wave owns 16 users and walks a split of the catalogue in
// 64-item stages (lane <-> item; the user's vectors are wave-uniform scalar loads, as above); a user's sorted list (64-bit keys =
// score image << 32 | item id) lives in the wave's LDS, a score is a candidate if it is below the user's n-th score (floats; the keys
// decide ties), is not filtered (bitmap of the split, built once per wave from the CSR lists) and candidates are inserted one by
// one by ballot position -- a few per user and stage once the list has warmed up.  The (users x items) matrix never exists: the
// batched route wrote and re-read 78 MB per ml1m pass and ran K17 twelve times; what is left is the noise (20 Philox draws and
// Gumbel transforms per pair) and the d-long distance.  The splits' partial lists are merged by ktup_eval_pass.hip's merge launch.
constexpr uint64_t SKEY_MAX = ~0ull;
constexpr int SW_NW = 8, SW_UW = 8;       // waves per workgroup, users per wave (LDS allows two workgroups per CU: 16 waves)

KTUP_DEV uint64_t sweep_key(float s, uint32_t id) {    // ktup_rank.hip make_key, ascending
  if (s == 0.f) s = 0.f;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | id;
}

struct SweepHardArgs {
  HardArgs h;
  const int64_t* filt_off; const int32_t* filt_ids;
  int topn, nsplit; int64_t split_items; uint64_t* part; int bm_words;
};

// ---- what every sweep keeps per wave in LDS for its UW users: sorted lists [UW][topn] u64 | n-th keys [UW] u64 | n-th scores [UW]
// (NaN while a list is short) | filter bits of the workgroup's catalogue split [UW][bm_words]
__host__ __device__ inline size_t sweep_wave_bytes(int uw, int bm_words, int topn) {
  return (((size_t)uw * topn * 8 + (size_t)uw * 8 + (size_t)uw * 4 + (size_t)uw * bm_words * 4) + 7) & ~(size_t)7;
}
struct SweepState { uint64_t *tk, *thrk; float* thrf; uint32_t* bm; int topn, bm_words; };

// lists empty, thresholds open (rows past the end: closed), the filter bitmap of items [i_lo, i_hi) from the CSR lists -- the wave's UW
// users are consecutive, so their lists are one run of ids that all 64 lanes walk together (sixteen loads in flight per lane)
template <int UW>
KTUP_DEV SweepState sweep_state_init(char* wb, int topn, int bm_words, int64_t u0, int64_t nq, const int64_t* __restrict__ filt_off,
                                     const int32_t* __restrict__ filt_ids, int64_t i_lo, int64_t i_hi, int lane) {
  SweepState st;
  st.tk = reinterpret_cast<uint64_t*>(wb); st.thrk = st.tk + UW * topn; st.thrf = reinterpret_cast<float*>(st.thrk + UW);
  st.bm = reinterpret_cast<uint32_t*>(st.thrf + UW); st.topn = topn; st.bm_words = bm_words;
  for (int idx = lane; idx < UW * topn; idx += 64) st.tk[idx] = SKEY_MAX;
  if (lane < UW) { st.thrk[lane] = u0 + lane < nq ? SKEY_MAX : 0; st.thrf[lane] = u0 + lane < nq ? __uint_as_float(0x7fffffffu) : -__builtin_inff(); }
  for (int idx = lane; idx < UW * bm_words; idx += 64) st.bm[idx] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (filt_off) {
    const int64_t uo = u0 + (lane < UW ? lane : UW);
    const int64_t myoff = filt_off[uo < nq ? uo : nq];
    const int64_t f_begin = __shfl(myoff, 0, 64), f_end = __shfl(myoff, UW, 64);
    uint32_t rel[UW > 1 ? UW - 1 : 1];
#pragma unroll
    for (int k = 0; k < UW - 1; ++k) rel[k] = (uint32_t)(__shfl(myoff, k + 1, 64) - f_begin);
    const int64_t span = i_hi - i_lo;
    constexpr int FB = 16;
    for (int64_t base = f_begin; base < f_end; base += 64 * FB) {
      int32_t ids[FB];
#pragma unroll
      for (int k = 0; k < FB; ++k) {
        const int64_t f = base + lane + 64 * k;
        ids[k] = f < f_end ? filt_ids[f] : -1;
      }
#pragma unroll
      for (int k = 0; k < FB; ++k) {
        const int64_t id = (int64_t)ids[k] - i_lo;
        const uint32_t pos = (uint32_t)(base - f_begin) + lane + 64 * k;
        int r = 0;
#pragma unroll
        for (int q = 0; q < UW - 1; ++q) r += pos >= rel[q] ? 1 : 0;
        if (ids[k] >= 0 && id >= 0 && id < span) atomicOr(st.bm + r * bm_words + (id >> 5), 1u << (id & 31));
      }
    }
  }
  return st;
}

// the lane's score of (user r of the wave, item): a candidate if it is below the user's n-th score (floats; the keys decide equality and
// NaNs) and its filter bit is clear; the wave inserts its candidates into the user's sorted list one by one at ballot positions
KTUP_DEV void sweep_rank(const SweepState& st, int r, float acc, int64_t item, bool iok, int64_t lid, int lane) {
  const int topn = st.topn;
  const float tf = st.thrf[r];
  bool c = acc < tf;
  const bool tie = !c && !(acc > tf);
  if (__builtin_amdgcn_ballot_w64(tie)) {
    if (tie) c = sweep_key(acc, (uint32_t)item) < st.thrk[r];
  }
  c = c && iok;
  if (c) c = ((st.bm[r * st.bm_words + (lid >> 5)] >> (lid & 31)) & 1u) == 0u;
  uint64_t m = __builtin_amdgcn_ballot_w64(c);
  if (m) {
    const uint64_t mine = sweep_key(acc, (uint32_t)item);
    uint64_t list = lane < topn ? st.tk[r * topn + lane] : SKEY_MAX;     // lanes 0..topn-1: the sorted list
    while (m) {
      const int src = __builtin_ctzll(m);
      m &= m - 1;
      const uint64_t k = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), src) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, src);
      const int pos = __popcll(__builtin_amdgcn_ballot_w64(list < k));  // entries below the newcomer (lanes >= topn hold MAX)
      if (pos < topn) {
        const uint64_t up = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(list >> 32), 1, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)list, 1, 64);
        list = lane < pos ? list : (lane == pos ? k : up);
        if (lane >= topn) list = SKEY_MAX;
      }
    }
    if (lane < topn) st.tk[r * topn + lane] = list;
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(list >> 32), topn - 1);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)list, topn - 1);
    if (lane == 0) {
      st.thrk[r] = ((uint64_t)hi << 32) | lo;
      st.thrf[r] = __uint_as_float((hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi);   // inverse image; NaN while the list is short
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int UW>
KTUP_DEV void sweep_store(const SweepState& st, uint64_t* __restrict__ part, int nsplit, int split, int64_t u0, int64_t nq, int lane) {
  if (lane < st.topn)
    for (int r = 0; r < UW; ++r)
      if (u0 + r < nq) part[((u0 + r) * nsplit + split) * st.topn + lane] = st.tk[r * st.topn + lane];
}

template <int MODE>
__global__ __launch_bounds__(SW_NW * 64) void sweep_hard_kernel(SweepHardArgs sa) {
  HardArgs a = sa.h;
  KTUP_RESOLVE_GUMBEL(a);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const HardStage h = hard_stage_carve<MODE>(smem, a.d / 4, a.P, a.dp / 4);
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t u0 = (int64_t)blockIdx.x * (SW_NW * SW_UW) + SW_UW * w;
  const int64_t i_lo = (int64_t)blockIdx.y * sa.split_items;
  const int64_t i_hi = min(a.n_cand, i_lo + sa.split_items);
  char* wb = smem + hard_stage_floats<MODE>(a.d / 4, a.P, a.dp / 4) * 4 + (size_t)w * sweep_wave_bytes(SW_UW, sa.bm_words, sa.topn);
  const SweepState st = sweep_state_init<SW_UW>(wb, sa.topn, sa.bm_words, u0, a.nq, sa.filt_off, sa.filt_ids, i_lo, i_hi, lane);
  hard_stage_tables<MODE>(a, h, t, SW_NW * 64);
  const sptr4 QW = as_scalar(a.QW);
  for (int64_t j0 = i_lo; j0 < i_hi; j0 += CT) {
    __syncthreads();                                                    // the previous stage has been consumed
    hard_stage_tile<MODE>(a, h, j0, t, SW_NW * 64);
    __syncthreads();
    const int64_t item = j0 + lane;
    const int64_t gj = min(item, a.n_cand - 1);
    const bool iok = item < i_hi;
    const int64_t lid = item - i_lo;
    for (int r = 0; r < SW_UW; ++r) {
      const int64_t b = u0 + r;
      if (b >= a.nq) break;
      const UserRow ur = hard_user_row<MODE>(a, b, lane);
      const uint64_t base = ((uint64_t)b * (uint64_t)a.n_cand + (uint64_t)gj) * (uint64_t)a.P;
      const int ps = gate_argmax(a.P, base, a.gumbel == KTUP_GUMBEL_INPUT, a.uniform, a.seed, a.offset,
                                 [&](int p) { return lane_entry(ur.ql, p) + h.lv[p * CT + lane]; });
      sweep_rank(st, r, hard_pair_score<MODE>(a, h, QW, b, ur, ps, lane), item, iok, lid, lane);
    }
  }
  sweep_store<SW_UW>(st, sa.part, sa.nsplit, blockIdx.y, u0, a.nq, lane);
}

// ---- the SOFT gate's pass with the pair arithmetic of pairs_kernel<2> (pair_group_scores: the scores are that route's bits), for L1
// and for widths the preference-space pass (ktup_eval_pass.hip) does not cover.  16 waves x 3 users share a 64-item stage of the three
// item arrays (v - RV, v, NV: 77 KB at d = 100, one workgroup per CU); the next stage's rows are in flight (registers) under the
// scores of this one -- the per-batch kernel restages 77 KB per 52 users and spends more time staging than scoring.
constexpr int SS_NW = 16, SS_UW = 3;
struct SweepSoftArgs {
  PairsArgs p;
  const int64_t* filt_off; const int32_t* filt_ids;
  int topn, nsplit; int64_t split_items; uint64_t* part; int bm_words;
};
// HYB: two of the three item arrays staged (v, NV) and v - RV streamed from the lane's row, as in pairs_kernel's hybrid stage: widths
// beyond 168 up to 256 (the stage is then 128 KB at d = 256, a thread's share of it still PF float4).
template <bool L1, bool HYB = false>
__global__ __launch_bounds__(SS_NW * 64) void sweep_soft_kernel(SweepSoftArgs sa) {
  const PairsArgs& a = sa.p;
  constexpr int NA = HYB ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* cand = reinterpret_cast<float4*>(smem);                       // [NA][nch4][CT]
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nch4 = a.dq / 4;
  const int64_t u0 = (int64_t)blockIdx.x * (SS_NW * SS_UW) + SS_UW * w;
  const int64_t i_lo = (int64_t)blockIdx.y * sa.split_items;
  const int64_t i_hi = min(a.n_cand, i_lo + sa.split_items);
  char* wb = smem + (size_t)NA * nch4 * CT * 16 + (size_t)w * sweep_wave_bytes(SS_UW, sa.bm_words, sa.topn);
  const SweepState st = sweep_state_init<SS_UW>(wb, sa.topn, sa.bm_words, u0, a.nq, sa.filt_off, sa.filt_ids, i_lo, i_hi, lane);
  const sptr4 QW = as_scalar(a.QW);
  sptr4 qa[SS_UW], qn[SS_UW], q1p[SS_UW];
#pragma unroll
  for (int qi = 0; qi < SS_UW; ++qi) {
    const int64_t b = min(u0 + qi, a.nq - 1);
    qa[qi] = QW + b * 3 * nch4; q1p[qi] = qa[qi] + nch4; qn[qi] = qa[qi] + 2 * nch4;
  }
  // a thread's share of a stage: 3 arrays x nch4 x 64 float4 over 1024 threads (at most PF each: d <= 168, the host side checks),
  // fetched one stage ahead into registers
  const int total = NA * nch4 * CT;
  constexpr int PF = 8;
  float4 pre[PF];
  const float* src[PF];
  int dst[PF];
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    const int idx = t + k * SS_NW * 64;
    const int slot = idx / (nch4 * CT), rem = idx - slot * nch4 * CT, c = rem >> 6;
    const int arr = slot + (HYB ? 1 : 0);                                // (HYB: the staged slots hold arrays 1 and 2)
    const bool on = idx < total;
    dst[k] = on ? (rem & (CT - 1)) : 0;                                  // the row inside the stage (slots past the end reload row 0: unused)
    src[k] = on ? (arr == 0 ? a.C0 : (arr == 1 ? a.C1 : a.C2)) + 4 * c : a.C0;
    pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#define KTUP_SS_FETCH(J0)                                                                                                   \
  _Pragma("unroll") for (int k = 0; k < PF; ++k)                                                                            \
    pre[k] = *reinterpret_cast<const float4*>(src[k] + min((J0) + dst[k], a.n_cand - 1) * a.d);
  KTUP_SS_FETCH(i_lo)
  for (int64_t j0 = i_lo; j0 < i_hi; j0 += CT) {
    __syncthreads();                                                    // the previous stage has been consumed
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (t + k * SS_NW * 64 < total) cand[t + k * SS_NW * 64] = pre[k];
    __syncthreads();
    if (j0 + CT < i_hi) { KTUP_SS_FETCH(j0 + CT) }                      // in flight under the scores below
    float acc[SS_UW];
    if constexpr (HYB)
      pair_group_scores<2, L1, SS_UW>(HybridCand{cand, reinterpret_cast<const float4*>(a.C0 + min(j0 + lane, a.n_cand - 1) * a.d), nch4, lane},
                                      nch4, qa, qn, q1p, acc);
    else
      pair_group_scores<2, L1, SS_UW>(StagedCand{cand, nch4, lane}, nch4, qa, qn, q1p, acc);
    const int64_t item = j0 + lane;
    const bool iok = item < i_hi;
    const int64_t lid = item - i_lo;
#pragma unroll
    for (int r = 0; r < SS_UW; ++r)
      if (u0 + r < a.nq) sweep_rank(st, r, acc[r], item, iok, lid, lane);
  }
#undef KTUP_SS_FETCH
  sweep_store<SS_UW>(st, sa.part, sa.nsplit, blockIdx.y, u0, a.nq, lane);
}

// ---------------------------------------------------------------------------------------------------------
// K11: out[b][j] = U[u_b] . I[j]  with v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate: bit-for-bit an fmaf chain).
// One wave per 32 x 32 output tile; A operand lane l = U[u_(m0 + l&31)][k + (l>>5)], B operand = I[n0 + l&31][k + (l>>5)].
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void bprmf_eval_kernel(const float* __restrict__ U, int64_t ldu, const float* __restrict__ I,
                                                         int64_t ldi, int d, const int64_t* __restrict__ u_ids, int64_t nq,
                                                         int64_t n_items, float* __restrict__ out, int64_t ldo, int vec) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.y * 64 + (wv & 1) * 32;
  const int64_t n0 = (int64_t)blockIdx.x * 64 + (wv >> 1) * 32;
  if (m0 >= nq || n0 >= n_items) return;
  const int hi = lane >> 5;
  const float* arow = U + u_ids[min(m0 + (lane & 31), nq - 1)] * ldu;
  const float* brow = I + min(n0 + (lane & 31), n_items - 1) * ldi;
  v16f acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (vec) {
    const int nch = d / 4;
    for (int c = 0; c < nch; ++c) {
      const float4 a4 = reinterpret_cast<const float4*>(arow)[c];
      const float4 b4 = reinterpret_cast<const float4*>(brow)[c];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? a4.y : a4.x, hi ? b4.y : b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? a4.w : a4.z, hi ? b4.w : b4.z, acc, 0, 0, 0);
    }
  } else {
    for (int k = 0; k < d; k += 2) {
      const int kk = k + hi;
      const float av = kk < d ? arow[kk] : 0.f, bv = kk < d ? brow[kk] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  const int64_t col = n0 + (lane & 31);
  if (col < n_items) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row < nq) out[row * ldo + col] = acc[r];
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// K14 (TransR): every entity projected by every relation's matrix, PE[rho][j] = M_rho e_j (misc.py:29-33); the
// reference recomputes (B x d x d).(d x E) per query although only n_rel distinct products exist.
// Workgroup = (64-entity tile, relation); lane <-> entity, waves split the output coordinates; M rows are
// wave-uniform -> scalar loads.
__global__ __launch_bounds__(256) void transr_project_kernel(const float* __restrict__ E, int64_t lde, const float* __restrict__ M,
                                                             int64_t ldm, int d, int dq, int64_t n_ent, int evec, int mvec,
                                                             float* __restrict__ PE) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* cand = reinterpret_cast<float4*>(smem);  // [nch4][CT]
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nch4 = dq / 4;
  const int64_t j0 = (int64_t)blockIdx.x * CT;
  const int rho = blockIdx.y;
  for (int idx = t; idx < nch4 * CT; idx += 256) {
    const int j = idx & (CT - 1), c = idx >> 6;
    cand[c * CT + j] = load_cand4(E, lde, min(j0 + j, n_ent - 1), c, d, evec);
  }
  __syncthreads();
  const float* Mr = M + (int64_t)rho * ldm;
  float* outrow = PE + ((int64_t)rho * n_ent + min(j0 + lane, n_ent - 1)) * dq;
  const bool live = j0 + lane < n_ent;
  for (int i = w; i < dq; i += 4) {
    float acc = 0.f;
    if (i < d) {
      if (mvec) {
        const sptr4 mrow = as_scalar(Mr + (int64_t)i * d);
        for (int c = 0; c < nch4; ++c) acc += dot4(sld(mrow, c), cand[c * CT + lane]);
      } else {
        const float* mrow = Mr + (int64_t)i * d;
        for (int k = 0; k < d; ++k) {
          const float4 e4 = cand[(k >> 2) * CT + lane];
          const float ev = (k & 3) == 0 ? e4.x : (k & 3) == 1 ? e4.y : (k & 3) == 2 ? e4.z : e4.w;
          acc = fmaf(mrow[k], ev, acc);
        }
      }
    }
    if (live) outrow[i] = acc;
  }
}

// TransR, squared L2, on the matrix cores:  |c - M_r e|^2 = |c|^2 - 2 (M_r^T c) . e + |M_r e|^2.
//   fold : per query, c' = M_r^T c replaces c in slot 0 (so the pair term is ONE (queries x d) . (d x entities) GEMM against the
//          UNPROJECTED entity table), |c|^2 and the relation id go to side arrays.  One wave per query.
//   norms: |M_r e|^2 for every (relation, entity) = the K4 matrix-core forward with h = e, no tail, no translation.
__global__ __launch_bounds__(256) void transr_query_fold_kernel(float* __restrict__ QW, int dq, const float* __restrict__ M,
                                                                int64_t ldm, int d, const int64_t* __restrict__ r, int64_t nq,
                                                                float* __restrict__ qcc, int32_t* __restrict__ qrel) {
  __shared__ float cs[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= nq) return;
  float* c = QW + b * 3 * dq;
  const int rel = (int)r[b];
  float part = 0.f;
  for (int k = lane; k < d; k += 64) { const float v = c[k]; cs[w][k] = v; part = fmaf(v, v, part); }
  part = wave_sum(part);
  const float* Mr = M + (int64_t)rel * ldm;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};                          // c'[k] for k = lane + 64 m  (d <= 256)
  for (int i = 0; i < d; ++i) {
    const float ci = cs[w][i];                                  // LDS broadcast
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = lane + 64 * m;
      if (k < d) acc[m] = fmaf(Mr[(int64_t)i * d + k], ci, acc[m]);
    }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int k = lane + 64 * m;
    if (k < d) c[k] = acc[m];
  }
  if (lane == 0) { qcc[b] = part; qrel[b] = rel; }
}

__global__ __launch_bounds__(256) void transr_all_pairs_ids_kernel(int64_t n_ent, int64_t n, int64_t* __restrict__ h, int64_t* __restrict__ r) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    h[k] = k % n_ent;
    r[k] = k / n_ent;
  }
}

// Counting sort of the queries by relation (single workgroup; nq is an evaluation batch, n_rel is small).
__global__ __launch_bounds__(256) void rel_bucket_kernel(const int64_t* __restrict__ r, int64_t nq, int n_rel,
                                                         int32_t* __restrict__ rel_off, int32_t* __restrict__ qperm) {
  extern __shared__ int cnt[];  // [n_rel] counts, then running cursors
  for (int i = threadIdx.x; i < n_rel; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int64_t b = threadIdx.x; b < nq; b += 256) atomicAdd(&cnt[(int)r[b]], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < n_rel; ++i) { const int c = cnt[i]; rel_off[i] = run; cnt[i] = run; run += c; }
    rel_off[n_rel] = run;
  }
  __syncthreads();
  for (int64_t b = threadIdx.x; b < nq; b += 256) qperm[atomicAdd(&cnt[(int)r[b]], 1)] = (int32_t)b;
}

// ---------------------------------------------------------------------------------------------------------
// grid.y splits the queries: enough workgroups to fill the chip, every split a whole number of NWV x QB query groups and none of
// them empty.  `slots` = the workgroups the chip holds at once (256 CUs x 6 at one staged vector, x 2 at three).  A short call gets
// about one round of them (`target`); a long one (a whole link-prediction pass: 230 tiles x 1280 groups) gets up to 8 rounds of
// workgroups with >= 6 query groups each -- at 2070 workgroups for 1536 slots the second round ran a third full and the kernel
// held 55 % of the vector pipe's issue rate.
dim3 pairs_grid(int64_t n_cand, int64_t nq, int nwv, int64_t target, int64_t slots) {
  const int64_t tiles = (n_cand + CT - 1) / CT;
  const int64_t group = (int64_t)nwv * QB;
  const int64_t ymax = (nq + group - 1) / group;
  const int64_t fine = tiles * ymax / 6;
  if (fine > target) target = fine < 8 * slots ? fine : 8 * slots;
  int64_t ysplit = (target + tiles - 1) / tiles;
  if (ysplit > ymax) ysplit = ymax;
  if (ysplit < 1) ysplit = 1;
  const int64_t per = ((nq + ysplit - 1) / ysplit + group - 1) / group * group;   // what the kernel computes from gridDim.y
  ysplit = (nq + per - 1) / per;
  return dim3((unsigned)tiles, (unsigned)(ysplit < 1 ? 1 : ysplit));
}

template <int MODE>
int launch_pairs(const PairsArgs& a, hipStream_t st, const char* name, int nrel = 1) {
  const int ncv = MODE == 2 ? 3 : 1;
  size_t lds = (size_t)ncv * (a.dq / 4) * CT * 16;
  dim3 grid = pairs_grid(a.n_cand, a.nq, PairsWG<MODE>::NWV, MODE == 2 ? 512 : 2048, MODE == 2 ? 512 : 1536);
  if constexpr (MODE == 2) {
    if (lds > 160 * 1024 && lds / 3 * 2 <= 160 * 1024 && a.cvec && a.d == a.dq && !a.qperm) {      // two vectors staged, C0 from its rows
      lds = lds / 3 * 2;
      auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(PairsWG<MODE>::NT), lds, st, a);
      };
      if (a.l1) launch(pairs_kernel<2, true, false, true>); else launch(pairs_kernel<2, false, false, true>);
      return check_launch(name);
    }
  }
  if (lds > 160 * 1024) return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size %d needs %zu B of LDS", name, a.d, lds);
  if (a.qperm) grid.z = (unsigned)nrel;
  if (a.l1) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)pairs_kernel<MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((pairs_kernel<MODE, true>), grid, dim3(PairsWG<MODE>::NT), lds, st, a);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)pairs_kernel<MODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((pairs_kernel<MODE, false>), grid, dim3(PairsWG<MODE>::NT), lds, st, a);
  }
  return check_launch(name);
}

// ---- the scores of every query's OWN list entries (golds, then filtered ids) with pair_group_scores: a wave takes one query at a
// time and scores up to 64 of its entries (lane <-> entry), every lane reading its entry's row where it lies (L2-resident: the sweep
// is streaming the same table) -- the thresholds and the subtrahend of the COUNT form's ranks (ktup_eval_kg_fused.hip), bit-identical
// to what the sweep computes for the same (query, candidate).  No LDS tile: staged through one, a CU held 6 waves and the kernel was
// a chain of exposed latencies (offsets -> ids -> rows -> LDS -> scores: 166 us per 20,480 keys of 22 entries).
constexpr int LIST_NW = 4;
template <bool VEC>
struct RowCand {              // UNROLL: the row loads of that many chunks are in flight together (one exposed latency per 5 chunks, not per chunk)
  static constexpr int UNROLL = 5;
  const float* row; int d;
  KTUP_DEV float4 operator()(int, uint32_t c) const {
    if constexpr (VEC) return *reinterpret_cast<const float4*>(row + 4 * c);
    return load_cand4(row, 0, 0, (int)c, d, false);
  }
};
template <int MODE, bool L1>
__global__ __launch_bounds__(LIST_NW * 64) void pairs_list_kernel(PairsArgs a) {
  static_assert(MODE != 2, "one staged vector per candidate");
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nch4 = a.dq / 4;
  const sptr4 QW = as_scalar(a.QW);
  for (int64_t key = (int64_t)blockIdx.x * LIST_NW + w; key < a.nq; key += (int64_t)gridDim.x * LIST_NW) {
    const int64_t g0 = a.gold_off[key], ng = a.gold_off[key + 1] - g0;
    const int64_t f0 = a.filt_off ? a.filt_off[key] : 0, nf = a.filt_off ? a.filt_off[key + 1] - f0 : 0;
    const sptr4 qa[1] = {QW + key * 3 * nch4}, q1p[1] = {qa[0] + nch4}, qn[1] = {qa[0] + 2 * nch4};
    for (int64_t base = 0; base < ng + nf; base += 64) {
      const int64_t e = base + lane;
      const bool on = e < ng + nf;
      const int32_t cid = !on ? 0 : (e < ng ? a.gold_ids[g0 + e] : a.filt_ids[f0 + (e - ng)]);
      const bool valid = on && cid >= 0 && cid < a.n_cand;
      const float* row = a.C0 + (int64_t)(valid ? cid : 0) * a.ldc0;
      float acc[1], stab[1] = {0.f};
      if constexpr (MODE == 3) stab[0] = a.wtab[a.rel[key] * a.ldw + (valid ? cid : 0)];
      if (a.cvec) pair_group_scores<MODE, L1, 1>(RowCand<true>{row, a.d}, nch4, qa, qn, q1p, acc, stab);
      else pair_group_scores<MODE, L1, 1>(RowCand<false>{row, a.d}, nch4, qa, qn, q1p, acc, stab);
      if (valid) {
        if (e < ng) a.gscore[g0 + e] = acc[0]; else a.fscore[f0 + (e - ng)] = acc[0];
      }
    }
  }
}

// ---- MODE 3's table: s = -(e . w_rel) for every (relation, candidate), once per pass (20 x 14,709 at ml1m-kg: a few microseconds)
// instead of once per (key, candidate) in the count kernel's first pass -- a fifth of TransH's pair arithmetic.  The pair kernels'
// stage (64 candidates in LDS, lane <-> candidate) with the relation normals in the queries' place, through transh_dots.
__global__ __launch_bounds__(256) void pairs_wtab_kernel(PairsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* cand = reinterpret_cast<float4*>(smem);
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nch4 = a.dq / 4;
  const int64_t j0 = (int64_t)blockIdx.x * CT;
  for (int idx = t; idx < nch4 * CT; idx += 256) {
    const int j = idx & (CT - 1), c = idx >> 6;
    cand[c * CT + j] = load_cand4(a.C0, a.ldc0, min(j0 + j, a.n_cand - 1), c, a.d, a.cvec);
  }
  __syncthreads();
  const sptr4 N4 = as_scalar(a.Nrm);
  const int64_t ldn4 = a.ldn / 4;
  for (int r0 = w * QB; r0 < a.n_rel; r0 += 4 * QB) {
    sptr4 qn[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) qn[qi] = N4 + (int64_t)min(r0 + qi, a.n_rel - 1) * ldn4;
    float s[QB];
    transh_dots<QB>(StagedCand{cand, nch4, lane}, nch4, qn, s);
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
      if (r0 + qi < a.n_rel) a.wtab[(int64_t)(r0 + qi) * a.ldw + j0 + lane] = s[qi];
  }
}

template <int MODE, bool L1>
int launch_pairs_count(const PairsArgs& a, hipStream_t st, const char* name) {
  const size_t tile = (size_t)(a.dq / 4) * CT * 16;
  if (tile > 160 * 1024) return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size %d needs %zu B of LDS", name, a.d, tile);
  if constexpr (MODE == 3) {                                    // (rows wider than 256 columns need more than the default 64 KB of LDS)
    if (tile > 64 * 1024) (void)hipFuncSetAttribute((const void*)pairs_wtab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile);
    hipLaunchKernelGGL(pairs_wtab_kernel, dim3((unsigned)((a.n_cand + CT - 1) / CT)), dim3(256), tile, st, a);
  }
  hipLaunchKernelGGL((pairs_list_kernel<MODE, L1>), dim3(grid_for((a.nq + LIST_NW - 1) / LIST_NW, 8192)), dim3(LIST_NW * 64), 0, st, a);
  const dim3 grid = pairs_grid(a.n_cand, a.nq, PairsWG<MODE>::NWV, 2048, 1536);
  if (tile > 64 * 1024) (void)hipFuncSetAttribute((const void*)pairs_kernel<MODE, L1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile);
  hipLaunchKernelGGL((pairs_kernel<MODE, L1, true>), grid, dim3(PairsWG<MODE>::NT), tile, st, a);
  return check_launch(name);
}

inline int round4(int d) { return (d + 3) / 4 * 4; }
inline size_t pad4(size_t x) { return (x + 3) & ~(size_t)3; }
// the evaluation workspaces of the preference models: items CW0 | CW1 | CW2 [N][d] | CL | CN [N][P] | consts [4][32]; users QW [nq][3][d] | QL | QN [nq][P]
inline size_t item_side_floats(int64_t n_items, int d, int P) { return (size_t)n_items * 3 * d + 2 * pad4((size_t)n_items * P) + 128; }
inline size_t user_side_floats(int64_t nq, int d, int P) { return (size_t)nq * 3 * d + 2 * pad4((size_t)nq * P); }

int kg_eval(int model, const char* name, const float* E, int64_t lde, const float* R, int64_t ldr, const float* X, int64_t ldx,
            int d, const float* C, int64_t ldc, int64_t n_cand, const int64_t* q, const int64_t* r, int64_t nq, int l1, int head,
            float* out, int64_t ldo, float* ws, void* stream) {
  KTUP_REQUIRE(d > 0 && nq >= 0 && n_cand >= 0, "%s: bad sizes", name);
  if (nq == 0 || n_cand == 0) return KTUP_OK;
  KTUP_REQUIRE(E && R && C && q && r && out && ws && (model == 0 || X), "%s: null pointer argument", name);
  KTUP_REQUIRE(ldo >= n_cand, "%s: output pitch %lld < n_cand", name, (long long)ldo);
  KTUP_REQUIRE(model != 2 || ldx >= (int64_t)d * d, "%s: projection pitch < d*d", name);
  hipStream_t st = (hipStream_t)stream;
  const int dq = round4(d);
  hipLaunchKernelGGL(kg_query_prep_kernel, dim3(grid_for((nq + 3) / 4)), dim3(256), 0, st, model, E, lde, R, ldr, X, ldx, d, dq,
                     q, r, nq, head, ws);
  if (int e = check_launch(name)) return e;
  if (!l1 && model <= 1) {   // squared L2: one (TransE) or two (TransH) GEMMs on the matrix cores; option eval_mc = 0 for A/B runs
    if (ktup::opt_eval_mc()) {
      const int rc = ktup::pairs_kg_l2_mc(model, ws, dq, C, ldc, d, nq, n_cand, out, ldo, st, name);
      if (rc != 1) return rc;
    }
  }
  PairsArgs a{};
  a.C0 = C; a.ldc0 = ldc; a.QW = ws; a.n_cand = n_cand; a.nq = nq; a.d = d; a.dq = dq; a.l1 = l1; a.out = out; a.ldo = ldo;
  a.cvec = (d % 4 == 0) && aligned16(C) && (ldc % 4 == 0);
  return model == 1 ? launch_pairs<1>(a, st, name) : launch_pairs<0>(a, st, name);
}

}  // namespace

// the query side alone (ktup_eval_kg_fused.hip): QW[nq][3][round4(d)], slot 0 = c, slot 2 = w (TransH)
int ktup::kg_query_prep(int model, const float* E, int64_t lde, const float* R, int64_t ldr, const float* X, int64_t ldx, int d,
                        const int64_t* q, const int64_t* r, int64_t nq, int head, float* QW, hipStream_t st, const char* name) {
  hipLaunchKernelGGL(kg_query_prep_kernel, dim3(grid_for((nq + 3) / 4)), dim3(256), 0, st, model, E, lde, R, ldr, X, ldx, d, round4(d), q, r,
                     nq, head, QW);
  return check_launch(name);
}

// the link-prediction pass's counts on the VALU route (any distance, any width, any number of golds per key): list scores, then the
// pair kernel in its COUNT form.  QW: kg_query_prep's rows; counts must be zero on entry.
int ktup::kg_valu_counts(int model, const float* QW, int d, const float* C, int64_t ldc, int64_t n_cand, int64_t nq, int l1, int descending,
                         const int64_t* gold_off, const int32_t* gold_ids, const int64_t* filt_off, const int32_t* filt_ids, float* gscore,
                         float* fscore, int32_t* counts, const int64_t* rel, const float* Nrm, int64_t ldn, int64_t n_rel, float* wtab,
                         int64_t ldw, hipStream_t st, const char* name) {
  PairsArgs a{};
  a.C0 = C; a.ldc0 = ldc; a.QW = QW; a.n_cand = n_cand; a.nq = nq; a.d = d; a.dq = round4(d); a.l1 = l1;
  a.cvec = (d % 4 == 0) && aligned16(C) && (ldc % 4 == 0);
  a.gold_off = gold_off; a.gold_ids = gold_ids; a.filt_off = filt_off; a.filt_ids = filt_ids; a.gscore = gscore; a.fscore = fscore;
  a.counts = counts; a.descending = descending;
  // TransH with room for the pass's table of e . w (wtab: [n_rel][ldw], ldw >= the candidates rounded up to whole tiles) and rows the
  // table kernel can read as the pair kernels do: the count and list kernels skip their first pass
  if (model == 1 && wtab && rel && Nrm && n_rel > 0 && a.cvec && aligned16(Nrm) && (ldn % 4 == 0) && ldw >= (n_cand + CT - 1) / CT * CT) {
    a.rel = rel; a.wtab = wtab; a.ldw = ldw; a.Nrm = Nrm; a.ldn = ldn; a.n_rel = (int)n_rel;
    return l1 ? launch_pairs_count<3, true>(a, st, name) : launch_pairs_count<3, false>(a, st, name);
  }
  if (model == 1) return l1 ? launch_pairs_count<1, true>(a, st, name) : launch_pairs_count<1, false>(a, st, name);
  return l1 ? launch_pairs_count<0, true>(a, st, name) : launch_pairs_count<0, false>(a, st, name);
}

extern "C" size_t ktup_eval_kg_workspace_bytes(int d, int64_t nq) { return (size_t)nq * 3 * round4(d) * sizeof(float); }

extern "C" size_t ktup_eval_pref_workspace_bytes(int d, int n_pref, int64_t nq, int64_t n_items) {
  // QW[nq][3][d] | QL[nq][P] | QN[nq][P] (padded to 16 B) | the item side (item_side)
  if (pref_row_covers(d, n_pref)) return 64;      // the one-wave-per-pair forward stages nothing
  return (user_side_floats(nq, d, n_pref) + (n_items > 0 ? item_side_floats(n_items, d, n_pref) : 0)) * sizeof(float);
}

extern "C" int ktup_eval_bprmf_scores(const float* U, int64_t ldu, const float* I, int64_t ldi, int d, const int64_t* u_ids,
                                      int64_t nq, int64_t n_items, float* out, int64_t ldo, void* stream) {
  KTUP_REQUIRE(d > 0 && nq >= 0 && n_items >= 0, "ktup_eval_bprmf_scores: bad sizes");
  if (nq == 0 || n_items == 0) return KTUP_OK;
  KTUP_REQUIRE(U && I && u_ids && out && ldo >= n_items, "ktup_eval_bprmf_scores: bad argument");
  const int vec = (d % 4 == 0) && aligned16(U) && aligned16(I) && ldu % 4 == 0 && ldi % 4 == 0;
  dim3 grid((unsigned)((n_items + 63) / 64), (unsigned)((nq + 63) / 64));
  hipLaunchKernelGGL(bprmf_eval_kernel, grid, dim3(256), 0, (hipStream_t)stream, U, ldu, I, ldi, d, u_ids, nq, n_items, out, ldo,
                     vec);
  return check_launch("ktup_eval_bprmf_scores");
}

extern "C" int ktup_eval_transe_scores(const float* E, int64_t lde, const float* R, int64_t ldr, int d, const float* C,
                                       int64_t ldc, int64_t n_cand, const int64_t* q, const int64_t* r, int64_t nq, int l1,
                                       int head, float* out, int64_t ldo, float* ws, void* stream) {
  return kg_eval(0, "ktup_eval_transe_scores", E, lde, R, ldr, nullptr, 0, d, C, ldc, n_cand, q, r, nq, l1, head, out, ldo, ws,
                 stream);
}

extern "C" int ktup_eval_transh_scores(const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                                       int d, const float* C, int64_t ldc, int64_t n_cand, const int64_t* q, const int64_t* r,
                                       int64_t nq, int l1, int head, float* out, int64_t ldo, float* ws, void* stream) {
  return kg_eval(1, "ktup_eval_transh_scores", E, lde, R, ldr, Nrm, ldn, d, C, ldc, n_cand, q, r, nq, l1, head, out, ldo, ws,
                 stream);
}


extern "C" size_t ktup_eval_transr_workspace_bytes(int d, int64_t nq, int64_t n_ent, int n_rel) {
  // QW[nq][3][dq] | PE[n_rel][n_ent][dq] | rel_off[n_rel + 1] | qperm[nq]
  return ((size_t)nq * 3 * round4(d) + (size_t)n_rel * n_ent * round4(d)) * sizeof(float) +
         pad4((size_t)n_rel + 1 + (size_t)nq) * sizeof(int32_t);
}

// The entity side of K14 for one evaluation pass (it does not depend on the queries): `ents_ws` receives either |M_rho e_j|^2 for
// every (relation, entity) -- the squared-L2 matrix-core route -- or the projected table PE[rho][j] = M_rho e_j (L1 and the other
// shapes).  ktup_eval_transr_scores recomputes this per call unless it is handed the prepared workspace: per 512 queries that is
// 140 of 217 us (L2) and 370 of 430 us (L1).
extern "C" size_t ktup_eval_transr_entities_workspace_bytes(int d, int64_t n_ent, int n_rel) {
  return (size_t)n_rel * n_ent * round4(d) * sizeof(float);
}

namespace {

bool transr_mc_route(int d, int l1, const float* E, int64_t lde) {
  return !l1 && (d == 64 || d == 100 || d == 128) && aligned16(E) && lde % 4 == 0 && ktup::opt_eval_mc();
}

// norms[n_rel][n_ent] at the head of `region` (`avail` bytes); the rest of the region is this route's scratch.
// Returns KTUP_OK / an error, or 1 when the region is too small or the shape has no matrix-core kernel.
int transr_entity_norms(const char* name, const float* E, int64_t lde, const float* M, int64_t ldm, int d, int64_t n_ent, int n_rel,
                        float* region, size_t avail, hipStream_t st) {
  const int64_t n = (int64_t)n_rel * n_ent;
  float* norms = region;
  int64_t* hid = reinterpret_cast<int64_t*>(((uintptr_t)(norms + pad4((size_t)n)) + 15) & ~(uintptr_t)15);
  int64_t* rid = hid + n;
  void* bws = rid + n;
  const size_t need = (size_t)((char*)bws - (char*)region) + ktup::transr_mc_workspace_bytes(n, n_rel);
  if (need > avail) return 1;
  hipLaunchKernelGGL(transr_all_pairs_ids_kernel, dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, n_ent, n, hid, rid);
  if (int e = check_launch(name)) return e;
  return ktup::transr_fwd_mc(E, lde, nullptr, 0, M, ldm, n_rel, d, hid, nullptr, rid, n, 0, norms, bws, st, name);
}

int transr_entity_project(const char* name, const float* E, int64_t lde, const float* M, int64_t ldm, int d, int64_t n_ent, int n_rel,
                          float* PE, hipStream_t st) {
  const int dq = round4(d);
  const size_t lds = (size_t)(dq / 4) * CT * 16;
  KTUP_REQUIRE(lds <= 64 * 1024 && (size_t)n_rel * 4 <= 64 * 1024, "%s: embedding_size / n_rel too large", name);
  const int evec = (d % 4 == 0) && aligned16(E) && lde % 4 == 0;
  const int mvec = (d % 4 == 0) && aligned16(M) && ldm % 4 == 0;
  hipLaunchKernelGGL(transr_project_kernel, dim3((unsigned)((n_ent + CT - 1) / CT), (unsigned)n_rel), dim3(256), lds, st, E, lde,
                     M, ldm, d, dq, n_ent, evec, mvec, PE);
  return check_launch(name);
}

}  // namespace

extern "C" int ktup_eval_transr_prepare(const float* E, int64_t lde, const float* M, int64_t ldm, int d, int64_t n_ent, int n_rel,
                                        int l1, float* ents_ws, void* stream) {
  const char* name = "ktup_eval_transr_prepare";
  KTUP_REQUIRE(d > 0 && n_ent >= 0 && n_rel > 0, "%s: bad sizes", name);
  if (n_ent == 0) return KTUP_OK;
  KTUP_REQUIRE(E && M && ents_ws && aligned16(ents_ws) && ldm >= (int64_t)d * d, "%s: bad argument", name);
  hipStream_t st = (hipStream_t)stream;
  if (transr_mc_route(d, l1, E, lde)) {
    const int rc = transr_entity_norms(name, E, lde, M, ldm, d, n_ent, n_rel, ents_ws, ktup_eval_transr_entities_workspace_bytes(d, n_ent, n_rel), st);
    if (rc != 1) return rc;
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: catalogue too small for the matrix-core route's scratch (n_rel x n_ent x d)", name);
  }
  return transr_entity_project(name, E, lde, M, ldm, d, n_ent, n_rel, ents_ws, st);
}

// ents_ws: NULL, or what ktup_eval_transr_prepare left for the SAME tables, l1 and shape.
extern "C" int ktup_eval_transr_scores(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm,
                                       int d, int64_t n_ent, int n_rel, const int64_t* q, const int64_t* r, int64_t nq, int l1,
                                       int head, float* out, int64_t ldo, float* ws, const float* ents_ws, void* stream) {
  const char* name = "ktup_eval_transr_scores";
  KTUP_REQUIRE(d > 0 && nq >= 0 && n_ent >= 0 && n_rel > 0, "%s: bad sizes", name);
  if (nq == 0 || n_ent == 0) return KTUP_OK;
  KTUP_REQUIRE(E && R && M && q && r && out && ws && ldo >= n_ent && ldm >= (int64_t)d * d, "%s: bad argument", name);
  hipStream_t st = (hipStream_t)stream;
  const int dq = round4(d);
  float* QW = ws;
  float* PE = QW + (size_t)nq * 3 * dq;
  int32_t* rel_off = reinterpret_cast<int32_t*>(PE + (size_t)n_rel * n_ent * dq);
  int32_t* qperm = rel_off + n_rel + 1;
  hipLaunchKernelGGL(kg_query_prep_kernel, dim3(grid_for((nq + 3) / 4)), dim3(256), 0, st, 2, E, lde, R, ldr, M, ldm, d, dq, q, r,
                     nq, head, QW);
  if (int e = check_launch(name)) return e;
  if (transr_mc_route(d, l1, E, lde)) {
    // squared L2 on the matrix cores: queries folded through M_r^T, the pair term one GEMM against the UNPROJECTED entity table,
    // |M_r e|^2 from the prepared workspace or computed here (side arrays and scratch live in the PE region of `ws`, which this
    // route does not otherwise use and which is far larger than they need)
    float* qc = PE;                                                       // [nq] |c|^2 of the folded queries
    int32_t* qrel = reinterpret_cast<int32_t*>(qc + pad4((size_t)nq));    // [nq] their relation ids
    const float* norms = ents_ws;
    int rc = KTUP_OK;
    if (!norms) {
      float* region = reinterpret_cast<float*>(qrel + pad4((size_t)nq));
      const size_t used = (size_t)((char*)region - (char*)PE), total = (size_t)n_rel * n_ent * dq * sizeof(float);
      rc = used < total ? transr_entity_norms(name, E, lde, M, ldm, d, n_ent, n_rel, region, total - used, st) : 1;
      norms = region;
    }
    if (rc == KTUP_OK) {
      hipLaunchKernelGGL(transr_query_fold_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, QW, dq, M, ldm, d, r, nq, qc, qrel);
      if (int e = check_launch(name)) return e;
      rc = ktup::pairs_kg_l2_mc(0, QW, dq, E, lde, d, nq, n_ent, out, ldo, st, name, qc, norms, qrel);
    }
    if (rc != 1) return rc;
    KTUP_REQUIRE(!ents_ws, "%s: the prepared workspace holds norms, but this shape needs the projected table", name);
    // (not an instantiated shape after all: c' may have overwritten c, so rebuild the queries for the VALU route)
    hipLaunchKernelGGL(kg_query_prep_kernel, dim3(grid_for((nq + 3) / 4)), dim3(256), 0, st, 2, E, lde, R, ldr, M, ldm, d, dq, q, r,
                       nq, head, QW);
  }
  const float* PEc = ents_ws;
  if (!PEc) {
    if (int e = transr_entity_project(name, E, lde, M, ldm, d, n_ent, n_rel, PE, st)) return e;
    PEc = PE;
  }
  hipLaunchKernelGGL(rel_bucket_kernel, dim3(1), dim3(256), (size_t)n_rel * 4, st, r, nq, n_rel, rel_off, qperm);
  if (int e = check_launch(name)) return e;
  PairsArgs a{};
  a.C0 = PEc; a.ldc0 = dq; a.QW = QW; a.n_cand = n_ent; a.nq = nq; a.d = d; a.dq = dq; a.l1 = l1; a.out = out; a.ldo = ldo;
  a.cvec = 1; a.qperm = qperm; a.rel_off = rel_off; a.rel_stride = (int64_t)n_ent * dq;
  return launch_pairs<0>(a, st, name, n_rel);
}

// TUP (E == NULL) / KTUP all-item scores.  `pref_ws` is the ktup_pref_prepare workspace; `item2ent` has one entry per
// ROW of I (the caller passes the evaluateRec pairing, jTransUP.py:174); `uniform` is (nq x n_items x n_pref).
namespace {

struct ItemSide { float *CW0, *CW1, *CW2, *CL, *CN, *consts; };

// items_ws layout: CW0 | CW1 | CW2 [N][d] | CL [N][P] | CN [N][P] | consts [4][32]
// (v - RV, v, NV, item logits, v . Cn_p, and the hard gate's per-preference constants |Ar_p|^2, Ar_p . Cn_p, |Cn_p|^2 - 2, {4 beta})
ItemSide item_side(float* base, int64_t n_items, int d, int P) {
  ItemSide s;
  s.CW0 = base; s.CW1 = s.CW0 + (size_t)n_items * d; s.CW2 = s.CW1 + (size_t)n_items * d; s.CL = s.CW2 + (size_t)n_items * d;
  s.CN = s.CL + pad4((size_t)n_items * P); s.consts = s.CN + pad4((size_t)n_items * P);
  return s;
}

// |Ar_p|^2, Ar_p . Cn_p, |Cn_p|^2 - 2 of the prepared tables, and 4 beta (beta = 1/2 with an entity side: ktup_pref_prepare)
__global__ __launch_bounds__(256) void pref_consts_kernel(const float* __restrict__ ws, int ppad, int dp, int P, int d, float beta4,
                                                          float* __restrict__ consts) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* Ar = ws + (size_t)ppad * dp;
  const float* Cn = ws + (size_t)(ppad + P) * dp;
  for (int p = w; p < 32; p += 4) {
    float rr = 0.f, rn = 0.f, nn = 0.f;
    if (p < P)
      for (int k = lane; k < d; k += 64) {
        const float r = Ar[(size_t)p * dp + k], n = Cn[(size_t)p * dp + k];
        rr = fmaf(r, r, rr); rn = fmaf(r, n, rn); nn = fmaf(n, n, nn);
      }
    rr = wave_sum(rr); rn = wave_sum(rn); nn = wave_sum(nn);
    if (lane == 0) { consts[p] = rr; consts[32 + p] = rn; consts[64 + p] = nn - 2.f; consts[96 + p] = beta4; }
  }
}

int pref_items_project(const char* name, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                       const float* pref_ws, int n_pref, int d, int64_t n_items, const ItemSide& it, hipStream_t st) {
  const PrefGeom g = pref_geom(d, n_pref);
  if (!g.ok) return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size must be a multiple of 4 in [4, 256] (got %d)", name, d);
  KTUP_REQUIRE(I && pref_ws && it.CW0, "%s: bad argument", name);
  KTUP_REQUIRE((E == nullptr) == (item2ent == nullptr), "%s: E and item2ent must be given together", name);
  KTUP_REQUIRE(aligned16(I) && aligned16(E) && aligned16(pref_ws) && aligned16(it.CW0) && ldi % 4 == 0 && (!E || lde % 4 == 0),
               "%s: tables must be 16-byte aligned with pitches %% 4 == 0", name);
  hipLaunchKernelGGL(pref_project_kernel, dim3(grid_for((n_items + 3) / 4)), dim3(256), 0, st, I, ldi, E, lde, item2ent,
                     (const int64_t*)nullptr, n_items, d, n_pref, pref_ws, g.ppad, g.dp, -1.0f, (int64_t)d, it.CW0, it.CW1, it.CW2, it.CL, it.CN);
  hipLaunchKernelGGL(pref_consts_kernel, dim3(1), dim3(256), 0, st, pref_ws, g.ppad, g.dp, n_pref < 32 ? n_pref : 32, d, E ? 2.0f : 4.0f, it.consts);
  return check_launch(name);
}

// users' projections + the pair kernels, the item side already in `it`
int pref_scores_tail(const char* name, const float* U, int64_t ldu, const float* pref_ws, int n_pref, int d, const int64_t* u_ids,
                     int64_t nq, int64_t n_items, int l1, int gumbel_mode, const float* uniform, uint64_t seed, uint64_t offset,
                     float* out, int64_t ldo, const ItemSide& it, float* ws, hipStream_t st) {
  const PrefGeom g = pref_geom(d, n_pref);
  if (!g.ok) return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size must be a multiple of 4 in [4, 256] (got %d)", name, d);
  KTUP_REQUIRE(U && pref_ws && u_ids && out && ws && ldo >= n_items, "%s: bad argument", name);
  KTUP_REQUIRE(aligned16(U) && aligned16(pref_ws) && aligned16(ws) && ldu % 4 == 0, "%s: tables must be 16-byte aligned with pitches %% 4 == 0", name);
  KTUP_REQUIRE(gumbel_mode >= KTUP_GUMBEL_OFF && gumbel_mode <= KTUP_GUMBEL_PHILOX_DEV, "%s: bad gumbel_mode", name);
  KTUP_REQUIRE((gumbel_mode != KTUP_GUMBEL_INPUT && gumbel_mode != KTUP_GUMBEL_PHILOX_DEV) || uniform,
               "%s: KTUP_GUMBEL_INPUT / KTUP_GUMBEL_PHILOX_DEV need the `uniform` pointer", name);
  float* QW = ws;
  float* QL = QW + (size_t)nq * 3 * d;
  float* QN = QL + pad4((size_t)nq * n_pref);
  const bool hard_l2 = gumbel_mode != KTUP_GUMBEL_OFF && !l1 && n_pref <= 32;      // the only reader of QN / CN / consts (mode 0)
  float *CW0 = it.CW0, *CW1 = it.CW1, *CW2 = it.CW2, *CL = it.CL;
  {
    // shapes whose staged item vectors pass the LDS of the pair kernels below (the soft gate beyond 212 columns unless the matrix
    // cores take it, the hard gate's tile likewise: d = 256, config 5's width, among them): every (user, item) pair through the
    // one-wave-per-pair forward, the items read from the projected item side (CW1 = v = i (+ e), pitch d)
    bool fits;
    if (gumbel_mode == KTUP_GUMBEL_OFF) {
      const bool mc = !l1 && ktup::opt_eval_mc() && (d == 20 || d == 36 || d == 64 || d == 100 || d == 128);
      fits = mc || (size_t)2 * (d / 4) * CT * 16 <= 160 * 1024;          // (two of the three staged: launch_pairs<2>'s hybrid stage)
    } else {
      const int mode = l1 ? 1 : (n_pref <= 32 ? 0 : 2);
      fits = (mode ? hard_stage_floats<1>(d / 4, n_pref, g.dp / 4) : hard_stage_floats<0>(d / 4, n_pref, g.dp / 4)) * 4 <= 160 * 1024;
      if (n_pref > HARD_PMAX) fits = false;
    }
    if (!fits)
      return pref_row(false, name, U, ldu, CW1, d, nullptr, 0, nullptr, -1, pref_ws, n_pref, d, u_ids, nullptr, nq * n_items, n_items, ldo, l1,
                      gumbel_mode, uniform, seed, offset, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st, g.ppad, g.dp);
  }
  // users: slot 0 = u + RU, slot 1 = u, slot 2 = NU, rows of pitch 3d;   items: v - RV, v, NV, rows of pitch d
  hipLaunchKernelGGL(pref_project_kernel, dim3(grid_for((nq + 3) / 4)), dim3(256), 0, st, U, ldu, (const float*)nullptr,
                     (int64_t)0, (const int32_t*)nullptr, u_ids, nq, d, n_pref, pref_ws, g.ppad, g.dp, 1.0f, (int64_t)3 * d, QW,
                     QW + d, QW + 2 * d, QL, hard_l2 ? QN : (float*)nullptr);
  if (int e = check_launch(name)) return e;
  if (gumbel_mode == KTUP_GUMBEL_OFF) {
    if (!l1) {     // squared L2: six (users x items) GEMMs on the matrix cores (ktup_eval_mc.hip); option eval_mc = 0 for A/B runs
      if (ktup::opt_eval_mc()) {
        const int rc = ktup::pairs_l2_mc(QW, CW0, CW1, CW2, d, nq, n_items, out, ldo, st, name);
        if (rc != 1) return rc;
      }
    }
    PairsArgs a{};
    a.C0 = CW0; a.C1 = CW1; a.C2 = CW2; a.ldc0 = a.ldc1 = a.ldc2 = d;
    a.QW = QW; a.n_cand = n_items; a.nq = nq; a.d = d; a.dq = d; a.l1 = l1; a.out = out; a.ldo = ldo; a.cvec = 1;
    return launch_pairs<2>(a, st, name);
  }
  HardArgs h{};
  h.V = CW1; h.LV = CL; h.QW = QW; h.QL = QL; h.ws = pref_ws; h.ppad = g.ppad; h.dp = g.dp; h.P = n_pref; h.d = d;
  h.n_cand = n_items; h.nq = nq; h.l1 = l1; h.gumbel = gumbel_mode; h.uniform = uniform; h.seed = seed; h.offset = offset;
  h.out = out; h.ldo = ldo; h.QN = QN; h.VN = it.CN; h.consts = it.consts;
  const int mode = l1 ? 1 : (n_pref <= 32 ? 0 : 2);
  const size_t lds = (mode ? hard_stage_floats<1>(d / 4, n_pref, g.dp / 4) : hard_stage_floats<0>(d / 4, n_pref, g.dp / 4)) * 4;
  if (lds > 160 * 1024) return set_error(KTUP_ERR_UNSUPPORTED, "%s: hard-gate tile needs %zu B of LDS", name, lds);
  const int64_t tiles = (n_items + CT - 1) / CT;
  int64_t ysplit = (768 + tiles - 1) / tiles;    // ~ the resident workgroups (3 per CU): each stages 47 KB, so not many more
  if (ysplit > (nq + HARD_NT / 64 - 1) / (HARD_NT / 64)) ysplit = (nq + HARD_NT / 64 - 1) / (HARD_NT / 64);
  if (ysplit < 1) ysplit = 1;
  const dim3 hgrid((unsigned)tiles, (unsigned)ysplit);
  auto launch = [&](auto kern) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, hgrid, dim3(HARD_NT), lds, st, h);
  };
  if (mode == 0) launch(pairs_hard_kernel<0>); else if (mode == 1) launch(pairs_hard_kernel<1>); else launch(pairs_hard_kernel<2>);
  return check_launch(name);
}

}  // namespace

extern "C" size_t ktup_eval_pref_items_workspace_bytes(int d, int n_pref, int64_t n_items) {
  return item_side_floats(n_items, d, n_pref) * sizeof(float);
}

extern "C" int ktup_eval_pref_items_prepare(const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                                            const float* pref_ws, int n_pref, int d, int64_t n_items, float* items_ws, void* stream) {
  const char* name = "ktup_eval_pref_items_prepare";
  KTUP_REQUIRE(n_items >= 0, "%s: bad sizes", name);
  if (n_items == 0) return KTUP_OK;
  return pref_items_project(name, I, ldi, E, lde, item2ent, pref_ws, n_pref, d, n_items, item_side(items_ws, n_items, d, n_pref), (hipStream_t)stream);
}

extern "C" int ktup_eval_pref_scores_prepared(const float* U, int64_t ldu, const float* pref_ws, int n_pref, int d,
                                              const int64_t* u_ids, int64_t nq, int64_t n_items, int l1, int gumbel_mode,
                                              const float* uniform, uint64_t seed, uint64_t offset, float* out, int64_t ldo,
                                              const float* items_ws, float* ws, void* stream) {
  const char* name = "ktup_eval_pref_scores_prepared";
  KTUP_REQUIRE(nq >= 0 && n_items >= 0, "%s: bad sizes", name);
  if (nq == 0 || n_items == 0) return KTUP_OK;
  KTUP_REQUIRE(items_ws && aligned16(items_ws), "%s: item-side workspace missing or unaligned", name);
  return pref_scores_tail(name, U, ldu, pref_ws, n_pref, d, u_ids, nq, n_items, l1, gumbel_mode, uniform, seed, offset, out, ldo,
                          item_side(const_cast<float*>(items_ws), n_items, d, n_pref), ws, (hipStream_t)stream);
}

// K16 + K17 for a whole evaluation pass in one sweep (ktup_eval_pass.hip): operand rows in preference space, then the fused score +
// filtered top-n kernel -- no (users x items) matrix.  Soft gate + squared L2 at d in {64, 100, 128}, n_pref <= 32 only (KTUP_ERR_UNSUPPORTED otherwise:
// the caller keeps ktup_eval_pref_scores_prepared + ktup_eval_topk_filtered per batch).
extern "C" size_t ktup_eval_pref_topk_workspace_bytes(int d, int n_pref, int64_t nq, int64_t n_items, int topn) {
  if (nq <= 0 || topn <= 0 || n_items < 0) return 0;
  return ktup::eval_pass_pspace_bytes(d, n_pref, nq, n_items, topn);
}

extern "C" int ktup_eval_pref_topk(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                   const int32_t* item2ent, const float* pref_ws, int n_pref, int d, const int64_t* u_ids, int64_t nq,
                                   int64_t n_items, int l1, const int64_t* filt_off, const int32_t* filt_ids, int topn,
                                   int32_t* top_ids, float* top_scores, float* ws, void* stream) {
  const char* name = "ktup_eval_pref_topk";
  KTUP_REQUIRE(nq >= 0 && n_items >= 0 && topn > 0, "%s: bad sizes", name);
  if (nq == 0) return KTUP_OK;
  const PrefGeom g = pref_geom(d, n_pref);
  if (!g.ok || l1 || n_items == 0) return set_error(KTUP_ERR_UNSUPPORTED, "%s: squared-L2 soft gate, d %% 4 == 0 only", name);
  KTUP_REQUIRE(U && I && pref_ws && u_ids && top_ids && ws && ((filt_off == nullptr) || filt_ids), "%s: null pointer argument", name);
  KTUP_REQUIRE((E == nullptr) == (item2ent == nullptr), "%s: E and item2ent must be given together", name);
  KTUP_REQUIRE(aligned16(pref_ws) && aligned16(ws), "%s: workspaces must be 16-byte aligned", name);
  const int rc = ktup::eval_pass_pspace(U, ldu, u_ids, nq, I, ldi, E, lde, item2ent, n_items, pref_ws, g.ppad, g.dp, n_pref, d, filt_off,
                                        filt_ids, topn, ws, top_ids, top_scores, (hipStream_t)stream, name);
  if (rc == 1) return set_error(KTUP_ERR_UNSUPPORTED, "%s: no fused pass kernel for d=%d, n_pref=%d, topn=%d", name, d, n_pref, topn);
  return rc;
}

extern "C" int ktup_eval_pref_scores(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                     const int32_t* item2ent, const float* pref_ws, int n_pref, int d, const int64_t* u_ids,
                                     int64_t nq, int64_t n_items, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                                     uint64_t offset, float* out, int64_t ldo, float* ws, void* stream) {
  const char* name = "ktup_eval_pref_scores";
  KTUP_REQUIRE(nq >= 0 && n_items >= 0, "%s: bad sizes", name);
  if (nq == 0 || n_items == 0) return KTUP_OK;
  if (pref_row_covers(d, n_pref)) {       // rows beyond 256 columns: every (user, item) pair through the one-wave-per-pair forward
    KTUP_REQUIRE(U && I && pref_ws && u_ids && out && ldo >= n_items, "%s: bad argument", name);
    KTUP_REQUIRE((E == nullptr) == (item2ent == nullptr), "%s: E and item2ent must be given together", name);
    KTUP_REQUIRE(aligned16(U) && aligned16(I) && aligned16(E) && aligned16(pref_ws) && ldu % 4 == 0 && ldi % 4 == 0 && (!E || lde % 4 == 0),
                 "%s: tables must be 16-byte aligned with pitches %% 4 == 0", name);
    KTUP_REQUIRE(gumbel_mode >= KTUP_GUMBEL_OFF && gumbel_mode <= KTUP_GUMBEL_PHILOX_DEV, "%s: bad gumbel_mode", name);
    KTUP_REQUIRE((gumbel_mode != KTUP_GUMBEL_INPUT && gumbel_mode != KTUP_GUMBEL_PHILOX_DEV) || uniform,
                 "%s: KTUP_GUMBEL_INPUT / KTUP_GUMBEL_PHILOX_DEV need the `uniform` pointer", name);
    return pref_row(false, name, U, ldu, I, ldi, E, lde, item2ent, -1, pref_ws, n_pref, d, u_ids, nullptr, nq * n_items, n_items, ldo, l1,
                    gumbel_mode, uniform, seed, offset, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
  }
  KTUP_REQUIRE(ws && aligned16(ws), "%s: bad argument", name);
  // QW[nq][3][d] | QL[nq][P] (padded to 16 B) | the item side
  const ItemSide it = item_side(ws + user_side_floats(nq, d, n_pref), n_items, d, n_pref);
  if (int e = pref_items_project(name, I, ldi, E, lde, item2ent, pref_ws, n_pref, d, n_items, it, (hipStream_t)stream)) return e;
  return pref_scores_tail(name, U, ldu, pref_ws, n_pref, d, u_ids, nq, n_items, l1, gumbel_mode, uniform, seed, offset, out, ldo, it,
                          ws, (hipStream_t)stream);
}


// ---- the hard gate's whole pass (see sweep_hard_kernel)
extern "C" size_t ktup_eval_pref_topk_hard_workspace_bytes(int d, int n_pref, int64_t nq, int64_t n_items, int topn) {
  if (d <= 0 || n_pref <= 0 || nq < 0 || n_items < 0 || topn <= 0) return 0;
  return (item_side_floats(n_items, d, n_pref) + user_side_floats(nq, d, n_pref)) * sizeof(float) + (size_t)nq * 8 * topn * sizeof(uint64_t) + 64;
}

extern "C" int ktup_eval_pref_topk_hard(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                        const int32_t* item2ent, const float* pref_ws, int n_pref, int d, const int64_t* u_ids, int64_t nq,
                                        int64_t n_items, int l1, int gumbel_mode, const float* uniform, uint64_t seed, uint64_t offset,
                                        const int64_t* filt_off, const int32_t* filt_ids, int topn, int32_t* top_ids, float* top_scores,
                                        float* ws, void* stream) {
  const char* name = "ktup_eval_pref_topk_hard";
  KTUP_REQUIRE(nq >= 0 && n_items >= 0, "%s: bad sizes", name);
  if (nq == 0) return KTUP_OK;
  KTUP_REQUIRE(n_items > 0 && n_items < (1ll << 31) && topn >= 1 && topn <= 16 && n_pref <= 32, "%s: needs items, 32-bit item ids, topn <= 16 and at most 32 preferences", name);
  KTUP_REQUIRE(gumbel_mode >= KTUP_GUMBEL_OFF && gumbel_mode <= KTUP_GUMBEL_PHILOX_DEV, "%s: bad gumbel_mode", name);
  KTUP_REQUIRE(U && I && pref_ws && u_ids && top_ids && ws && aligned16(ws), "%s: bad argument", name);
  KTUP_REQUIRE(gumbel_mode == KTUP_GUMBEL_OFF || gumbel_mode == KTUP_GUMBEL_PHILOX || uniform, "%s: KTUP_GUMBEL_INPUT / KTUP_GUMBEL_PHILOX_DEV need the `uniform` pointer", name);
  KTUP_REQUIRE((filt_off == nullptr) || filt_ids, "%s: filter offsets without ids", name);
  const PrefGeom g = pref_geom(d, n_pref);
  if (!g.ok) return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size must be a multiple of 4 in [4, 256] (got %d)", name, d);
  hipStream_t st = (hipStream_t)stream;
  float* items_ws = ws;
  const ItemSide it = item_side(items_ws, n_items, d, n_pref);
  if (int e = pref_items_project(name, I, ldi, E, lde, item2ent, pref_ws, n_pref, d, n_items, it, st)) return e;
  float* QW = items_ws + item_side_floats(n_items, d, n_pref);
  float* QL = QW + (size_t)nq * 3 * d;
  float* QN = QL + pad4((size_t)nq * n_pref);
  uint64_t* part = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(QN + pad4((size_t)nq * n_pref)) + 15) & ~(uintptr_t)15);
  KTUP_REQUIRE(aligned16(U) && aligned16(pref_ws) && ldu % 4 == 0, "%s: tables must be 16-byte aligned with pitches %% 4 == 0", name);
  hipLaunchKernelGGL(pref_project_kernel, dim3(grid_for((nq + 3) / 4)), dim3(256), 0, st, U, ldu, (const float*)nullptr,
                     (int64_t)0, (const int32_t*)nullptr, u_ids, nq, d, n_pref, pref_ws, g.ppad, g.dp, 1.0f, (int64_t)3 * d, QW,
                     QW + d, QW + 2 * d, QL, (l1 || gumbel_mode == KTUP_GUMBEL_OFF) ? (float*)nullptr : QN);
  if (int e = check_launch(name)) return e;
  if (gumbel_mode == KTUP_GUMBEL_OFF) {           // the soft gate's pair arithmetic (pairs_kernel<2>) with the top-n in its epilogue
    SweepSoftArgs ss{};
    PairsArgs& pa = ss.p;
    pa.C0 = it.CW0; pa.C1 = it.CW1; pa.C2 = it.CW2; pa.ldc0 = pa.ldc1 = pa.ldc2 = d;
    pa.QW = QW; pa.n_cand = n_items; pa.nq = nq; pa.d = d; pa.dq = d; pa.l1 = l1; pa.cvec = 1;
    ss.filt_off = filt_off; ss.filt_ids = filt_ids; ss.topn = topn; ss.part = part;
    const int64_t ub = (nq + SS_NW * SS_UW - 1) / (SS_NW * SS_UW);
    int ns = (int)(256 / ub);                       // one workgroup per CU is resident (the stage is 77 KB at d = 100): one round
    if (ns > 8) ns = 8;
    if (ns < 1) ns = 1;
    const int64_t stg = (n_items + CT - 1) / CT;
    if (ns > stg) ns = (int)stg;
    ss.split_items = ((stg + ns - 1) / ns) * CT;
    ns = (int)((n_items + ss.split_items - 1) / ss.split_items);
    ss.nsplit = ns;
    ss.bm_words = (int)((ss.split_items + 31) / 32);
    const size_t wave_b = SS_NW * sweep_wave_bytes(SS_UW, ss.bm_words, topn);
    size_t sl = (size_t)3 * (d / 4) * CT * 16 + wave_b;
    const bool whole = sl <= 160 * 1024 && 3 * (d / 4) * CT <= 8 * SS_NW * 64;
    if (!whole) {                                   // two arrays staged, the third streamed (d <= 256)
      sl = (size_t)2 * (d / 4) * CT * 16 + wave_b;
      if (sl > 160 * 1024 || 2 * (d / 4) * CT > 8 * SS_NW * 64)
        return set_error(KTUP_ERR_UNSUPPORTED, "%s: the stage needs %zu B of LDS (per-batch calls remain)", name, sl);
    }
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl);
      hipLaunchKernelGGL(kern, dim3((unsigned)ub, (unsigned)ns), dim3(SS_NW * 64), sl, st, ss);
    };
    if (whole) { if (l1) go(sweep_soft_kernel<true>); else go(sweep_soft_kernel<false>); }
    else { if (l1) go(sweep_soft_kernel<true, true>); else go(sweep_soft_kernel<false, true>); }
    if (int e = check_launch(name)) return e;
    return ktup::launch_topk_merge(part, nq, ns, topn, top_ids, top_scores, st, name);
  }
  SweepHardArgs sa{};
  HardArgs& h = sa.h;
  h.V = it.CW1; h.LV = it.CL; h.QW = QW; h.QL = QL; h.ws = pref_ws; h.ppad = g.ppad; h.dp = g.dp; h.P = n_pref; h.d = d;
  h.n_cand = n_items; h.nq = nq; h.l1 = l1; h.gumbel = gumbel_mode; h.uniform = uniform; h.seed = seed; h.offset = offset;
  h.QN = QN; h.VN = it.CN; h.consts = it.consts;
  sa.filt_off = filt_off; sa.filt_ids = filt_ids; sa.topn = topn; sa.part = part;
  const int64_t ublocks = (nq + SW_NW * SW_UW - 1) / (SW_NW * SW_UW);
  int nsplit = (int)(768 / ublocks);                     // three workgroups per CU are resident (LDS): one round
  if (nsplit > 8) nsplit = 8;
  if (nsplit < 1) nsplit = 1;
  const int64_t stages = (n_items + CT - 1) / CT;
  if (nsplit > stages) nsplit = (int)stages;
  sa.split_items = ((stages + nsplit - 1) / nsplit) * CT;
  nsplit = (int)((n_items + sa.split_items - 1) / sa.split_items);
  sa.nsplit = nsplit;
  sa.bm_words = (int)((sa.split_items + 31) / 32);
  const size_t lds = (l1 ? hard_stage_floats<1>(d / 4, n_pref, g.dp / 4) : hard_stage_floats<0>(d / 4, n_pref, g.dp / 4)) * 4 +
                     SW_NW * sweep_wave_bytes(SW_UW, sa.bm_words, topn);
  if (lds > 160 * 1024 || ublocks > 0x7fffffffll)
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: the stage needs %zu B of LDS (per-batch calls remain)", name, lds);
  const dim3 grid((unsigned)ublocks, (unsigned)nsplit);
  auto launch = [&](auto kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(SW_NW * 64), lds, st, sa);
  };
  if (l1) launch(sweep_hard_kernel<1>); else launch(sweep_hard_kernel<0>);
  if (int e = check_launch(name)) return e;
  return ktup::launch_topk_merge(part, nq, nsplit, topn, top_ids, top_scores, st, name);
}
