// K5/K6/K7 for rows wider than the tile kernels hold (-embedding_size > 256): TUP / KTUP score, its backward, and the all-item
// evaluation scores, for ANY width that is a multiple of 4 (ops stages other widths with a zero tail) and either gate / distance.
// Also the per-batch all-item scores of the widths in (212, 256] whose three staged item vectors pass the LDS of the pair kernels
// (ktup_eval.hip pref_scores_tail: BASELINE config 5's d = 256 among them).
//   reference: jTransUP/models/transUP.py:69-102,105-170 ; jTransUP/models/jTransUP.py:122-143,163-191,250-315 ;
//              jTransUP/models/base.py:52 (embedding_size is any integer)
//
// The arithmetic is that of ktup_score_pref.hip (see its header: x = u + v, q = u - v, l = A x / 2, w = l or onehot(argmax(l + g)),
// r = beta A^T w, n = beta C^T w, s = q . n, z = q + r - s n, score = sum |z| or sum z^2, and the backward written above
// pref_bwd_kernel).  The mapping is the opposite one: ONE WAVE PER PAIR, the lanes across the row's 16-byte chunks (chunk c = lane +
// 64 k), so nothing about a pair is held per coordinate -- no LDS tile, no register array sized by d.  A quantity that needs a sum
// over the whole row before the coordinates can go on (the logits, then s, then a = gz . n, then the per-preference gw) ends a PASS
// over the row; the next pass re-reads u, v (, e) and rebuilds r and n from the prepared tables (cache hits: a row is d x 4 bytes, the
// tables 3 P d x 4).  Forward: 3 passes, backward: 5.  This is the route for widths nobody tunes for; it is bound by the L2's table
// reads, and the row gradients leave by float atomics.
#include "ktup_pref_geom.h"

using namespace ktup;

namespace {

constexpr int NWV = 4;  // waves (= pairs in flight) per workgroup

struct RowArgs {
  const float4 *U, *I, *E;
  int64_t ldu4, ldi4, lde4;
  const int32_t* item2ent;
  const float4 *Alog, *Ar, *Cn;   // prepared tables (ktup_pref_prepare), row pitch tp float4
  int P, nch, tp;
  const int64_t *u_ids, *i_ids;   // i_ids null: the pairs are (u_ids[b], j) for j in [0, n_items), pair index b n_items + j
  int64_t n, n_items, ldo;
  int l1, gumbel;
  const float* uniform;
  uint64_t seed, offset;
  float* score;
  const float* gscore;
  float *gU, *gI, *gE, *gA, *gC;
  int64_t ent_pad;
  float beta;
};

KTUP_DEV float wsum(float v) { return group_sum<64>(v); }

// the lanes of one wave hand values to each other through LDS: DS operations of a wave complete in order, the fences keep the
// compiler from moving them across
KTUP_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

KTUP_DEV float row_uniform(const RowArgs& a, int64_t pair, int p) {
  if (a.gumbel == KTUP_GUMBEL_INPUT) return a.uniform[pair * a.P + p];
  const uint64_t idx = (uint64_t)pair * (uint64_t)a.P + (uint64_t)p + a.offset;
  const uint4 r = Philox(a.seed)(idx >> 2, 0x4b545550ull);
  const uint32_t w = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
  return u01(w);
}

struct Pair {
  const float4 *pu, *pv, *pe;
  int64_t uid, iid, eid;
  KTUP_DEV float4 v(int c) const { return pe ? pv[c] + pe[c] : pv[c]; }
};

KTUP_DEV Pair pair_of(const RowArgs& a, int64_t pr) {
  Pair k;
  if (a.i_ids) { k.uid = a.u_ids[pr]; k.iid = a.i_ids[pr]; }
  else { const int64_t b = pr / a.n_items; k.uid = a.u_ids[b]; k.iid = pr - b * a.n_items; }
  k.eid = a.E ? (int64_t)a.item2ent[k.iid] : 0;
  k.pu = a.U + k.uid * a.ldu4;
  k.pv = a.I + k.iid * a.ldi4;
  k.pe = a.E ? a.E + k.eid * a.lde4 : nullptr;
  return k;
}

// pass 1: the P logits of the pair into lg[0, P) (the noisy logits l + g under the hard gate); returns the hard gate's choice
KTUP_DEV int row_logits(const RowArgs& a, const Pair& k, int64_t pr, int lane, float* lg) {
  const int nch = a.nch, P = a.P;
  for (int p0 = 0; p0 < P; p0 += 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int row[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) row[pp] = min(p0 + pp, P - 1) * a.tp;
    for (int c = lane; c < nch; c += 64) {
      const float4 x = k.pu[c] + k.v(c);
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) acc[pp] += dot4(x, a.Alog[row[pp] + c]);
    }
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const float tot = wsum(acc[pp]);
      if (lane == 0 && p0 + pp < P) lg[p0 + pp] = tot;
    }
  }
  wave_sync();
  if (a.gumbel == KTUP_GUMBEL_OFF) return 0;
  for (int p = lane; p < P; p += 64) lg[p] += gumbel_from_uniform(row_uniform(a, pr, p));
  wave_sync();
  int best = 0;                                        // first maximum, like torch.max (ktup_score_pref.hip row_argmax)
  float bv = lg[0];
  for (int p = 1; p < P; ++p) {
    const float v = lg[p];
    if (v > bv) { bv = v; best = p; }
  }
  return best;
}

// r and n of chunk c: the soft gate mixes every preference row with its logit, the hard gate takes row ps
KTUP_DEV void row_mix(const RowArgs& a, const float* lg, int ps, int c, float4& r, float4& n) {
  if (a.gumbel != KTUP_GUMBEL_OFF) {
    r = a.Ar[ps * a.tp + c];
    n = a.Cn[ps * a.tp + c];
    return;
  }
  r = f4zero(); n = f4zero();
  for (int p = 0; p < a.P; ++p) {
    const float w = lg[p];
    r = fma4(w, a.Ar[p * a.tp + c], r);
    n = fma4(w, a.Cn[p * a.tp + c], n);
  }
}

__global__ __launch_bounds__(NWV * 64) void pref_row_fwd_kernel(RowArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  extern __shared__ __attribute__((aligned(16))) float rsm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* lg = rsm + wv * a.P;
  const bool l1 = a.l1 != 0;
  for (int64_t pr = (int64_t)blockIdx.x * NWV + wv; pr < a.n; pr += (int64_t)gridDim.x * NWV) {
    const Pair k = pair_of(a, pr);
    const int ps = row_logits(a, k, pr, lane, lg);
    float sp = 0.f;
    for (int c = lane; c < a.nch; c += 64) {
      float4 r, n;
      row_mix(a, lg, ps, c, r, n);
      sp += dot4(k.pu[c] - k.v(c), n);
    }
    const float s = wsum(sp);
    float zp = 0.f;
    for (int c = lane; c < a.nch; c += 64) {
      float4 r, n;
      row_mix(a, lg, ps, c, r, n);
      zp += dist4(fma4(-s, n, (k.pu[c] - k.v(c)) + r), l1);
    }
    const float tot = wsum(zp);
    if (lane == 0) {
      if (a.i_ids) a.score[pr] = tot;
      else { const int64_t b = pr / a.n_items; a.score[b * a.ldo + (pr - b * a.n_items)] = tot; }
    }
    wave_sync();                                       // lg is rewritten by the next pair
  }
}

// per-wave LDS of the backward: lg[P] | gl[P] | part[P][64]
__global__ __launch_bounds__(NWV * 64) void pref_row_bwd_kernel(RowArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  extern __shared__ __attribute__((aligned(16))) float rsm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int P = a.P, nch = a.nch, d = nch * 4;
  float* lg = rsm + (size_t)wv * (66 * P);
  float* gl = lg + P;
  float* part = gl + P;
  const bool l1 = a.l1 != 0, hard = a.gumbel != KTUP_GUMBEL_OFF;
  const float beta = a.beta;
  for (int64_t pr = (int64_t)blockIdx.x * NWV + wv; pr < a.n; pr += (int64_t)gridDim.x * NWV) {
    const Pair k = pair_of(a, pr);
    const int ps = row_logits(a, k, pr, lane, lg);
    const float g = a.gscore[pr];
    float sp = 0.f;
    for (int c = lane; c < nch; c += 64) {
      float4 r, n;
      row_mix(a, lg, ps, c, r, n);
      sp += dot4(k.pu[c] - k.v(c), n);
    }
    const float s = wsum(sp);
    float ap = 0.f;
    for (int c = lane; c < nch; c += 64) {
      float4 r, n;
      row_mix(a, lg, ps, c, r, n);
      const float4 gz = g * ddist4(fma4(-s, n, (k.pu[c] - k.v(c)) + r), l1);
      ap += dot4(gz, n);
    }
    const float av = wsum(ap);
    // gw_p = Ar_p . gz + Cn_p . gn  (the tables carry beta): this lane's chunks into part[p][lane], then one sum per preference
    for (int p = 0; p < P; ++p) part[p * 64 + lane] = 0.f;
    for (int c = lane; c < nch; c += 64) {
      float4 r, n;
      row_mix(a, lg, ps, c, r, n);
      const float4 q = k.pu[c] - k.v(c);
      const float4 gz = g * ddist4(fma4(-s, n, q + r), l1);
      const float4 gn = fma4(-av, q, (-s) * gz);
      for (int p = 0; p < P; ++p) part[p * 64 + lane] += dot4(a.Ar[p * a.tp + c], gz) + dot4(a.Cn[p * a.tp + c], gn);
    }
    for (int p = 0; p < P; ++p) {
      const float tot = wsum(part[p * 64 + lane]);
      if (lane == 0) gl[p] = tot;
    }
    wave_sync();
    if (hard) {                                        // gl = y (gw - y . gw), y = softmax(l + g): the ST estimator's backward
      float m = -INFINITY;
      for (int p = 0; p < P; ++p) m = fmaxf(m, lg[p]);
      float den = 0.f, dotyg = 0.f;
      for (int p = 0; p < P; ++p) { const float e = expf(lg[p] - m); den += e; dotyg += e * gl[p]; }
      const float inv = 1.f / den;
      dotyg *= inv;
      wave_sync();                                     // every lane has read gw before lane 0 overwrites it
      if (lane == 0)
        for (int p = 0; p < P; ++p) gl[p] = expf(lg[p] - m) * inv * (gl[p] - dotyg);
      wave_sync();
    }
    float* gu = a.gU + k.uid * a.ldu4 * 4;
    float* gi = a.gI + k.iid * a.ldi4 * 4;
    float* ge = (a.E && k.eid != a.ent_pad) ? a.gE + k.eid * a.lde4 * 4 : nullptr;
    for (int c = lane; c < nch; c += 64) {
      float4 r, n;
      row_mix(a, lg, ps, c, r, n);
      const float4 u = k.pu[c], v = k.v(c);
      const float4 q = u - v, x = u + v;
      const float4 gz = g * ddist4(fma4(-s, n, q + r), l1);
      const float4 gq = fma4(-av, n, gz);
      const float4 gn = fma4(-av, q, (-s) * gz);
      float4 gx = f4zero();
      for (int p = 0; p < P; ++p) gx = fma4(gl[p], a.Alog[p * a.tp + c], gx);       // Alog carries the 1/2
      const float4 gv = gx - gq;
      atomic_add4(gu + 4 * c, gq + gx);
      atomic_add4(gi + 4 * c, gv);
      if (ge) atomic_add4(ge + 4 * c, gv);
      // mixed-table gradients: gA_p += gl_p x / 2 + beta w_p gz ; gC_p += beta w_p gn
      for (int p = 0; p < P; ++p) {
        const float w = hard ? (p == ps ? 1.f : 0.f) : lg[p];
        atomic_add4(a.gA + (int64_t)p * d + 4 * c, fma4(0.5f * gl[p], x, (beta * w) * gz));
        if (w != 0.f) atomic_add4(a.gC + (int64_t)p * d + 4 * c, (beta * w) * gn);
      }
    }
    wave_sync();
  }
}

}  // namespace

namespace ktup {

bool pref_row_covers(int d, int n_pref) { return d > 256 && d % 4 == 0 && n_pref > 0 && n_pref <= 128; }

int pref_row(bool bwd, const char* name, const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
             const int32_t* item2ent, int64_t ent_pad, const float* pref_ws, int n_pref, int d, const int64_t* u_ids, const int64_t* i_ids,
             int64_t n, int64_t n_items, int64_t ldo, int l1, int gumbel_mode, const float* uniform, uint64_t seed, uint64_t offset,
             float* score, const float* gscore, float* gU, float* gI, float* gE, float* gA, float* gC, hipStream_t st, int ppad, int dp) {
  if (d <= 0 || d % 4 || n_pref <= 0 || n_pref > 128)
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: the one-wave-per-pair kernels take a multiple of 4 columns and at most 128 preferences (d=%d, n_pref=%d)", name, d, n_pref);
  if (ppad <= 0) { ppad = n_pref; dp = d; }             // the plain [P][d] blocks of a width beyond 256
  RowArgs a{};
  a.U = reinterpret_cast<const float4*>(U); a.I = reinterpret_cast<const float4*>(I); a.E = reinterpret_cast<const float4*>(E);
  a.ldu4 = ldu / 4; a.ldi4 = ldi / 4; a.lde4 = lde / 4; a.item2ent = item2ent;
  a.tp = dp / 4;
  a.Alog = reinterpret_cast<const float4*>(pref_ws);
  a.Ar = a.Alog + (size_t)ppad * a.tp;
  a.Cn = a.Ar + (size_t)n_pref * a.tp;
  a.P = n_pref; a.nch = d / 4; a.u_ids = u_ids; a.i_ids = i_ids; a.n = n; a.n_items = n_items; a.ldo = ldo;
  a.l1 = l1; a.gumbel = gumbel_mode; a.uniform = uniform; a.seed = seed; a.offset = offset; a.score = score;
  a.gscore = gscore; a.gU = gU; a.gI = gI; a.gE = gE; a.gA = gA; a.gC = gC; a.ent_pad = ent_pad; a.beta = E ? 0.5f : 1.0f;
  const int grid = grid_for((n + NWV - 1) / NWV, 256 * 8);
  if (bwd) {
    const size_t lds = (size_t)NWV * 66 * n_pref * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)pref_row_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(pref_row_bwd_kernel, dim3(grid), dim3(NWV * 64), lds, st, a);
  } else {
    hipLaunchKernelGGL(pref_row_fwd_kernel, dim3(grid), dim3(NWV * 64), (size_t)NWV * n_pref * sizeof(float), st, a);
  }
  return check_launch(name);
}

}  // namespace ktup
