// Config 5, the KG half of the joint schedule (three of every ten steps at joint_ratio 0.7): KTUP's kg step
// (knowledgable_recommendation.py:345-383 over jTransUP.py:144-157: model(pos triples), model(neg triples), marginLoss,
// orthogonalLoss(rel rows, norm rows), normLoss(entity rows), normLoss(rel rows), x kg_lambda) on a ROW-SHARDED entity table, in
// the fixed-shape form of ktup_shard_step.hip: the entity-row gradients of triple k leave as rows k, B + k, 2B + k, 3B + k of GE
// (plain stores, [ph ; pt ; nh ; nt] -- the order of the route's entry list) for the sorted-segment reduce -> norm -> apply walk,
// instead of float atomics into a table-shaped gradient that a 5 M x 256 shard cannot afford.
//
// The relation-side gradients (rel, norm: a few dozen replicated rows hit by every triple) would be B x 2 x 2 x d float atomics on
// ~20 rows.  Instead the triples are walked in RELATION-SORTED order (kg_rel_order_kernel: one counting sort per step): a lane
// group keeps the gradient of its current relation in registers across its consecutive triples, adds it to the workgroup's LDS
// row when the relation is the workgroup's first one (the common case: 16 consecutive sorted triples share a relation) and the
// workgroup flushes that row once -- ~512 x 2 x d atomics per step instead of 4 M.
#include "ktup_rows.h"

using namespace ktup;

namespace {

constexpr int ORD_T = 1024;                 // threads of the sort's one workgroup
constexpr int ORD_H = 16384;                // LDS histogram words: per-wave private copies while n_rel <= 1024

// order[0 .. B) = the triples' indices k sorted by rel[k] (any order inside a relation); n_rel > ORD_H: identity
__global__ __launch_bounds__(ORD_T) void kg_rel_order_kernel(const int64_t* __restrict__ rel, int64_t B, int64_t n_rel, int32_t* __restrict__ order) {
  __shared__ int32_t hist[ORD_H];
  __shared__ int32_t wtot[ORD_T / 64];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (n_rel > ORD_H) {
    for (int64_t k = t; k < B; k += ORD_T) order[k] = (int32_t)k;
    return;
  }
  int nwh = (int)(ORD_H / n_rel);
  nwh = nwh > ORD_T / 64 ? ORD_T / 64 : nwh;
  const int N = (int)n_rel * nwh;
  const int col = wv % nwh;
  for (int i = t; i < N; i += ORD_T) hist[i] = 0;
  __syncthreads();
  auto rel_of = [&](int64_t k) {
    int64_t r = rel[k];
    r = r < 0 ? 0 : (r >= n_rel ? n_rel - 1 : r);
    return (int)r;
  };
  for (int64_t k = t; k < B; k += ORD_T) atomicAdd(&hist[rel_of(k) * nwh + col], 1);
  __syncthreads();
  // exclusive scan of hist[0 .. N): thread t owns PER consecutive words
  const int PER = (N + ORD_T - 1) / ORD_T;
  int32_t mine = 0;
  for (int i = 0; i < PER; ++i) { const int idx = t * PER + i; if (idx < N) mine += hist[idx]; }
  int32_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) wtot[wv] = inc;
  __syncthreads();
  int32_t run = inc - mine;
  for (int k = 0; k < wv; ++k) run += wtot[k];
  for (int i = 0; i < PER; ++i) {
    const int idx = t * PER + i;
    if (idx < N) { const int32_t c = hist[idx]; hist[idx] = run; run += c; }
  }
  __syncthreads();
  for (int64_t k = t; k < B; k += ORD_T) order[atomicAdd(&hist[rel_of(k) * nwh + col], 1)] = (int32_t)k;
}

struct KgRowsArgs {
  const float *E, *R, *Nm; int64_t lde, ldr, ldn;
  const int64_t* ent;            // [ph ; pt ; nh ; nt], B each: rows of E (global ids of a shard, or wire rows of a compact table); an nh / nt
                                 // that is negative or == pad: the corrupted triple shares that entity with the positive one
  int64_t pad;
  const int64_t* rel;            // [pr ; nr]
  const int32_t* order;          // relation-sorted order of the triples (nullptr: as given)
  int64_t B; int nch; bool l1;
  float margin, gscale;
  int regs;                      // bit 0 orthogonalLoss(rel, norm) rows, bit 1 normLoss(entity rows), bit 2 normLoss(relation rows)
  int chunk;                     // consecutive sorted triples per lane group
  float* loss;                   // [4]: margin sum, orth, normE, normR  (accumulated)
  float* GE;                     // 4B x d, row x B + k = gradient of ent[x B + k]'s row from triple k
  float *gR, *gN;
  double* sumsq; int sumsq_slots; // may be null: += sum of |row|^2 over the 4B stored rows
};

struct KgTriple {
  int64_t k, id[4], rid[2];
  bool same[2];                  // the corrupted triple's head / tail IS the positive one's entry
  float4 e[4];
};

template <int GL, bool TRANSH>
__global__ __launch_bounds__(256) void kg_step_rows_kernel(KgRowsArgs a) {
  constexpr int GPB = 256 / GL;
  __shared__ float home[2][4 * GL];                        // the workgroup's first relation: its gR / gN rows
  __shared__ float red[4][4];
  const int lane = threadIdx.x % GL, grp = threadIdx.x / GL;
  const bool on = lane < a.nch;
  const int64_t p_wg = (int64_t)blockIdx.x * GPB * a.chunk;
  if (p_wg >= a.B) return;
  const int64_t hr = a.rel[a.order ? a.order[p_wg] : p_wg];
  for (int i = threadIdx.x; i < 2 * 4 * GL; i += 256) (&home[0][0])[i] = 0.f;
  __syncthreads();
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  float ssq = 0.f;
  const float g1 = a.gscale;
  const int64_t p0 = p_wg + (int64_t)grp * a.chunk, p1 = min(a.B, p0 + (int64_t)a.chunk);
  float4 accR = f4zero(), accN = f4zero();
  int64_t cur = -1;
  auto flush = [&]() {
    if (cur >= 0 && on) {
      if (cur == hr) {
        float* hR = &home[0][4 * lane];
        atomicAdd(hR + 0, accR.x); atomicAdd(hR + 1, accR.y); atomicAdd(hR + 2, accR.z); atomicAdd(hR + 3, accR.w);
        if (TRANSH) {
          float* hN = &home[1][4 * lane];
          atomicAdd(hN + 0, accN.x); atomicAdd(hN + 1, accN.y); atomicAdd(hN + 2, accN.z); atomicAdd(hN + 3, accN.w);
        }
      } else {
        atomic_add4(a.gR + cur * a.ldr + 4 * lane, accR);
        if (TRANSH) atomic_add4(a.gN + cur * a.ldn + 4 * lane, accN);
      }
    }
    accR = f4zero(); accN = f4zero();
  };
  auto fetch = [&](int64_t p, KgTriple& t) {             // ids and the four entity rows of sorted position p
    t.k = a.order ? a.order[p] : p;
#pragma unroll
    for (int x = 0; x < 4; ++x) t.id[x] = a.ent[(int64_t)x * a.B + t.k];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      t.same[x] = t.id[2 + x] < 0 || t.id[2 + x] == a.pad;
      if (t.same[x]) t.id[2 + x] = t.id[x];
    }
    t.rid[0] = a.rel[t.k]; t.rid[1] = a.rel[a.B + t.k];
#pragma unroll
    for (int x = 0; x < 4; ++x) t.e[x] = on ? reinterpret_cast<const float4*>(a.E + t.id[x] * a.lde)[lane] : f4zero();
  };
  KgTriple nx;
  if (p0 < p1) fetch(p0, nx);
  for (int64_t p = p0; p < p1; ++p) {
    const KgTriple t = nx;
    if (p + 1 < p1) fetch(p + 1, nx);                    // the next triple's rows fly under this one's arithmetic
    float4 rr[2], ww[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      rr[x] = on ? reinterpret_cast<const float4*>(a.R + t.rid[x] * a.ldr)[lane] : f4zero();
      ww[x] = (TRANSH && on) ? reinterpret_cast<const float4*>(a.Nm + t.rid[x] * a.ldn)[lane] : f4zero();
    }
    // ---- forward of both triples (jTransUP.py:144-157 = transH.py:58-71; TransE: transE.py:51-63)
    float dh[2] = {0.f, 0.f}, dt[2] = {0.f, 0.f}, sc[2];
    float4 z[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const float4 hh = t.e[2 * x], tt = t.e[2 * x + 1];
      if (TRANSH) {
        dh[x] = group_sum<GL>(dot4(hh, ww[x]));
        dt[x] = group_sum<GL>(dot4(tt, ww[x]));
        const float4 ph = fma4(-dh[x], ww[x], hh), pt = fma4(-dt[x], ww[x], tt);
        z[x] = (ph + rr[x]) - pt;
      } else {
        z[x] = (hh + rr[x]) - tt;
      }
      sc[x] = group_sum<GL>(dist4(z[x], a.l1));
    }
    // ---- marginLoss (utils/loss.py:8-16): sum_k max(pos - neg + margin, 0)
    const float diff = sc[0] - sc[1];
    const bool act = diff + a.margin > 0.f;
    if (lane == 0) part[0] += fmaxf(diff + a.margin, 0.f);
    const float gs[2] = {act ? g1 : 0.f, act ? -g1 : 0.f};
    float4 ge[4], gr[2], gw[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const float4 gz = gs[x] * ddist4(z[x], a.l1);
      if (TRANSH) {
        const float aw = group_sum<GL>(dot4(gz, ww[x]));
        const float sq = dh[x] - dt[x];
        const float4 q = t.e[2 * x] - t.e[2 * x + 1];
        ge[2 * x] = fma4(-aw, ww[x], gz);
        ge[2 * x + 1] = -1.f * ge[2 * x];
        gw[x] = fma4(-aw, q, (-sq) * gz);
      } else {
        ge[2 * x] = gz;
        ge[2 * x + 1] = -1.f * gz;
        gw[x] = f4zero();
      }
      gr[x] = gz;
    }
    if (a.regs & 2) {                 // normLoss(ent rows): sum max(|x|^2 - 1, 0)  (utils/loss.py:21-23)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float s = group_sum<GL>(dot4(t.e[x], t.e[x]));
        if (s - 1.f > 0.f) ge[x] = fma4(2.f * g1, t.e[x], ge[x]);
        if (lane == 0) part[2] += fmaxf(s - 1.f, 0.f);
      }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      if (TRANSH && (a.regs & 1)) {   // orthogonalLoss(rel rows, norm rows): sum (w.r)^2 / |r|^2  (utils/loss.py:18-19)
        const float dot = group_sum<GL>(dot4(rr[x], ww[x])), nr = group_sum<GL>(dot4(rr[x], rr[x]));
        const float c1 = g1 * 2.f * dot / nr, c2 = g1 * 2.f * dot * dot / (nr * nr);
        gr[x] = gr[x] + fma4(-c2, rr[x], c1 * ww[x]);
        gw[x] = fma4(c1, rr[x], gw[x]);
        if (lane == 0) part[1] += dot * dot / nr;
      }
      if (a.regs & 4) {               // normLoss(rel rows)
        const float s = group_sum<GL>(dot4(rr[x], rr[x]));
        if (s - 1.f > 0.f) gr[x] = fma4(2.f * g1, rr[x], gr[x]);
        if (lane == 0) part[3] += fmaxf(s - 1.f, 0.f);
      }
    }
    if (on) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
        if (t.same[x]) { ge[x] = ge[x] + ge[2 + x]; ge[2 + x] = f4zero(); }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (x < 2 || !t.same[x - 2]) reinterpret_cast<float4*>(a.GE + ((int64_t)x * a.B + t.k) * (4 * (int64_t)a.nch))[lane] = ge[x];
        ssq += dot4(ge[x], ge[x]);
      }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      if (t.rid[x] != cur) { flush(); cur = t.rid[x]; }
      accR = accR + gr[x];
      accN = accN + gw[x];
    }
  }
  flush();
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * a.nch; i += 256) {
    const float vr = home[0][i];
    if (vr != 0.f) atomicAdd(a.gR + hr * a.ldr + i, vr);
    if (TRANSH) {
      const float vn = home[1][i];
      if (vn != 0.f) atomicAdd(a.gN + hr * a.ldn + i, vn);
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float v = group_sum<64>(part[s]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][s] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (v != 0.f) atomicAdd(a.loss + threadIdx.x, v);
  }
  if (a.sumsq) {                  // one double atomic per workgroup
    ssq = group_sum<64>(ssq);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = ssq;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double t = ((double)red[0][0] + (double)red[0][1]) + ((double)red[0][2] + (double)red[0][3]);
      if (t != 0.0) atomicAdd(a.sumsq + (a.sumsq_slots > 1 ? blockIdx.x % a.sumsq_slots : 0), t);
    }
  }
}

template <bool TRANSH>
int launch_kg_rows(KgRowsArgs& a, hipStream_t st, const char* name) {
#define KTUP_KGR(GL)                                                                                            \
  {                                                                                                             \
    constexpr int GPB = 256 / GL;                                                                               \
    int64_t ch = a.B / ((int64_t)GPB * 512);                                                                    \
    a.chunk = (int)(ch < 1 ? 1 : ch > 16 ? 16 : ch);                                                            \
    const int64_t grid = (a.B + (int64_t)GPB * a.chunk - 1) / ((int64_t)GPB * a.chunk);                         \
    hipLaunchKernelGGL((kg_step_rows_kernel<GL, TRANSH>), dim3((unsigned)grid), dim3(256), 0, st, a);           \
    return check_launch(name);                                                                                  \
  }
  if (a.nch <= 16) KTUP_KGR(16)
  if (a.nch <= 32) KTUP_KGR(32)
  KTUP_KGR(64)
#undef KTUP_KGR
}

}  // namespace

extern "C" int ktup_shard_kg_rel_order(const int64_t* rel, int64_t B, int64_t n_rel, int32_t* order, void* stream) {
  const char* name = "ktup_shard_kg_rel_order";
  KTUP_REQUIRE(rel && order && B > 0 && B < (1ll << 31) && n_rel > 0, "%s: null pointer argument or bad sizes", name);
  hipLaunchKernelGGL(kg_rel_order_kernel, dim3(1), dim3(ORD_T), 0, (hipStream_t)stream, rel, B, n_rel, order);
  return check_launch(name);
}

extern "C" int ktup_train_kg_step_rows(int transh, const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                                       int d, const int64_t* ent_ids, int64_t ent_pad, const int64_t* rel_ids, const int32_t* order, int64_t B, int l1,
                                       float margin, float gscale, int regs, float* loss, float* GE, float* gR, float* gN, double* sumsq,
                                       int n_slots, void* stream) {
  const char* name = "ktup_train_kg_step_rows";
  KTUP_REQUIRE(!sumsq || n_slots >= 1, "%s: the sum of squares needs at least one slot", name);
  KTUP_REQUIRE(B >= 0, "%s: negative batch", name);
  if (B == 0) return KTUP_OK;
  KTUP_REQUIRE(E && R && ent_ids && rel_ids && loss && GE && gR && (!transh || (Nrm && gN)), "%s: null pointer argument", name);
  if (d <= 0 || d % 4 || d > 256 || (lde | ldr | (transh ? ldn : 0)) % 4 || !aligned16(E) || !aligned16(R) || !aligned16(GE) || !aligned16(gR) ||
      (transh && (!aligned16(Nrm) || !aligned16(gN))))
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: needs d %% 4 == 0 (<= 256) and 16-byte aligned rows", name);
  KgRowsArgs a{E, R, Nrm, lde, ldr, ldn, ent_ids, ent_pad, rel_ids, order, B, d / 4, l1 != 0, margin, gscale, regs, 1, loss, GE, gR, gN, sumsq, n_slots};
  return transh ? launch_kg_rows<true>(a, (hipStream_t)stream, name) : launch_kg_rows<false>(a, (hipStream_t)stream, name);
}
