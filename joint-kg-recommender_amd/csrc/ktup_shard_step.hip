// Config 5 (KTUP, d = 256, user / item / entity tables row-sharded by `row % world`): the device side of a training step whose
// every buffer has a FIXED shape, so that the step has no host synchronisation and replays as a HIP graph (one graph on one
// rank; three segments around the two all-to-alls on several).  The reference has no distributed code: the step
// it distributes is knowledgable_recommendation.py:335-344,394-403 (model(pos), model(neg), bprLoss, backward, clip_grad_norm,
// optimizer.step) over jTransUP.py:122-143.
//
//   route   one id list for ALL tables of the step (entries [ent_off[t], ent_off[t+1]) of a block belong to table t; negative
//           ids are padding).  Distinct ids get a WIRE ROW  w = owner * capsum + toff[t] + slot  (owner = id % world, slot = order
//           of first appearance among the (owner, table) pair's ids, fewer than cap[t] of them or the step is flagged as
//           overflowed and skipped by `apply`): the send buffer of the id all-to-all (send_ids[w] = id / world, -1 padded), the
//           row buffer the rows come back into, the compact table the scorer reads, and the gradient buffer that travels back all
//           share this one layout, so nothing is ever re-ordered.  inverse[e] = wire row of entry e (what the scorer is
//           handed as ids); pair_map (KTUP's item -> entity map on the compact tables); and, as a by-product of the same pass
//           over the entries, the counting sort of the entries by wire row that the row-gradient reduction needs
//           (ktup_segreduce.hip) -- its histogram is the rank every entry takes among the entries of its row.
//   pack    owner side: X[w] = table_t[ids[w]] for a whole wire buffer (all tables, one launch).
//   apply   global-norm clip + row-sparse SGD / Adagrad on the rows a step touched, for all tables and the small replicated
//           tables in one launch; consumes (zero-fills) the gradient buffers so the next step starts clean without a memset.
//   bucket  the small tables' gradients + the sum of squares + the overflow flag as ONE fp64 all-reduce bucket.
// The owner side of several ranks combines the rows different peers asked for with the same route + reduction (world = 1,
// block = capsum): a row requested by k peers is k entries of one key.
#include "ktup_pref_geom.h"
#include <cstring>

#include "ktup_lane_swap.h"
#include "ktup_rows.h"

using namespace ktup;

namespace {

constexpr int MAXT = KTUP_SHARD_MAX_TABLES;

struct RouteArgs {
  const int64_t* ids; int64_t n, block; int T; int64_t eoff[MAXT + 1];
  int world; int64_t cap[MAXT], toff[MAXT], capsum, W;
  int pair_a, pair_b;
  unsigned long long* keys; int32_t* slot_pos; uint64_t slots;
  int64_t* inverse; int64_t* send_ids; int32_t* pair_map;
  int32_t *start, *rank, *perm, *skey, *counters;
  double* zero_d; int n_zero_d;
  int32_t* tile_tot; int n_tiles;                      // two-level scan of the histogram (TILE counters per tile); n_tiles = 0: one-workgroup scan
  // KTUP source (ktup_shard_route_ktup): the init launch also BUILDS ids = [u | pos ; neg | item2ent[pos ; neg]] from batch
  // (*cursor mod n_batches) of the id columns, and the NEXT launch moves the cursor on
  const int64_t *src_u, *src_pos, *src_neg; int64_t B, n_batches; const int32_t* item2ent; int64_t ent_pad; int64_t* cursor;
  int64_t* ids_out;
  // KG source (ktup_shard_route_kg): ids = [ph ; pt ; nh ; nt] from the six triple columns (src_u / src_pos / src_neg / src_nt =
  // ph / pt / nh / nt), and the batch's relation ids [pr ; nr] copied to rel_out for the step kernel
  const int64_t *src_nt, *src_pr, *src_nr; int64_t* rel_out;
  bool keep_cursor;                                    // phase 3: the caller moves the cursor itself (ktup_shard_reduce_norm), after every reader of it
};

constexpr int TILE = 1024;                             // histogram counters per scan tile (256 threads x 4)
constexpr int MAX_TILES = 1024;                        // 4 KB of LDS in the scatter launch: it still fits on a CU beside the step kernel's 152 KB

KTUP_DEV uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

KTUP_DEV int table_of(const RouteArgs& a, int64_t b) {
  int t = 0;
#pragma unroll
  for (int i = 1; i < MAXT; ++i) t += (i < a.T && b >= a.eoff[i]) ? 1 : 0;
  return t;
}

// every piece of scratch the step reads before writing: hash keys, the id send buffer, the histogram, the slot counters and the
// caller's accumulators (sum of squares ...) -- a kernel, not memset nodes (a captured memset stopped taking effect)
__global__ __launch_bounds__(256) void route_init_kernel(RouteArgs a) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  for (int64_t i = tid; i < (int64_t)a.slots; i += nth) a.keys[i] = ~0ull;
  for (int64_t i = tid; i < a.W; i += nth) a.send_ids[i] = -1;
  for (int64_t i = tid; i <= a.W; i += nth) a.start[i] = 0;
  for (int64_t i = tid; i <= (int64_t)a.world * a.T; i += nth) a.counters[i] = 0;          // + the overflow word
  for (int64_t i = tid; i < a.n_zero_d; i += nth) a.zero_d[i] = 0.0;
  if (a.src_nt) {                                        // kg step: the triple columns of the cursor's batch
    const int64_t b0 = a.cursor ? ((*a.cursor) % a.n_batches) * a.B : 0;
    for (int64_t k = tid; k < a.B; k += nth) {
      // a corrupted triple keeps its head or its tail (utils/data.py:12-18): that entity is ONE entry -- the twin's slot is padding
      // and ktup_train_kg_step_rows adds the twin's gradient to the positive's stored row
      const int64_t ph = a.src_u[b0 + k], pt = a.src_pos[b0 + k], nh = a.src_neg[b0 + k], nt = a.src_nt[b0 + k];
      a.ids_out[k] = ph;
      a.ids_out[a.B + k] = pt;
      a.ids_out[2 * a.B + k] = nh == ph ? -1 : nh;
      a.ids_out[3 * a.B + k] = nt == pt ? -1 : nt;
      a.rel_out[k] = a.src_pr[b0 + k];
      a.rel_out[a.B + k] = a.src_nr[b0 + k];
    }
  } else if (a.src_u) {                                  // jTransUP.py:122-130: paddingItems as a table lookup
    const int64_t b0 = a.cursor ? ((*a.cursor) % a.n_batches) * a.B : 0;
    for (int64_t k = tid; k < 2 * a.B; k += nth) {              // a user is ONE entry: its positive and its negative pair share the row
      const int64_t kk = k < a.B ? k : k - a.B;
      const int64_t item = k < a.B ? a.src_pos[b0 + kk] : a.src_neg[b0 + kk];
      if (k < a.B) a.ids_out[k] = a.src_u[b0 + k];
      a.ids_out[a.B + k] = item;
      if (a.item2ent) {
        const int64_t ent = a.item2ent[item];
        a.ids_out[3 * a.B + k] = (ent < 0 || ent == a.ent_pad) ? -1 : ent;
      }
    }
  }
}

__global__ __launch_bounds__(256) void route_insert_kernel(RouteArgs a) {
  const int lane = threadIdx.x & 63;
  if (a.cursor && !a.keep_cursor && blockIdx.x == 0 && threadIdx.x == 0) *a.cursor = *a.cursor + 1;     // the init launch has read it
  const uint64_t mask = a.slots - 1;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < a.n; base += (int64_t)gridDim.x * 256) {
    const int64_t e = base + threadIdx.x;
    const int64_t id = e < a.n ? a.ids[e] : -1;
    const bool on = id >= 0;
    const int t = on ? table_of(a, e % a.block) : 0;
    uint64_t h = 0;
    bool created = false;
    if (on) {
      const unsigned long long key = ((unsigned long long)t << 56) | (unsigned long long)id;
      h = mix64(key) & mask;
      for (;;) {
        const unsigned long long prev = atomicCAS(a.keys + h, ~0ull, key);
        if (prev == ~0ull) { created = true; break; }     // this thread names the row
        if (prev == key) break;
        h = (h + 1) & mask;
      }
    }
    const int p = (created && a.world > 1) ? (int)(id % a.world) : 0;
    const int combo = p * a.T + t;
    // slots: ONE counter atomic per (owner, table) present in the wave, all of them in flight together
    int leader = 0, rk = 0, cnt = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int c = 0; c < a.world * a.T; ++c) {
      const unsigned long long m = __ballot(created && combo == c);
      if (m != 0ull && created && combo == c) {
        leader = __ffsll((long long)m) - 1;
        cnt = __popcll(m);
        rk = __popcll(m & lt);
      }
    }
    int32_t basev = 0;
    if (created && lane == leader) basev = atomicAdd(a.counters + combo, cnt);
    basev = __shfl(basev, leader, 64);
    if (created) {
      int64_t slot = basev + rk;
      const bool fits = slot < a.cap[t];
      if (!fits) { atomicAdd(a.counters + (int64_t)a.world * a.T, 1); slot = 0; }
      const int64_t w = (int64_t)p * a.capsum + a.toff[t] + slot;
      a.slot_pos[h] = (int32_t)w;
      if (fits) a.send_ids[w] = a.world > 1 ? id / a.world : id;
    }
  }
}

KTUP_DEV int32_t route_find(const RouteArgs& a, int t, int64_t id) {
  const unsigned long long key = ((unsigned long long)t << 56) | (unsigned long long)id;
  const uint64_t mask = a.slots - 1;
  uint64_t h = mix64(key) & mask;
  while (a.keys[h] != key) h = (h + 1) & mask;
  return a.slot_pos[h];
}

__global__ __launch_bounds__(256) void route_finish_kernel(RouteArgs a) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.n; e += (int64_t)gridDim.x * 256) {
    const int64_t id = a.ids[e];
    if (id < 0) {
      a.inverse[e] = a.W;                                  // the shared zero row
      a.rank[e] = -1;
      continue;
    }
    const int64_t b = e % a.block;
    const int t = table_of(a, b);
    const int32_t w = route_find(a, t, id);
    a.inverse[e] = w;
    a.rank[e] = atomicAdd(a.start + w, 1);
    if (a.pair_map && t == a.pair_a) {                     // KTUP: the compact item row's entity row
      const int64_t idb = a.ids[e + (a.eoff[a.pair_b] - a.eoff[a.pair_a])];
      a.pair_map[w] = idb < 0 ? (int32_t)a.W : route_find(a, a.pair_b, idb);
    }
  }
}

// exclusive scan of the histogram, level 1: every workgroup scans one tile of TILE counters in place and leaves its total
__global__ __launch_bounds__(256) void route_tile_scan_kernel(RouteArgs a) {
  __shared__ int32_t wsum[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * TILE + 4 * t;
  int4 v = make_int4(0, 0, 0, 0);
  if (i0 + 3 < a.W) v = *reinterpret_cast<const int4*>(a.start + i0);
  else {
    if (i0 < a.W) v.x = a.start[i0];
    if (i0 + 1 < a.W) v.y = a.start[i0 + 1];
    if (i0 + 2 < a.W) v.z = a.start[i0 + 2];
  }
  const int32_t mine = (v.x + v.y) + (v.z + v.w);
  int32_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int32_t run = inc - mine;
  for (int k = 0; k < wv; ++k) run += wsum[k];
  const int4 o4 = make_int4(run, run + v.x, run + v.x + v.y, run + v.x + v.y + v.z);
  if (i0 + 3 < a.W) *reinterpret_cast<int4*>(a.start + i0) = o4;
  else {
    if (i0 < a.W) a.start[i0] = o4.x;
    if (i0 + 1 < a.W) a.start[i0 + 1] = o4.y;
    if (i0 + 2 < a.W) a.start[i0 + 2] = o4.z;
  }
  if (t == 255) a.tile_tot[blockIdx.x] = run + mine;
}

// level 2 rides in the scatter launch: every workgroup scans the (few) tile totals into LDS
__global__ __launch_bounds__(256) void route_scatter_kernel(RouteArgs a) {
  __shared__ int32_t tpre[MAX_TILES];
  __shared__ int32_t wsum2[4];
  if (a.n_tiles > 0) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = (a.n_tiles + 255) / 256;
    int32_t mine = 0;
    for (int k = 0; k < per; ++k) { const int i = t * per + k; if (i < a.n_tiles) mine += a.tile_tot[i]; }
    int32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane == 63) wsum2[wv] = inc;
    __syncthreads();
    int32_t run = inc - mine;
    for (int k = 0; k < wv; ++k) run += wsum2[k];
    for (int k = 0; k < per; ++k) { const int i = t * per + k; if (i < a.n_tiles) { tpre[i] = run; run += a.tile_tot[i]; } }
    if (blockIdx.x == 0 && t == 255) a.start[a.W] = run;          // the number of sorted entries (ktup_shard_reduce_rows reads it)
    __syncthreads();
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.n; e += (int64_t)gridDim.x * 256) {
    const int32_t r = a.rank[e];
    if (r < 0) continue;
    const int64_t w = a.inverse[e];
    const int32_t pos = a.start[w] + r + (a.n_tiles > 0 ? tpre[w / TILE] : 0);
    a.perm[pos] = (int32_t)e;
    a.skey[pos] = (int32_t)w;
  }
}

uint64_t route_slots(int64_t n) {
  uint64_t s = 64;
  while (s < (uint64_t)(2 * n)) s <<= 1;
  return s;
}

// ---- [u], [pos ; neg], item2ent[pos ; neg]: the entry list of a KTUP rec step (jTransUP.py:122-130 paddingItems as a table)
__global__ __launch_bounds__(256) void ktup_entries_kernel(const int64_t* __restrict__ u, const int64_t* __restrict__ pi,
                                                           const int64_t* __restrict__ ni, int64_t B, const int32_t* __restrict__ item2ent,
                                                           int64_t ent_pad, int64_t* __restrict__ out) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < 2 * B; k += (int64_t)gridDim.x * 256) {
    const int64_t item = k < B ? pi[k] : ni[k - B];
    if (k < B) out[k] = u[k];
    out[B + k] = item;
    if (item2ent) {
      const int64_t ent = item2ent[item];
      out[3 * B + k] = (ent < 0 || ent == ent_pad) ? -1 : ent;
    }
  }
}

struct WireTables {
  int T; float* tab[MAXT]; int64_t ldt[MAXT]; float* st[MAXT]; int64_t lds[MAXT]; int64_t toff[MAXT + 1]; int64_t capsum;
};

KTUP_DEV int wire_table(const WireTables& w, int64_t row) {
  const int64_t b = row % w.capsum;
  int t = 0;
#pragma unroll
  for (int i = 1; i < MAXT; ++i) t += (i < w.T && b >= w.toff[i]) ? 1 : 0;
  return t;
}

struct PackWire {
  WireTables w; const int64_t* ids; float* out; int64_t ldo;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t id = ids[row];
    if (id < 0) return;                                    // padding slots are never addressed by the scorer
    const int t = wire_table(w, row);
    V x[CPL];
    cx.load(x, w.tab[t] + id * w.ldt[t]);
    V* o = reinterpret_cast<V*>(out + row * ldo);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = cx.lane + j * G;
      if (c < cx.nch) o[c] = x[j];
    }
  }
};

KTUP_DEV void up1(float& p, float& st, float g, float lr, float eps, bool adagrad) {
  if (adagrad) {
    st = fmaf(g, g, st);
    p = p - lr * (g / (sqrtf(st) + eps));
  } else {
    p = fmaf(-lr, g, p);
  }
}
KTUP_DEV void upv(float& p, float& st, float g, float lr, float eps, bool adagrad) { up1(p, st, g, lr, eps, adagrad); }
KTUP_DEV void upv(float4& p, float4& st, float4 g, float lr, float eps, bool adagrad) {
  up1(p.x, st.x, g.x, lr, eps, adagrad); up1(p.y, st.y, g.y, lr, eps, adagrad);
  up1(p.z, st.z, g.z, lr, eps, adagrad); up1(p.w, st.w, g.w, lr, eps, adagrad);
}
KTUP_DEV void vfrom(float& o, const double* p) { o = (float)p[0]; }
KTUP_DEV void vfrom(float4& o, const double* p) { o = make_float4((float)p[0], (float)p[1], (float)p[2], (float)p[3]); }

constexpr int MAXS = KTUP_SHARD_MAX_SMALL;
constexpr int NSLOT = KTUP_SHARD_SUMSQ_SLOTS;

// ---- row-sparse ADAM that reproduces the reference's DENSE Adam (utils/trainer.py:63-66: torch.optim.Adam over every table, l2_lambda = 0)
// to within ~1e-6 per element: the replay below is cut after r.replay steps and starts its bias corrections from __expf(last * log beta)
// (~1e-4 relative on 1 - beta2^last for small `last`; the apply step itself takes the corrections ktup_shard_step_count computed in fp64)
// A dense Adam step moves EVERY row that has ever received a gradient -- a row the batch does not touch still takes
//     m <- beta1 m,  v <- beta2 v,  p <- p - lr / (1 - beta1^s) * m / (sqrt(v) / sqrt(1 - beta2^s) + eps)
// at every step s.  A row's state is kept as [m (d) | v (d) | last (int32) + 3 words of padding] (KTUP_SHARD_ADAM_STATE_PITCH(d) floats);
// `last` = the step its state was written at (0: never touched: m = v = 0 and the row has not moved).  Whoever touches a row at step t
// first REPLAYS the zero-gradient steps last + 1 .. t - 1 in registers -- the dense recurrence, one step after the other, or (eight steps
// or more of a state that is not young: adam_zero_series below) their sum as one short series per row -- then applies step t.  The replayed increments fall like (beta1 / sqrt(beta2))^k, so the replay stops after `replay` steps (the host
// picks it so that what is dropped is below 1e-4 of the first increment, i.e. < 2e-6 absolute at the learning rates in use) and the
// remaining steps only decay m and v (closed form).  ktup_shard_adam_flush brings every row of a shard up to the current step (before an
// evaluation or a checkpoint reads the tables).
struct AdamRule {
  float b1, b2; int replay; const int64_t* step;       // step[0] = number of the step being applied (>= 1; ktup_shard_step_count moves it and
                                                       // leaves that step's bias corrections {1 - beta1^t, sqrt(1 - beta2^t)} as two floats in step[1])
  float ln1, ln2;                                      // log(beta1), log(beta2): beta^k = exp(k log beta) without a pow() per row
  // WEIGHT DECAY (utils/trainer.py:63-77: every torch.optim optimizer is built with weight_decay = l2_lambda, 1e-5 by default): the dense
  // step adds wd * p to the gradient of EVERY row of every table at every step -- also of rows no batch has touched yet.  A row then owes,
  // for each step it was not touched, the optimizer's step on g = wd * p; whoever touches it replays those steps one by one (no
  // closed form for Adagrad / Adam, no geometric tail to cut), from step 1 on (`last` = 0 is a row that has never been written).
  // rule: which optimizer the replay and the step are: 0 Adam, 1 Adagrad (its sum lives in the `v` half of the state row), 2 plain SGD.
  int rule; float wd;
  int istar;                                            // the step the series form of the replay expands around (make_rule)
};
// (a pow() per row and lane -- the bias corrections of the step, the powers the replay starts from -- made the Adam apply walk of config 5
//  193 us against Adagrad's 47: fp64 pow is several hundred instructions)
inline AdamRule make_rule(const ktup_adam_t* a) {
  // the replayed increments fall like rho^i, rho = beta1 / sqrt(beta2): their weight sits around step 1 / (1 - rho)
  const double rho = (double)a->beta1 / sqrt((double)a->beta2);
  const int istar = rho < 1.0 ? (int)fmin(1048576.0, fmax(1.0, floor(1.0 / (1.0 - rho) + 0.5))) : 1;
  return AdamRule{a->beta1, a->beta2, a->replay, a->step, logf(a->beta1), logf(a->beta2), a->rule, a->weight_decay, istar};
}

constexpr int ADAM_SERIES_MIN = 8;      // replays shorter than this stay step by step (the series costs ~a dozen steps of the loop)
// one zero-gradient step on (p, m, sqrt(v)): sqrt(beta2^k v) = sqrt(v) sqrt(beta2)^k, so the replay carries sqrt(v) and multiplies it -- a
// square root per element and step was a quarter of the loop (transcendental rate)
KTUP_DEV void adam_zero_steps(float4& p, float4& m, float4& sv, float c1, float inv_bc2s, float eps, float b1, float sb2) {
#define KTUP_AZ(c)                                                                            \
  m.c = fmaf(-m.c, 1.f - b1, m.c); sv.c = sb2 * sv.c;                                         \
  p.c = fmaf(-c1 * m.c, __builtin_amdgcn_rcpf(fmaf(sv.c, inv_bc2s, eps)), p.c);
  KTUP_AZ(x) KTUP_AZ(y) KTUP_AZ(z) KTUP_AZ(w)
#undef KTUP_AZ
}
// The K zero-gradient steps last + 1 .. last + K at once.  Step i moves an element by
//     lr m0 c_i / (s + e_i),   s = sqrt(v0),   c_i = beta1^i sqrt(bc2_i) / (bc1_i beta2^(i/2)),   e_i = eps sqrt(bc2_i) / beta2^(i/2)
// (bc1_i = 1 - beta1^(last+i), bc2_i = 1 - beta2^(last+i)): c_i and e_i are the same for every element of every row with this (last, K),
// only s and m0 are the element's.  e_i drifts slowly (d ln e / di = -ln(beta2) / (2 bc2) = 5e-4 per step once bc2 ~ 1) while c_i falls like
// 0.9^i, so around a reference E = e_(i*) the sum over i is a short series in U = E / (s + E) in (0, 1]:
//     sum_i c_i / (s + e_i) = 1 / (s + E) * sum_n (-U)^n R_n,   R_n = sum_i c_i (e_i / E - 1)^n
// The lanes of the row's group take the steps i (one each, then the next GL ...) and sum R_0 .. R_4 and the remainder's bound
// R_5 = sum_i c_i |e_i / E - 1|^5 across the group; each element then costs a square root, a reciprocal and a degree-4 Horner form
// instead of K steps of the recurrence (the steady-state catch-up of config 5 replays ~60 steps x 33 k rows: 95 us of the step's 312
// were this loop).  The series is taken only when R_5 <= 1e-6 R_0 (the truncation error relative to the row's whole replayed
// displacement, for every s >= 0); otherwise -- young steps, where bc2 still moves by percents per step -- the caller replays step by step.  ~190 instructions per row at d = 256 (the first version, with expm1f / sqrtf / a division per
// lane and step and shuffles through LDS for the six sums, was 540 and a third of the catch-up kernel's time).
// Sum over the G lanes of a group, in all of them, without LDS: DPP inside a row of 16, the gfx950 lane swaps across rows.
template <int G>
KTUP_DEV float group_sum_dpp(float v) {
#define KTUP_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true));
  KTUP_DPP_ADD(0xB1)      // quad_perm:[1,0,3,2]
  KTUP_DPP_ADD(0x4E)      // quad_perm:[2,3,0,1]
  KTUP_DPP_ADD(0x141)     // row_half_mirror
  KTUP_DPP_ADD(0x140)     // row_mirror
#undef KTUP_DPP_ADD
  if constexpr (G >= 32) v = swap16_sum(v, v);
  if constexpr (G >= 64) v = swap32_sum(v, v);
  return v;
}
// 1 - e^x for x <= 0 without cancellation: the hardware exponential below -1/4, the Taylor polynomial (remainder < 2e-9 relative) above
KTUP_DEV float one_minus_exp(float x) {
  const float t = fmaf(x, fmaf(x, fmaf(x, fmaf(x, fmaf(x, fmaf(x, 1.f / 5040.f, 1.f / 720.f), 1.f / 120.f), 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
  return x > -0.25f ? -x * t : 1.f - __expf(x);
}
template <int GL, int CPL>
KTUP_DEV bool adam_zero_series(float4 (&p)[CPL], const float4 (&m)[CPL], const float4 (&v)[CPL], int last, int K, float lr, float eps,
                               const AdamRule& r, int lane) {
  const float tl = (float)last;
  if (!(eps > 0.f)) return false;
  const float lrho = fmaf(-0.5f, r.ln2, r.ln1);
  const float fs = (float)(K < r.istar ? K : r.istar);
  const float E = eps * __builtin_amdgcn_sqrtf(one_minus_exp((tl + fs) * r.ln2)) * __expf(-0.5f * fs * r.ln2);
  const float iE = __builtin_amdgcn_rcpf(E);
  float R0 = 0.f, R1 = 0.f, R2 = 0.f, R3 = 0.f, R4 = 0.f, R5 = 0.f;
  for (int i = lane + 1; i <= K; i += GL) {
    const float x = (float)i;
    const float bc2s = __builtin_amdgcn_sqrtf(one_minus_exp((tl + x) * r.ln2));
    const float c = __expf(x * lrho) * bc2s * __builtin_amdgcn_rcpf(one_minus_exp((tl + x) * r.ln1));
    const float q = fmaf(eps * bc2s * __expf(-0.5f * x * r.ln2), iE, -1.f), q2 = q * q;
    R0 += c; R1 = fmaf(c, q, R1); R2 = fmaf(c, q2, R2); R3 = fmaf(c * q, q2, R3); R4 = fmaf(c * q2, q2, R4); R5 = fmaf(c * q2, q2 * fabsf(q), R5);
  }
  R0 = group_sum_dpp<GL>(R0); R5 = group_sum_dpp<GL>(R5);
  if (!(R5 <= 1e-6f * R0)) return false;
  R1 = group_sum_dpp<GL>(R1); R2 = group_sum_dpp<GL>(R2); R3 = group_sum_dpp<GL>(R3); R4 = group_sum_dpp<GL>(R4);
#define KTUP_AS(c)                                                                              \
  {                                                                                             \
    const float u = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v[j].c) + E), U = E * u;       \
    const float h = fmaf(-U, fmaf(-U, fmaf(-U, fmaf(-U, R4, R3), R2), R1), R0);                 \
    p[j].c = fmaf(-lr * m[j].c, u * h, p[j].c);                                                 \
  }
#pragma unroll
  for (int j = 0; j < CPL; ++j) { KTUP_AS(x) KTUP_AS(y) KTUP_AS(z) KTUP_AS(w) }
#undef KTUP_AS
  return true;
}
KTUP_DEV void adam_step1(float& p, float& m, float& v, float g, float c1, float bc2s, float eps, float b1, float b2) {
  m = m + (g - m) * (1.f - b1);                      // exp_avg.lerp_(grad, 1 - beta1)          (adam.py _single_tensor_adam, as ktup_optim.hip)
  v = fmaf(1.f - b2, g * g, b2 * v);
  p = p - c1 * (m / (sqrtf(v) / bc2s + eps));
}
// The three dense rules on one element, gradient g (weight decay already added): torch.optim's _single_tensor_* forms
KTUP_DEV void lazy_step1(int rule, float& p, float& m, float& v, float g, float lr, float c1, float bc2s, float eps, float b1, float b2) {
  if (rule == 0) { adam_step1(p, m, v, g, c1, bc2s, eps, b1, b2); return; }
  if (rule == 1) {                                    // adagrad.py: state_sum.addcmul_(g, g); p.addcdiv_(g, sqrt(state_sum) + eps, value=-lr)
    v = fmaf(g, g, v);
    p = p - lr * (g / (sqrtf(v) + eps));
    return;
  }
  p = fmaf(-lr, g, p);                                // sgd.py (momentum 0)
}
// One row's CPL float4 chunks per lane: replay the untouched steps last + 1 .. upto, then (has_g) step t = upto + 1 with gradient g.
// WD (weight decay) is a TEMPLATE parameter: compiled into the same instantiation, its replay (an optimizer step per missed step, fp64
// bias corrections, a third rule) cost the plain Adam walk of config 5 23 registers, 488 bytes of scratch and a wave per SIMD
// (seg_fused_kernel<64,1,1,...>: 115 -> 138 VGPRs; joint_adam 0.395 -> 0.474 ms in the bench line) -- the lesson of the Adagrad / Adam
// split of round 5 once more.
template <int GL, int CPL, bool WD>
KTUP_DEV void adam_row(float4 (&p)[CPL], float4 (&m)[CPL], float4 (&v)[CPL], const float4 (&g)[CPL], bool has_g, int upto, int last,
                       float lr, float eps, const AdamRule& r, int lane) {
  const int miss = upto - last;
  if constexpr (WD) {
    // weight decay: the steps last + 1 .. upto on g = wd * p, exactly as the dense optimizer took them
    if (miss > 0) {
      if (r.rule == 2) {                              // p <- p (1 - lr wd), `miss` times
        const float f = expf((float)miss * log1pf(-lr * r.wd));
#pragma unroll
        for (int j = 0; j < CPL; ++j) p[j] = f * p[j];
      } else {
        double b1p = (double)__expf((float)last * r.ln1), b2p = (double)__expf((float)last * r.ln2);     // (as the plain replay below)
        for (int k = 0; k < miss; ++k) {
          float c1 = lr, bc2s = 1.f;
          if (r.rule == 0) {
            b1p *= (double)r.b1; b2p *= (double)r.b2;
            c1 = lr / (float)(1.0 - b1p); bc2s = (float)sqrt(1.0 - b2p);
          }
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            lazy_step1(r.rule, p[j].x, m[j].x, v[j].x, r.wd * p[j].x, lr, c1, bc2s, eps, r.b1, r.b2);
            lazy_step1(r.rule, p[j].y, m[j].y, v[j].y, r.wd * p[j].y, lr, c1, bc2s, eps, r.b1, r.b2);
            lazy_step1(r.rule, p[j].z, m[j].z, v[j].z, r.wd * p[j].z, lr, c1, bc2s, eps, r.b1, r.b2);
            lazy_step1(r.rule, p[j].w, m[j].w, v[j].w, r.wd * p[j].w, lr, c1, bc2s, eps, r.b1, r.b2);
          }
        }
      }
    }
    if (has_g) {
      const float* bc = reinterpret_cast<const float*>(r.step + 1);
      const float c1 = r.rule == 0 ? lr / bc[0] : lr, bc2s = r.rule == 0 ? bc[1] : 1.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        lazy_step1(r.rule, p[j].x, m[j].x, v[j].x, fmaf(r.wd, p[j].x, g[j].x), lr, c1, bc2s, eps, r.b1, r.b2);
        lazy_step1(r.rule, p[j].y, m[j].y, v[j].y, fmaf(r.wd, p[j].y, g[j].y), lr, c1, bc2s, eps, r.b1, r.b2);
        lazy_step1(r.rule, p[j].z, m[j].z, v[j].z, fmaf(r.wd, p[j].z, g[j].z), lr, c1, bc2s, eps, r.b1, r.b2);
        lazy_step1(r.rule, p[j].w, m[j].w, v[j].w, fmaf(r.wd, p[j].w, g[j].w), lr, c1, bc2s, eps, r.b1, r.b2);
      }
    }
    return;
  } else {
  if (last > 0 && miss > 0) {
    const int K = miss < r.replay ? miss : r.replay;
    // (`last` and K are the row's: the whole lane group takes the same branch)
    if (K >= ADAM_SERIES_MIN && adam_zero_series<GL, CPL>(p, m, v, last, K, lr, eps, r, lane)) {
      const float fk = __expf((float)K * r.ln1);               // m as the K replayed steps leave it
#pragma unroll
      for (int j = 0; j < CPL; ++j) m[j] = fk * m[j];
    } else {
    double b1p = (double)__expf((float)last * r.ln1), b2p = (double)__expf((float)last * r.ln2);
    const float sb2 = sqrtf(r.b2);
    float4 sv[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) sv[j] = make_float4(sqrtf(v[j].x), sqrtf(v[j].y), sqrtf(v[j].z), sqrtf(v[j].w));
    // (not unrolled: with the series above this loop only takes short replays and young states, and unrolled by four it cost the
    //  catch-up kernel 21 registers = a wave per SIMD: steady-state config-5 step 0.238 -> 0.233 ms)
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      b1p *= (double)r.b1; b2p *= (double)r.b2;
      const float c1 = lr * __builtin_amdgcn_rcpf((float)(1.0 - b1p));
      const float ib = __builtin_amdgcn_rsqf((float)(1.0 - b2p));
#pragma unroll
      for (int j = 0; j < CPL; ++j) adam_zero_steps(p[j], m[j], sv[j], c1, ib, eps, r.b1, sb2);
    }
    }
    {   // v after all `miss` steps in closed form (the loop carried its square root); m's remaining decay likewise
      const float f2 = __expf((float)miss * r.ln2);
      const float f1 = miss > K ? __expf((float)(miss - K) * r.ln1) : 1.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) { m[j] = f1 * m[j]; v[j] = f2 * v[j]; }
    }
  }
  if (has_g) {
    const float* bc = reinterpret_cast<const float*>(r.step + 1);          // this step's bias corrections (ktup_shard_step_count)
    const float c1 = lr / bc[0], bc2s = bc[1];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      adam_step1(p[j].x, m[j].x, v[j].x, g[j].x, c1, bc2s, eps, r.b1, r.b2); adam_step1(p[j].y, m[j].y, v[j].y, g[j].y, c1, bc2s, eps, r.b1, r.b2);
      adam_step1(p[j].z, m[j].z, v[j].z, g[j].z, c1, bc2s, eps, r.b1, r.b2); adam_step1(p[j].w, m[j].w, v[j].w, g[j].w, c1, bc2s, eps, r.b1, r.b2);
    }
  }
  }
}
// the same on a row in memory: lane `lane` of a group of GL owns chunks lane, lane + GL, ...; srow = [m | v | last]
template <int GL, int CPL, bool WD>
KTUP_DEV void adam_row_mem(float4* prow, float4* srow, int nch, int lane, const float4 (&g)[CPL], bool has_g, int t, float lr, float eps,
                           const AdamRule& r) {
  static_assert(GL <= 64, "a lane group reads the row's stamp and lane 0 rewrites it without a barrier: the group must lie inside one wave64");
  int32_t* lastp = reinterpret_cast<int32_t*>(srow + 2 * nch);
  const int last = *lastp;
  const int upto = has_g ? t - 1 : t;
  if (!has_g && ((last <= 0 && !WD) || last >= t)) return;          // (weight decay moves rows that were never touched too)
  float4 p[CPL], m[CPL], v[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int ch = lane + j * GL;
    if (ch < nch) { p[j] = prow[ch]; m[j] = srow[ch]; v[j] = srow[nch + ch]; } else { p[j] = f4zero(); m[j] = f4zero(); v[j] = f4zero(); }
  }
  adam_row<GL, CPL, WD>(p, m, v, g, has_g, upto, last, lr, eps, r, lane);
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int ch = lane + j * GL;
    if (ch < nch) { prow[ch] = p[j]; srow[ch] = m[j]; srow[nch + ch] = v[j]; }
  }
  if (lane == 0) *lastp = t;
}

// min(1, max_norm / (||g|| + 1e-6)) with ||g||^2 spread over 1 or NSLOT accumulator words (independent loads, one latency)
KTUP_DEV float clip_coef(float max_norm, const double* __restrict__ sumsq, int slots) {
  if (!(max_norm > 0.f)) return 1.f;
  double tot = sumsq[0];
  if (slots == NSLOT) {
    double v[NSLOT];
#pragma unroll
    for (int k = 1; k < NSLOT; ++k) v[k] = sumsq[k];
#pragma unroll
    for (int k = 1; k < NSLOT; ++k) tot += v[k];
  }
  const float c = max_norm / ((float)sqrt(tot) + 1e-6f);
  return c < 1.f ? c : 1.f;
}

// Same rule as ktup_shard.hip SparseRowStep (utils/trainer.py:63-77 with l2_lambda = 0 restricted to the touched rows), for
// every wire row of every table + the rows of the small replicated tables; the gradient rows are zero-filled once consumed.
// LZ: 0 the plain row-sparse forms (SGD / Adagrad), 1 Adam with catch-up, 2 any rule under weight decay (ktup_adam_t)
template <int LZ>
struct ApplyRowsT {
  static constexpr bool ADAM = LZ != 0;
  WireTables w; const int64_t* ids; int64_t W; float* g; int64_t ldg;
  int n_small, small_rows; float* sg[MAXS]; float* sp0[MAXS]; float* ss0[MAXS]; float* sp1[MAXS]; float* ss1[MAXS];
  const double* small_g64;     // non-null: the all-reduced small gradients (fp64 bucket, entries in sg order) replace sg's values
  int d;
  float lr, eps, max_norm; const double* sumsq; int sumsq_slots; const int32_t* skip_i; const double* skip_d; bool adagrad;
  bool adam; AdamRule ar; int64_t small_lds;       // adam: `adagrad` is set too (= the rows have a state); small_lds: pitch of the small tables' states
  const int32_t* xkeys;        // non-null: rows [0, W) are LIST entries -- the wire row is xkeys[row], taken at its first occurrence only
  // end of step (may be null): loss_sum[k] += loss_step[k] unless the step is skipped, loss_step := 0; *skipped += 1 if it is
  float* loss_step; int n_loss; float* loss_sum; int32_t* skipped;

  template <int G, int CPL>
  KTUP_DEV void one_adam(const RowCtx<float4, G, CPL>& cx, float* prow, float* srow, const float4 (&gr)[CPL], float coef) const {
    float4 g[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) g[j] = coef * gr[j];
    adam_row_mem<G, CPL, LZ == 2>(reinterpret_cast<float4*>(prow), reinterpret_cast<float4*>(srow), cx.nch, cx.lane, g, true, (int)*ar.step, lr, eps, ar);
  }
  template <int G, int CPL>
  KTUP_DEV void one_adam(const RowCtx<float, G, CPL>&, float*, float*, const float (&)[CPL], float) const {}   // (refused on the host: Adam rows are float4)

  template <typename V, int G, int CPL>
  KTUP_DEV void one(const RowCtx<V, G, CPL>& cx, float* prow, float* srow, const V (&gr)[CPL], float coef) const {
    // (ADAM is a template parameter: with the replay loop and its fp64 powers compiled into the Adagrad / SGD instantiation the apply
    //  walk of config 5 took 169 us instead of 47 -- registers)
    if constexpr (ADAM) { one_adam<G, CPL>(cx, prow, srow, gr, coef); return; }
    V p[CPL], st[CPL];
    cx.load(p, prow);
    if (adagrad) cx.load(st, srow);
    V* po = reinterpret_cast<V*>(prow);
    V* so = reinterpret_cast<V*>(srow);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = cx.lane + j * G;
      if (c < cx.nch) {
        upv(p[j], st[j], vscale(coef, gr[j]), lr, eps, adagrad);
        po[c] = p[j];
        if (adagrad) so[c] = st[j];
      }
    }
  }

  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const bool skip = (skip_i && *skip_i != 0) || (skip_d && *skip_d != 0.0);
    if (row == 0 && cx.lane == 0) {                      // one lane of the launch closes the step's books
      if (skip && skipped) *skipped = *skipped + 1;
      for (int k = 0; k < n_loss; ++k) {
        const float v = loss_step[k];
        if (!skip) loss_sum[k] += v;
        loss_step[k] = 0.f;
      }
    }
    const float coef = clip_coef(max_norm, sumsq, sumsq_slots);
    V gr[CPL], zero[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) vzero(zero[j]);
    if (row < W) {
      if (xkeys) {
        const int32_t key = xkeys[row];
        if (key < 0 || (row > 0 && xkeys[row - 1] == key)) return;
        row = key;
      }
      const int64_t id = ids[row];
      if (id < 0) return;
      float* grow = g + row * ldg;
      if (!skip) {
        const int t = wire_table(w, row);
        cx.load(gr, grow);
        one(cx, w.tab[t] + id * w.ldt[t], adagrad ? w.st[t] + id * w.lds[t] : nullptr, gr, coef);
      }
      V* go = reinterpret_cast<V*>(grow);
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int c = cx.lane + j * G;
        if (c < cx.nch) go[c] = zero[j];
      }
      return;
    }
    const int64_t r = row - W;
    const int k = (int)(r / small_rows);
    const int64_t sr = r % small_rows;
    float* grow = sg[k] + sr * d;
    if (!skip) {
      if (small_g64) {
        const double* src = small_g64 + ((int64_t)k * small_rows + sr) * d;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const int c = cx.lane + j * G;
          if (c < cx.nch) vfrom(gr[j], src + (int64_t)c * VW<V>::W); else vzero(gr[j]);
        }
      } else {
        cx.load(gr, grow);
      }
      one(cx, sp0[k] + sr * d, adagrad ? ss0[k] + sr * small_lds : nullptr, gr, coef);
      if (sp1[k]) one(cx, sp1[k] + sr * d, adagrad ? ss1[k] + sr * small_lds : nullptr, gr, coef);
    }
    V* go = reinterpret_cast<V*>(grow);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = cx.lane + j * G;
      if (c < cx.nch) go[c] = zero[j];
    }
  }
};
using ApplyRows = ApplyRowsT<0>;


// ---- reduction by sorted segments WITHOUT a gradient buffer (one rank's reduce -> norm -> apply, and the owner side of several):
// the same chunk walk as ktup_segreduce.hip's seg_reduce_kernel over the route's sorted order, run twice.  A row whose entries
// all lie inside one workgroup ("interior": all but ~2 rows per workgroup) is finished in registers:
//   MODE 0 (norm)   its squared norm joins the step's sum of squares; nothing is written
//   MODE 1 (apply)  the clipped row-sparse SGD / Adagrad update is applied to its table row straight from the registers
// Rows that continue into a neighbouring workgroup ("boundary") are summed by float atomics into gw (zero before, zero after) by
// MODE 0, which also lists their keys (xkeys[2 wg], [2 wg + 1] = the head / tail boundary row of workgroup wg or -1; equal keys
// are adjacent); two small launches (xnorm_kernel, ApplyRows with xkeys) then treat each listed row once.  Against reduce + norm
// + apply through a W x d gradient buffer this moves 2 x |G| + (p, state) instead of |G| + 5 W d + (p, state) bytes.
struct FusedArgs {
  const float4* G; int64_t ldg4; int nch; int64_t n_src, src_off;
  const int32_t *perm, *skey; const int32_t* m_dev; int chunk;
  float* gw; int64_t ldw; int32_t* xkeys;
  double* sumsq; int slots;                         // MODE 0: accumulated; MODE 1: read
  bool dup_only;                                    // MODE 0: the step kernel has already added |G row|^2 for every entry: add only what rows SHARED
                                                    // by several entries change, |sum|^2 - sum |.|^2 -- an entry alone on its row is not even read
  WireTables w; const int64_t* ids; float lr, eps, max_norm; bool adagrad; const int32_t* skip_i; const double* skip_d;
  bool adam; AdamRule ar;
};

// What the norm walk's EXTRA workgroups add (blocks [walk_grid, gridDim.x) of its launch): the small replicated gradients' squares,
// accumulators another graph branch filled, and the batch cursor's move.  (Round 3 ran a second launch for this and for the boundary
// rows; those now account for themselves: see boundary() below.)
struct XNormArgs {
  int n_small; const float* sg[MAXS]; int64_t small_elems; float small_weight;
  double* sumsq; int slots;         // (sumsq null: nothing is added -- the fold below alone, as extra workgroups of the store walk)
  // the step kernel's REPLICAS of the preference-table gradients (ktup_train_rec_step_rows_ws): rep[r][A | C][rep_elems]; element j of A is
  // added to rdst[0][j] (and rdst[2][j] if not null), of C to rdst[1][j] (rdst[3][j]); the replicas are left zero.  One thread owns an
  // element: no atomics.  The squares added to sumsq are then those of the FOLDED values (rdst replaces sg as the list of small gradients).
  float* rep; int n_rep; int64_t rep_elems; float* rdst[4];
  double* fold; int n_fold;          // accumulators another stream filled while `sumsq` was being cleared: added in, left zero
  int64_t* cursor;                   // moved on here: every reader of the step's batch position is done
};

KTUP_DEV void xnorm_blocks(const XNormArgs& a, int blk, int nblk) {
  if (blk == 0 && threadIdx.x == 0) {
    if (a.fold) {
      double t = 0.0;
      for (int k = 0; k < a.n_fold; ++k) { t += a.fold[k]; a.fold[k] = 0.0; }
      if (t != 0.0) atomicAdd(a.sumsq, t);
    }
    if (a.cursor) *a.cursor = *a.cursor + 1;
  }
  float ss = 0.f;
  if (a.rep) {
    for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < 2 * a.rep_elems; i += (int64_t)nblk * 256) {
      const int which = i >= a.rep_elems ? 1 : 0;
      const int64_t j = i - which * a.rep_elems;
      float* r0 = a.rep + which * a.rep_elems + j;
      float t = 0.f;
      for (int r = 0; r < a.n_rep; ++r) { t += r0[(int64_t)r * 2 * a.rep_elems]; r0[(int64_t)r * 2 * a.rep_elems] = 0.f; }     // (independent loads: one round trip)
      const float v1 = a.rdst[which][j] + t;
      a.rdst[which][j] = v1;
      ss = fmaf(a.small_weight * v1, v1, ss);
      if (a.rdst[2 + which]) {
        const float v2 = a.rdst[2 + which][j] + t;
        a.rdst[2 + which][j] = v2;
        ss = fmaf(a.small_weight * v2, v2, ss);
      }
    }
  } else {
    const int64_t N = (int64_t)a.n_small * a.small_elems;
    for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < N; i += (int64_t)nblk * 256) {
      const float v = a.sg[i / a.small_elems][i % a.small_elems];
      ss = fmaf(a.small_weight * v, v, ss);
    }
  }
  if (!a.sumsq) return;
  __shared__ float xred[4];
  ss = group_sum<64>(ss);
  if ((threadIdx.x & 63) == 0) xred[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = ((double)xred[0] + (double)xred[1]) + ((double)xred[2] + (double)xred[3]);
    if (t != 0.0) atomicAdd(a.sumsq + (a.slots > 1 ? blk % a.slots : 0), t);
  }
}

// MODE 1 carries the step's LAST launch along as extra workgroups [walk_grid, gridDim.x): the listed boundary rows (from gw, then
// zero-filled), the small replicated tables and the step's bookkeeping (ApplyRows over op_rows rows) depend on the norm only, not on
// this walk -- as a launch of their own they were 6 us of dependent latencies at the very end of every step.
template <int GL, int CPL, int MODE, int LZ>
__global__ __launch_bounds__(256) void seg_fused_kernel(FusedArgs a, ApplyRowsT<LZ> op, int64_t op_rows, int walk_grid, XNormArgs xn) {
  const int lane = threadIdx.x % GL, grp = threadIdx.x / GL;
  constexpr int GPB = 256 / GL;
  if ((MODE == 0 || MODE == 2) && (int)blockIdx.x >= walk_grid) {
    xnorm_blocks(xn, (int)blockIdx.x - walk_grid, (int)gridDim.x - walk_grid);
    return;
  }
  if (MODE == 1 && (int)blockIdx.x >= walk_grid) {
    const RowCtx<float4, GL, CPL> cx{a.nch, lane};
    for (int64_t row = (int64_t)((int)blockIdx.x - walk_grid) * GPB + grp; row < op_rows; row += (int64_t)((int)gridDim.x - walk_grid) * GPB)
      op.template run<float4, GL, CPL>(cx, row);
    return;
  }
  constexpr int ROW4 = GL * CPL;
  __shared__ float4 edge[2 * GPB * ROW4];
  __shared__ int32_t ekey[2 * GPB];
  const int64_t m = *a.m_dev;
  const int64_t nchunks = (m + a.chunk - 1) / a.chunk;
  const int64_t c0 = (int64_t)blockIdx.x * GPB;
  if (MODE == 0 && threadIdx.x < 2) a.xkeys[2 * (int64_t)blockIdx.x + threadIdx.x] = -1;
  float ss = 0.f;
  float coef = 1.f;
  bool skip = false;
  if (MODE == 1) {
    coef = clip_coef(a.max_norm, a.sumsq, a.slots);
    skip = (a.skip_i && *a.skip_i != 0) || (a.skip_d && *a.skip_d != 0.0);
  }
  if (c0 >= nchunks) return;
  auto interior = [&](int32_t key, const float4* acc) {
    if (MODE == 2) {                                                     // the reduced row itself, stored (no read, no zero-fill before)
      float4* row = reinterpret_cast<float4*>(a.gw + (int64_t)key * a.ldw);
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int ch = lane + j * GL;
        if (ch < a.nch) row[ch] = acc[j];
      }
      return;
    }
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < CPL; ++j) ss += dot4(acc[j], acc[j]);            // chunks past the row hold zeros
      return;
    }
    if (skip) return;
    const int64_t id = a.ids[key];
    if (id < 0) return;
    const int t = wire_table(a.w, key);
    float4* prow = reinterpret_cast<float4*>(a.w.tab[t] + id * a.w.ldt[t]);
    float4* srow = a.adagrad ? reinterpret_cast<float4*>(a.w.st[t] + id * a.w.lds[t]) : nullptr;
    if constexpr (LZ != 0) {
      float4 g[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) g[j] = coef * acc[j];
      adam_row_mem<GL, CPL, LZ == 2>(prow, srow, a.nch, lane, g, true, (int)*a.ar.step, a.lr, a.eps, a.ar);
      return;
    }
    float4 p[CPL], st[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int ch = lane + j * GL;
      if (ch < a.nch) { p[j] = prow[ch]; if (a.adagrad) st[j] = srow[ch]; }
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int ch = lane + j * GL;
      if (ch < a.nch) {
        upv(p[j], st[j], coef * acc[j], a.lr, a.eps, a.adagrad);
        prow[ch] = p[j];
        if (a.adagrad) srow[ch] = st[j];
      }
    }
  };
  auto boundary = [&](int32_t key, const float4* acc, bool head, bool tail) {
    if (MODE == 2) {                                                     // several workgroups share the row: it was zero-filled beforehand
      float* row = a.gw + (int64_t)key * a.ldw;                          // (ktup_shard_zero_shared_rows: every row with two or more entries)
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int ch = lane + j * GL;
        if (ch < a.nch) atomic_add4(row + 4 * ch, acc[j]);
      }
      return;
    }
    if (MODE != 0) return;                                               // MODE 0 put it into gw; the apply launch's extra workgroups apply it
    // A boundary row is the sum of several workgroups' partials, added to gw by float atomics in any order.  Its squared norm needs no
    // pass of its own: an add of v onto `old` raises the row's square by (old + v)^2 - old^2 = 2 old v + v^2, and the atomic RETURNS old --
    // summed over all the adds that telescopes to |final row|^2 whatever the order
    float* row = a.gw + (int64_t)key * a.ldw;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int ch = lane + j * GL;
      if (ch < a.nch) {
        float* p = row + 4 * ch;
        const float4 v = acc[j];
        const float ox = atomicAdd(p + 0, v.x), oy = atomicAdd(p + 1, v.y), oz = atomicAdd(p + 2, v.z), ow = atomicAdd(p + 3, v.w);
        ss += fmaf(2.f * ox, v.x, v.x * v.x) + fmaf(2.f * oy, v.y, v.y * v.y) + fmaf(2.f * oz, v.z, v.z * v.z) + fmaf(2.f * ow, v.w, v.w * v.w);
      }
    }
    // a workgroup that lies wholly inside one hot row lists it on BOTH sides: equal keys must stay adjacent in the list
    if (lane == 0 && head) a.xkeys[2 * (int64_t)blockIdx.x] = key;
    if (lane == 0 && tail) a.xkeys[2 * (int64_t)blockIdx.x + 1] = key;
  };
  {
    const int64_t c = c0 + grp;
    const bool active = c < nchunks;
    if (lane == 0) { ekey[2 * grp] = -1; ekey[2 * grp + 1] = -1; }
    if (active) {
      const int64_t k0 = c * a.chunk, k1 = min(m, k0 + a.chunk);
      float4 acc[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) acc[j] = f4zero();
      int32_t cur = a.skey[k0];
      const int32_t before = k0 > 0 ? a.skey[k0 - 1] : -1, after = k1 < m ? a.skey[k1] : -1;
      bool head_open = true;
      auto flush = [&](int32_t key, bool last) {
        const bool is_head = head_open && key == before, is_tail = last && key == after;
        if (is_head || is_tail) {
          const int slot = is_head ? 2 * grp : 2 * grp + 1;
          if (lane == 0) ekey[slot] = key;
#pragma unroll
          for (int j = 0; j < CPL; ++j) edge[slot * ROW4 + lane + j * GL] = acc[j];
        } else {
          interior(key, acc);
        }
        head_open = false;
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] = f4zero();
      };
      constexpr int UNR = CPL == 1 ? 8 : 4;
      for (int64_t k = k0; k < k1; k += UNR) {
        int32_t key[UNR], e[UNR];
        float4 v[UNR][CPL];
        int32_t kprev = (MODE == 0 && a.dup_only && k > 0) ? a.skey[k - 1] : -1;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          bool on = k + u < k1;
          key[u] = on ? a.skey[k + u] : -1;
          if (MODE == 0 && a.dup_only) {                                 // (the neighbours may lie in another chunk: only the keys matter)
            const int32_t knext = (on && k + u + 1 < m) ? a.skey[k + u + 1] : -1;
            on = on && (key[u] == kprev || key[u] == knext);
            kprev = key[u];
          }
          e[u] = on ? a.perm[k + u] : 0;
          const int64_t src = e[u] >= a.n_src ? e[u] - a.src_off : e[u];
          const float4* row = a.G + src * a.ldg4;
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            const int ch = lane + j * GL;
            v[u][j] = (on && ch < a.nch) ? row[ch] : f4zero();
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (key[u] < 0) break;
          if (key[u] != cur) { flush(cur, false); cur = key[u]; }
#pragma unroll
          for (int j = 0; j < CPL; ++j) acc[j] = acc[j] + v[u][j];
          if (MODE == 0 && a.dup_only) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) ss -= dot4(v[u][j], v[u][j]);
          }
        }
      }
      flush(cur, true);
    }
    __syncthreads();
    if (grp == 0) {                                               // join the edge partials of this workgroup's consecutive chunks
      const int64_t w0 = c0 * a.chunk, w1 = min(m, (c0 + GPB) * a.chunk);
      const int32_t wbefore = w0 > 0 ? a.skey[w0 - 1] : -1, wafter = w1 < m ? a.skey[w1] : -1;
      float4 acc[CPL];
      int32_t cur = -1;
      auto done = [&](int32_t key) {
        if (key == wbefore || key == wafter) boundary(key, acc, key == wbefore, key == wafter);
        else interior(key, acc);
      };
      for (int sl = 0; sl < 2 * GPB; ++sl) {
        const int32_t key = ekey[sl];
        if (key < 0) continue;
        if (key != cur) {
          if (cur >= 0) done(cur);
          cur = key;
#pragma unroll
          for (int j = 0; j < CPL; ++j) acc[j] = f4zero();
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] = acc[j] + edge[sl * ROW4 + lane + j * GL];
      }
      if (cur >= 0) done(cur);
    }
  }
  if (MODE == 0) {
    __shared__ float red[4];
    ss = group_sum<64>(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double t = ((double)red[0] + (double)red[1]) + ((double)red[2] + (double)red[3]);
      if (t != 0.0) atomicAdd(a.sumsq + (a.slots > 1 ? blockIdx.x % a.slots : 0), t);
    }
  }
}

// Sorted entries per lane group.  Measured at config 5 (49,152 entries, d = 256): 8 -> 0.171 ms per step, 16 -> 0.173, 24 -> 0.189,
// 32 -> 0.205, 64 -> 0.275: the two walks live on memory-level parallelism (many lane groups with a few rows each), so the
// chunk stays small until the batch is large enough to fill the chip anyway.
int fused_chunk(int64_t m_max) {
  if (opt_shard_chunk() > 0) return (opt_shard_chunk() + 3) & ~3;
  int64_t ch = m_max / 8192;
  int c = (int)(ch < 8 ? 8 : ch > 64 ? 64 : ch);
  return (c + 3) & ~3;
}
int fused_gl(int d) { const int nch = d / 4; return nch <= 16 ? 16 : nch <= 32 ? 32 : 64; }
int64_t fused_grid(int64_t m_max, int d) {
  const int chunk = fused_chunk(m_max);
  const int64_t nchunks = (m_max + chunk - 1) / chunk;
  const int gpb = 256 / fused_gl(d);
  return (nchunks + gpb - 1) / gpb;
}

template <int MODE, int LZ = 0>
int launch_fused(const FusedArgs& a, int64_t grid, hipStream_t st, const char* name, const ApplyRowsT<LZ>* op = nullptr, int64_t op_rows = 0,
                 const XNormArgs* xn = nullptr) {
  const ApplyRowsT<LZ> none{};
  const XNormArgs xnone{};
#define KTUP_F(GL, CPL)                                                                                  \
  {                                                                                                      \
    int64_t extra = op ? grid_for((op_rows + (256 / GL) - 1) / (256 / GL), 256) : 0;                     \
    if (xn) extra = xn->rep ? grid_for((2 * xn->rep_elems + 255) / 256, 256)                             \
                            : grid_for(((int64_t)xn->n_small * xn->small_elems + 2047) / 2048, 64);         \
    hipLaunchKernelGGL((seg_fused_kernel<GL, CPL, MODE, LZ>), dim3((unsigned)(grid + extra)), dim3(256), 0, st, a, op ? *op : none, op_rows, \
                       (int)grid, xn ? *xn : xnone);                                                     \
    return check_launch(name);                                                                           \
  }
  if (a.nch <= 16) KTUP_F(16, 1)
  if (a.nch <= 32) KTUP_F(32, 1)
  if (a.nch <= 64) KTUP_F(64, 1)
  if (a.nch <= 128) KTUP_F(64, 2)
  KTUP_F(64, 4)
#undef KTUP_F
}

int fill_fused(const char* name, FusedArgs& a, const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
               int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, int32_t* xkeys) {
  KTUP_REQUIRE(G && sort_ws && gwire && xkeys && n_entries > 0 && n_wire_rows > 0 && d > 0, "%s: null pointer argument or bad sizes", name);
  KTUP_REQUIRE(n_src > 0 && n_src <= n_entries && src_off >= 0 && src_off <= n_src, "%s: bad source layout", name);
  if (d % 4 || d > 1024 || ldg % 4 || ldw % 4 || !aligned16(G) || !aligned16(gwire) || n_entries >= (1ll << 31))
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: needs d %% 4 == 0 (<= 1024) and 16-byte aligned rows", name);
  a.G = reinterpret_cast<const float4*>(G); a.ldg4 = ldg / 4; a.nch = d / 4; a.n_src = n_src; a.src_off = src_off;
  a.perm = sort_ws + ((n_wire_rows + 2) & ~(int64_t)1) + n_entries;
  a.skey = a.perm + n_entries;
  a.m_dev = sort_ws + n_wire_rows;
  a.chunk = fused_chunk(n_entries);
  a.gw = gwire; a.ldw = ldw; a.xkeys = xkeys;
  return KTUP_OK;
}

struct BucketArgs {
  int n_small; const float* sg[MAXS]; int64_t small_elems;     // elements per small gradient (rows x d, contiguous)
  double* bucket; const double* sumsq_in; int sumsq_slots; const int32_t* overflow; double* sumsq_out; double small_weight;
};

// mode 0: bucket = [small gradients as doubles | local sum of squares | overflow count]  (what ONE all-reduce carries)
// mode 1: *sumsq_out = bucket[sum of squares] + sum of the (now global) small gradients' squares -- identical on every rank
template <int MODE>
__global__ __launch_bounds__(1024) void bucket_kernel(BucketArgs a) {
  const int64_t N = (int64_t)a.n_small * a.small_elems;
  if (MODE == 0) {
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < N + 2; i += (int64_t)gridDim.x * 1024) {
      double v;
      if (i < N) v = (double)a.sg[i / a.small_elems][i % a.small_elems];
      else if (i == N) { v = 0.0; for (int k = 0; k < a.sumsq_slots; ++k) v += a.sumsq_in[k]; }
      else v = (double)*a.overflow;
      a.bucket[i] = v;
    }
  } else {
    __shared__ double red[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 1024) acc += a.bucket[i] * a.bucket[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int k = 0; k < 16; ++k) tot += red[k];
      *a.sumsq_out = a.bucket[N] + a.small_weight * tot;     // a gradient shared by two tables counts twice in the norm
    }
  }
}

__global__ __launch_bounds__(256) void zero_f4_kernel(float4* __restrict__ p, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) p[i] = f4zero();
}


// Rows of the wire-layout gradient buffer that SEVERAL entries share, zero-filled: the second entry of a row (rank 1 in its row: there is
// exactly one per shared row) names it.  With that, ktup_shard_reduce_store needs no zero-filled buffer: a row with one entry or with all
// its entries inside one workgroup is STORED, and only rows cut by a workgroup's edge are summed by atomics -- onto these zeros.
__global__ __launch_bounds__(256) void zero_shared_rows_kernel(const int32_t* __restrict__ rank, const int64_t* __restrict__ inverse, int64_t n,
                                                               float* __restrict__ gw, int64_t ldw, int nch) {
  const int lane = threadIdx.x & 63;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; base < n; base += (int64_t)gridDim.x * 256) {
    const int64_t e = base + lane;
    const bool second = e < n && rank[e] == 1;
    const int64_t w = second ? inverse[e] : 0;
    unsigned long long m = __ballot(second);
    while (m != 0ull) {                                             // the wave clears one listed row per round, a float4 per lane
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1ull;
      const int64_t row = __shfl(w, src, 64);
      float4* dst = reinterpret_cast<float4*>(gw + row * ldw);
      for (int ch = lane; ch < nch; ch += 64) dst[ch] = f4zero();
    }
  }
}

int fill_route(const char* name, RouteArgs& a, const int64_t* ids, int64_t n_entries, int64_t block, int n_tables, const int64_t* ent_off,
               int world, const int64_t* cap) {
  KTUP_REQUIRE(n_entries > 0 && n_entries < (1ll << 30) && block > 0 && n_entries % block == 0, "%s: bad entry count / block", name);
  KTUP_REQUIRE(n_tables >= 1 && n_tables <= MAXT && world >= 1 && world <= 64, "%s: 1..%d tables, 1..64 ranks", name, MAXT);
  KTUP_REQUIRE(ids && ent_off && cap, "%s: null pointer argument", name);
  KTUP_REQUIRE(ent_off[0] == 0 && ent_off[n_tables] == block, "%s: ent_off must run from 0 to block", name);
  a.ids = ids; a.n = n_entries; a.block = block; a.T = n_tables; a.world = world;
  int64_t off = 0;
  for (int t = 0; t < n_tables; ++t) {
    KTUP_REQUIRE(ent_off[t + 1] >= ent_off[t] && cap[t] > 0, "%s: table %d: bad entry range or capacity", name, t);
    a.eoff[t] = ent_off[t]; a.cap[t] = cap[t]; a.toff[t] = off;
    off += cap[t];
  }
  for (int t = n_tables; t <= MAXT; ++t) a.eoff[t] = block;
  a.capsum = off; a.W = off * world;
  KTUP_REQUIRE(a.W < (1ll << 31) - 1, "%s: wire buffer too large", name);
  return KTUP_OK;
}

int fill_wire(const char* name, WireTables& w, int n_tables, float* const* tables, const int64_t* ld, float* const* states,
              const int64_t* lds, const int64_t* cap, int d) {
  KTUP_REQUIRE(n_tables >= 1 && n_tables <= MAXT && tables && ld && cap, "%s: 1..%d tables with pitches and capacities", name, MAXT);
  w.T = n_tables;
  int64_t off = 0;
  for (int t = 0; t < n_tables; ++t) {
    KTUP_REQUIRE(tables[t] && ld[t] >= d && cap[t] > 0, "%s: table %d: null pointer, pitch < d or no capacity", name, t);
    w.tab[t] = tables[t]; w.ldt[t] = ld[t];
    w.st[t] = states ? states[t] : nullptr; w.lds[t] = (states && lds) ? lds[t] : ld[t];
    w.toff[t] = off;
    off += cap[t];
  }
  for (int t = n_tables; t <= MAXT; ++t) w.toff[t] = off;
  w.capsum = off;
  return KTUP_OK;
}

int check_kind(const char* name, int kind, const ktup_adam_t* adam, int d) {
  KTUP_REQUIRE(kind == KTUP_OPT_SGD || kind == KTUP_OPT_ADAGRAD || kind == KTUP_OPT_ADAM,
               "%s: row-sparse steps exist for plain SGD, Adagrad and Adam (with catch-up of the untouched steps)", name);
  if (kind == KTUP_OPT_ADAM) {
    KTUP_REQUIRE(adam && adam->step, "%s: Adam needs its rule (betas, replay length, the device step counter)", name);
    KTUP_REQUIRE(adam->beta1 >= 0.f && adam->beta1 < 1.f && adam->beta2 > 0.f && adam->beta2 < 1.f && adam->replay >= 0, "%s: bad Adam rule", name);
    KTUP_REQUIRE(adam->rule >= 0 && adam->rule <= 2 && adam->weight_decay >= 0.f && (adam->rule == 0 || adam->weight_decay > 0.f),
                 "%s: rule is 0 (Adam), or 1 (Adagrad) / 2 (SGD) with a weight decay > 0 (without one those two have plain row-sparse forms)", name);
    KTUP_REQUIRE(d % 4 == 0, "%s: Adam rows need d %% 4 == 0", name);
  }
  return KTUP_OK;
}

// *step += 1 unless the step is skipped (the same two flags the apply launch reads): the launch BEFORE the apply launch of a step
__global__ void step_count_kernel(int64_t* step, const int32_t* skip_i, const double* skip_d, float b1, float b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const bool skip = (skip_i && *skip_i != 0) || (skip_d && *skip_d != 0.0);
    if (!skip) {
      const int64_t t = *step + 1;
      *step = t;
      float* bc = reinterpret_cast<float*>(step + 1);                       // adam.py: bias_correction1, sqrt(bias_correction2), as ktup_optim.hip
      bc[0] = (float)(1.0 - pow((double)b1, (double)t));
      bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
    }
  }
}

// Rows of up to KTUP_SHARD_ADAM_MAX_SEG tables brought up to step *r.step (zero-gradient steps replayed): segment k = rows ids[k][0 .. n[k])
// of table k (negative ids: padding), or rows 0 .. n[k] - 1 when ids[k] is null.  The ids of a segment are DISTINCT.
struct CatchupArgs {
  int n_seg; float* tab[KTUP_SHARD_ADAM_MAX_SEG]; int64_t ldt[KTUP_SHARD_ADAM_MAX_SEG]; float* st[KTUP_SHARD_ADAM_MAX_SEG];
  int64_t lds[KTUP_SHARD_ADAM_MAX_SEG]; const int64_t* ids[KTUP_SHARD_ADAM_MAX_SEG]; int64_t end[KTUP_SHARD_ADAM_MAX_SEG];   // end: prefix sums of n
  int nch; float lr, eps; AdamRule r;
};

template <int GL, int CPL, bool WD>
__global__ __launch_bounds__(256) void adam_catchup_kernel(CatchupArgs a) {
  // A lane group per row, RB rows per round: their ids, then their `last` stamps, are RB independent loads in flight per group -- every
  // state row lies on another page, so a stamp is a TLB miss, and one row after the other the misses of a group queued up behind each
  // other.  All lanes of a group read the SAME address (one translation per row).  One LANE per row -- 64 different pages per load
  // instruction -- was measured too: 209 us against 76 for the 40,960 stamps of a config-5 step; the translations of one instruction
  // are served one after the other.
  constexpr int GPB = 256 / GL, RB = 4;
  const int lane = threadIdx.x % GL;
  const int t = (int)*a.r.step;
  float4 none[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) none[j] = f4zero();
  const int64_t total = a.end[a.n_seg - 1];
  const int64_t ngrp = (int64_t)gridDim.x * GPB;
  for (int64_t e0 = (int64_t)blockIdx.x * GPB + threadIdx.x / GL; e0 < total; e0 += ngrp * RB) {
    int k[RB];
    int64_t row[RB];
    int last[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int64_t e = e0 + u * ngrp;
      k[u] = 0;
#pragma unroll
      for (int i = 0; i < KTUP_SHARD_ADAM_MAX_SEG - 1; ++i) k[u] += (i < a.n_seg - 1 && e >= a.end[i]) ? 1 : 0;
      row[u] = -1;
      if (e < total) {
        const int64_t pos = e - (k[u] > 0 ? a.end[k[u] - 1] : 0);
        row[u] = a.ids[k[u]] ? a.ids[k[u]][pos] : pos;
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u)
      last[u] = row[u] >= 0 ? *reinterpret_cast<const int32_t*>(a.st[k[u]] + row[u] * a.lds[k[u]] + 8 * a.nch) : 0;
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      if (row[u] >= 0 && (last[u] > 0 || WD) && last[u] < t)
        adam_row_mem<GL, CPL, WD>(reinterpret_cast<float4*>(a.tab[k[u]] + row[u] * a.ldt[k[u]]), reinterpret_cast<float4*>(a.st[k[u]] + row[u] * a.lds[k[u]]),
                              a.nch, lane, none, false, t, a.lr, a.eps, a.r);
    }
  }
}

}  // namespace

extern "C" int ktup_shard_step_count(int64_t* step, const int32_t* skip_count, const double* skip_value, float beta1, float beta2, void* stream) {
  KTUP_REQUIRE(step, "ktup_shard_step_count: null counter");
  hipLaunchKernelGGL(step_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, skip_count, skip_value, beta1, beta2);
  return check_launch("ktup_shard_step_count");
}

extern "C" int ktup_shard_adam_catchup(int n_seg, float* const* tables, const int64_t* ld, float* const* states, const int64_t* lds,
                                       const int64_t* const* ids, const int64_t* n_rows, int d, float lr, float eps,
                                       const ktup_adam_t* adam_rule, void* stream) {
  const char* name = "ktup_shard_adam_catchup";
  if (int e = check_kind(name, KTUP_OPT_ADAM, adam_rule, d)) return e;
  KTUP_REQUIRE(n_seg >= 1 && n_seg <= KTUP_SHARD_ADAM_MAX_SEG && tables && ld && states && lds && n_rows && d > 0, "%s: 1..%d segments with their arrays", name,
               KTUP_SHARD_ADAM_MAX_SEG);
  CatchupArgs a{};
  a.n_seg = n_seg;
  int64_t end = 0;
  for (int k = 0; k < n_seg; ++k) {
    KTUP_REQUIRE(tables[k] && states[k] && n_rows[k] >= 0 && ld[k] >= d && lds[k] >= KTUP_SHARD_ADAM_STATE_PITCH(d), "%s: segment %d: null pointer or bad sizes", name, k);
    if (ld[k] % 4 || lds[k] % 4 || !aligned16(tables[k]) || !aligned16(states[k])) return set_error(KTUP_ERR_UNSUPPORTED, "%s: needs 16-byte aligned rows", name);
    a.tab[k] = tables[k]; a.ldt[k] = ld[k]; a.st[k] = states[k]; a.lds[k] = lds[k]; a.ids[k] = ids ? ids[k] : nullptr;
    end += n_rows[k]; a.end[k] = end;
  }
  if (end == 0) return KTUP_OK;
  a.nch = d / 4; a.lr = lr; a.eps = eps;
  a.r = make_rule(adam_rule);
  hipStream_t st = (hipStream_t)stream;
#define KTUP_AF(GL, CPL)                                                                                   \
  {                                                                                                        \
    const int grid = grid_for((end + 4 * (256 / GL) - 1) / (4 * (256 / GL)), 256 * 8);                     \
    if (a.r.wd != 0.f) hipLaunchKernelGGL((adam_catchup_kernel<GL, CPL, true>), dim3(grid), dim3(256), 0, st, a);  \
    else hipLaunchKernelGGL((adam_catchup_kernel<GL, CPL, false>), dim3(grid), dim3(256), 0, st, a);            \
    return check_launch(name);                                                                             \
  }
  if (a.nch <= 16) KTUP_AF(16, 1)
  if (a.nch <= 32) KTUP_AF(32, 1)
  if (a.nch <= 64) KTUP_AF(64, 1)
  if (a.nch <= 128) KTUP_AF(64, 2)
  if (a.nch <= 256) KTUP_AF(64, 4)
#undef KTUP_AF
  return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size %d too large", name, d);
}

extern "C" int ktup_shard_adam_flush(float* table, int64_t ldt, float* state, int64_t lds, int d, int64_t n_rows, float lr, float eps,
                                     const ktup_adam_t* adam_rule, void* stream) {
  float* tabs[1] = {table}; float* sts[1] = {state};
  const int64_t l1[1] = {ldt}, l2[1] = {lds}, n[1] = {n_rows};
  KTUP_REQUIRE(table && state, "ktup_shard_adam_flush: null pointer");
  return ktup_shard_adam_catchup(1, tabs, l1, sts, l2, nullptr, n, d, lr, eps, adam_rule, stream);
}

extern "C" size_t ktup_shard_route_workspace_bytes(int64_t n_entries) {
  if (n_entries <= 0) return 0;
  return (size_t)route_slots(n_entries) * (sizeof(unsigned long long) + sizeof(int32_t));
}

extern "C" size_t ktup_shard_route_sort_bytes(int64_t n_entries, int64_t n_wire_rows) {
  if (n_entries <= 0 || n_wire_rows <= 0) return 0;
  return ((size_t)((n_wire_rows + 2) & ~(int64_t)1) + (size_t)3 * n_entries + (size_t)((n_wire_rows + TILE - 1) / TILE)) * sizeof(int32_t);
}

namespace {

// phase 0: all five launches; 1: the first (scratch init + the KTUP entry list) only; 2: the other four -- so that a caller whose
// scorer needs nothing but the entry list can run the rest of the route on a second stream beside it; 4: the first three (what an
// id exchange waits for); 5: the last two (the sort), which only the row-gradient reduction waits for
int route_impl(const char* name, RouteArgs& a, int pair_a, int pair_b, int64_t* inverse, int64_t* send_ids, int32_t* pair_map,
               int32_t* sort_ws, int32_t* counters, double* zero_doubles, int n_zero_doubles, void* ws, hipStream_t st, int phase = 0) {
  KTUP_REQUIRE(phase >= 0 && phase <= 5, "%s: phase must be 0 (all), 1 (first launch), 2 (the rest), 3 (all, the cursor stays), 4 (ids: init + "
               "insert + finish) or 5 (the sort: scan + scatter)", name);
  a.keep_cursor = phase == 3;
  if (phase == 3) phase = 0;
  KTUP_REQUIRE(inverse && send_ids && sort_ws && counters && ws, "%s: null pointer argument", name);
  KTUP_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7u) == 0 && (reinterpret_cast<uintptr_t>(sort_ws) & 15u) == 0,
               "%s: workspace must be 8-byte, sort_ws 16-byte aligned", name);
  KTUP_REQUIRE(!pair_map || (pair_a >= 0 && pair_a < a.T && pair_b >= 0 && pair_b < a.T && pair_a != pair_b &&
                             a.eoff[pair_a + 1] - a.eoff[pair_a] == a.eoff[pair_b + 1] - a.eoff[pair_b] && a.n == a.block),
               "%s: pair_map needs two tables with equally many entries in a single block", name);
  KTUP_REQUIRE(n_zero_doubles >= 0 && (n_zero_doubles == 0 || zero_doubles), "%s: bad accumulator list", name);
  a.pair_a = pair_a; a.pair_b = pair_b; a.pair_map = pair_map;
  a.slots = route_slots(a.n);
  a.keys = reinterpret_cast<unsigned long long*>(ws);
  a.slot_pos = reinterpret_cast<int32_t*>(a.keys + a.slots);
  a.inverse = inverse; a.send_ids = send_ids;
  a.start = sort_ws;
  a.rank = sort_ws + ((a.W + 2) & ~(int64_t)1);
  a.perm = a.rank + a.n;
  a.skey = a.perm + a.n;
  a.tile_tot = a.skey + a.n;
  const int64_t n_tiles = (a.W + TILE - 1) / TILE;
  a.n_tiles = n_tiles <= MAX_TILES ? (int)n_tiles : 0;
  a.counters = counters; a.zero_d = zero_doubles; a.n_zero_d = n_zero_doubles;
  const int64_t init_items = (int64_t)a.slots > a.W + 1 ? (int64_t)a.slots : a.W + 1;
  if (phase != 2 && phase != 5) hipLaunchKernelGGL(route_init_kernel, dim3(grid_for((init_items + 255) / 256, 1024)), dim3(256), 0, st, a);
  if (phase == 1) return check_launch(name);
  const int grid = grid_for((a.n + 255) / 256, 1024);
  if (phase != 5) {
    // everything the id exchange and the scorer need: send_ids, inverse, pair_map (+ the histogram and each entry's rank in its row)
    hipLaunchKernelGGL(route_insert_kernel, dim3(grid), dim3(256), 0, st, a);
    hipLaunchKernelGGL(route_finish_kernel, dim3(grid), dim3(256), 0, st, a);
    if (int e = check_launch(name)) return e;
    if (phase == 4) return KTUP_OK;
  }
  // the counting sort's second half (only the row-gradient reduction reads it): it may run beside the id exchange and the pack launch
  if (a.n_tiles > 0) {
    hipLaunchKernelGGL(route_tile_scan_kernel, dim3(a.n_tiles), dim3(256), 0, st, a);
  } else if (int e = seg_scan_wide(a.start, a.W, st, name)) {
    return e;
  }
  hipLaunchKernelGGL(route_scatter_kernel, dim3(grid), dim3(256), 0, st, a);
  return check_launch(name);
}

}  // namespace

extern "C" int ktup_shard_route(const int64_t* ids, int64_t n_entries, int64_t block, int n_tables, const int64_t* ent_off, int world,
                                const int64_t* cap, int pair_a, int pair_b, int64_t* inverse, int64_t* send_ids, int32_t* pair_map,
                                int32_t* sort_ws, int32_t* counters, double* zero_doubles, int n_zero_doubles, void* ws, void* stream) {
  const char* name = "ktup_shard_route";
  RouteArgs a{};
  if (int e = fill_route(name, a, ids, n_entries, block, n_tables, ent_off, world, cap)) return e;
  return route_impl(name, a, pair_a, pair_b, inverse, send_ids, pair_map, sort_ws, counters, zero_doubles, n_zero_doubles, ws, (hipStream_t)stream);
}

// The KTUP rec step's route with its entry list built by the first launch (no separate entries launch): batch (*cursor mod
// n_batches) of the id columns u / pos / neg (n_batches x B each; cursor may be NULL: batch 0), entries = [u | pos ; neg |
// item2ent[pos ; neg]] written to `entries` (5B; 3B and two tables when item2ent is NULL), tables 0 / 1 / 2 = users / items /
// entities, pair_map = item wire row -> entity wire row.  *cursor is incremented by the call.
extern "C" int ktup_shard_route_ktup(const int64_t* u, const int64_t* pos_items, const int64_t* neg_items, int64_t B, int64_t n_batches,
                                     int64_t* cursor, const int32_t* item2ent, int64_t ent_pad, int64_t* entries, int world,
                                     const int64_t* cap, int64_t* inverse, int64_t* send_ids, int32_t* pair_map, int32_t* sort_ws,
                                     int32_t* counters, double* zero_doubles, int n_zero_doubles, void* ws, int phase, void* stream) {
  const char* name = "ktup_shard_route_ktup";
  KTUP_REQUIRE(B > 0 && n_batches > 0 && u && pos_items && neg_items && entries, "%s: null pointer argument or empty batch", name);
  const int T = item2ent ? 3 : 2;
  const int64_t eoff[4] = {0, B, 3 * B, 5 * B};
  RouteArgs a{};
  if (int e = fill_route(name, a, entries, eoff[T], eoff[T], T, eoff, world, cap)) return e;
  a.src_u = u; a.src_pos = pos_items; a.src_neg = neg_items; a.B = B; a.n_batches = n_batches; a.cursor = cursor;
  a.item2ent = item2ent; a.ent_pad = ent_pad; a.ids_out = entries;
  return route_impl(name, a, 1, 2, inverse, send_ids, item2ent ? pair_map : nullptr, sort_ws, counters, zero_doubles, n_zero_doubles, ws,
                    (hipStream_t)stream, phase);
}

// The KTUP kg step's route (knowledgable_recommendation.py:346-362: the entity ids of a triple batch and of its corrupted twin):
// batch (*cursor mod n_batches) of the six triple columns -> entries = [ph ; pt ; nh ; nt] (4B, written), one table (entities);
// rels = [pr ; nr] (2B, written) for ktup_train_kg_step_rows.  *cursor is incremented by the call.
extern "C" int ktup_shard_route_kg(const int64_t* ph, const int64_t* pt, const int64_t* pr, const int64_t* nh, const int64_t* nt,
                                   const int64_t* nr, int64_t B, int64_t n_batches, int64_t* cursor, int64_t* entries, int64_t* rels,
                                   int world, const int64_t* cap, int64_t* inverse, int64_t* send_ids, int32_t* sort_ws,
                                   int32_t* counters, double* zero_doubles, int n_zero_doubles, void* ws, int phase, void* stream) {
  const char* name = "ktup_shard_route_kg";
  KTUP_REQUIRE(B > 0 && n_batches > 0 && ph && pt && pr && nh && nt && nr && entries && rels, "%s: null pointer argument or empty batch", name);
  const int64_t eoff[2] = {0, 4 * B};
  RouteArgs a{};
  if (int e = fill_route(name, a, entries, 4 * B, 4 * B, 1, eoff, world, cap)) return e;
  a.src_u = ph; a.src_pos = pt; a.src_neg = nh; a.src_nt = nt; a.src_pr = pr; a.src_nr = nr; a.rel_out = rels;
  a.B = B; a.n_batches = n_batches; a.cursor = cursor; a.ids_out = entries;
  return route_impl(name, a, 0, 0, inverse, send_ids, nullptr, sort_ws, counters, zero_doubles, n_zero_doubles, ws, (hipStream_t)stream, phase);
}

extern "C" int ktup_shard_reduce_rows(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                                      int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, void* stream) {
  const char* name = "ktup_shard_reduce_rows";
  KTUP_REQUIRE(G && sort_ws && gwire && n_entries > 0 && n_wire_rows > 0 && d > 0, "%s: null pointer argument or bad sizes", name);
  KTUP_REQUIRE(n_src > 0 && n_src <= n_entries && src_off >= 0 && src_off <= n_src, "%s: bad source layout", name);
  const int32_t* start = sort_ws;
  const int32_t* perm = sort_ws + ((n_wire_rows + 2) & ~(int64_t)1) + n_entries;
  const int32_t* skey = perm + n_entries;
  const int rc = seg_apply_sorted(G, ldg, d, n_src, src_off, perm, skey, n_entries, start + n_wire_rows, gwire, ldw, (hipStream_t)stream, name);
  if (rc == 1) return set_error(KTUP_ERR_UNSUPPORTED, "%s: needs d %% 4 == 0 and 16-byte aligned rows", name);
  return rc;
}

extern "C" int ktup_shard_ktup_entries(const int64_t* u, const int64_t* pos_items, const int64_t* neg_items, int64_t B,
                                       const int32_t* item2ent, int64_t ent_pad, int64_t* entries, void* stream) {
  const char* name = "ktup_shard_ktup_entries";
  KTUP_REQUIRE(B > 0 && u && pos_items && neg_items && entries, "%s: null pointer argument or empty batch", name);
  hipLaunchKernelGGL(ktup_entries_kernel, dim3(grid_for((2 * B + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream, u, pos_items,
                     neg_items, B, item2ent, ent_pad, entries);
  return check_launch(name);
}

extern "C" int ktup_shard_pack_wire(int n_tables, float* const* tables, const int64_t* ld, const int64_t* cap, int d, const int64_t* ids,
                                    int64_t n_blocks, float* out, int64_t ldo, void* stream) {
  const char* name = "ktup_shard_pack_wire";
  WireTables w{};
  if (int e = fill_wire(name, w, n_tables, tables, ld, nullptr, nullptr, cap, d)) return e;
  KTUP_REQUIRE(ids && out && n_blocks > 0 && ldo >= d && d > 0, "%s: null pointer argument or bad sizes", name);
  bool v4 = d % 4 == 0 && aligned16(out) && ldo % 4 == 0;
  for (int t = 0; t < n_tables; ++t) v4 = v4 && aligned16(tables[t]) && ld[t] % 4 == 0;
  PackWire op{w, ids, out, ldo};
  return launch_rows(op, d, v4, n_blocks * w.capsum, (hipStream_t)stream, name);
}

extern "C" int ktup_shard_apply(int kind, int n_tables, float* const* tables, const int64_t* ld, float* const* states, const int64_t* lds,
                                const int64_t* cap, int d, const int64_t* ids, int64_t n_blocks, float* grads, int64_t ldg, int n_small,
                                int small_rows, float* const* small_grads, float* const* small_p0, float* const* small_s0,
                                float* const* small_p1, float* const* small_s1, const double* small_g64, float lr, float eps,
                                const double* sumsq, int sumsq_slots, float max_norm, const int32_t* skip_count, const double* skip_value,
                                float* loss_step, int n_loss, float* loss_sum, int32_t* skipped_steps, const ktup_adam_t* adam_rule, void* stream) {
  const char* name = "ktup_shard_apply";
  KTUP_REQUIRE(n_loss >= 0 && (n_loss == 0 || (loss_step && loss_sum)), "%s: the loss slots need both arrays", name);
  if (int e = check_kind(name, kind, adam_rule, d)) return e;
  ApplyRows op{};
  if (int e = fill_wire(name, op.w, n_tables, tables, ld, states, lds, cap, d)) return e;
  KTUP_REQUIRE(ids && grads && n_blocks > 0 && ldg >= d && d > 0, "%s: null pointer argument or bad sizes", name);
  KTUP_REQUIRE(n_small >= 0 && n_small <= MAXS && (n_small == 0 || (small_rows > 0 && small_grads && small_p0)), "%s: bad small-table list", name);
  KTUP_REQUIRE(max_norm <= 0.f || (sumsq && (sumsq_slots == 1 || sumsq_slots == NSLOT)), "%s: clipping needs the sum of squared gradients (1 or %d words)", name, NSLOT);
  const bool adam = kind == KTUP_OPT_ADAM;
  const bool adagrad = kind == KTUP_OPT_ADAGRAD || adam;          // "the rows have a state" (Adam: [m | v | last], pitch >= 2 d + 4)
  bool v4 = d % 4 == 0 && aligned16(grads) && ldg % 4 == 0;
  for (int t = 0; t < n_tables; ++t) {
    KTUP_REQUIRE(!adagrad || (states && states[t]), "%s: table %d: Adagrad state missing", name, t);
    v4 = v4 && aligned16(tables[t]) && ld[t] % 4 == 0 && (!adagrad || (aligned16(states[t]) && op.w.lds[t] % 4 == 0));
  }
  for (int k = 0; k < n_small; ++k) {
    KTUP_REQUIRE(small_grads[k] && small_p0[k] && (!adagrad || (small_s0 && small_s0[k])), "%s: small table %d: null pointer", name, k);
    op.sg[k] = small_grads[k]; op.sp0[k] = small_p0[k]; op.ss0[k] = small_s0 ? small_s0[k] : nullptr;
    op.sp1[k] = small_p1 ? small_p1[k] : nullptr; op.ss1[k] = small_s1 ? small_s1[k] : nullptr;
    KTUP_REQUIRE(!op.sp1[k] || !adagrad || op.ss1[k], "%s: small table %d: second table's Adagrad state missing", name, k);
    v4 = v4 && aligned16(op.sg[k]) && aligned16(op.sp0[k]) && aligned16(op.ss0[k]) && aligned16(op.sp1[k]) && aligned16(op.ss1[k]);
  }
  op.ids = ids; op.W = n_blocks * op.w.capsum; op.g = grads; op.ldg = ldg;
  op.n_small = n_small; op.small_rows = small_rows > 0 ? small_rows : 1; op.small_g64 = small_g64; op.d = d;
  op.lr = lr; op.eps = eps; op.max_norm = max_norm; op.sumsq = sumsq; op.sumsq_slots = sumsq_slots; op.skip_i = skip_count; op.skip_d = skip_value; op.adagrad = adagrad;
  op.loss_step = loss_step; op.n_loss = n_loss; op.loss_sum = loss_sum; op.skipped = skipped_steps;
  op.adam = adam; op.small_lds = adam ? KTUP_SHARD_ADAM_STATE_PITCH(d) : d;
  if (adam) {
    op.ar = make_rule(adam_rule);
    for (int t = 0; t < n_tables; ++t) KTUP_REQUIRE(op.w.lds[t] >= KTUP_SHARD_ADAM_STATE_PITCH(d), "%s: table %d: an Adam state row is [m | v | last]: pitch >= 2 d + 4", name, t);
    if (!v4) return set_error(KTUP_ERR_UNSUPPORTED, "%s: Adam rows need d %% 4 == 0 and 16-byte aligned tables, states and gradients", name);
    ApplyRowsT<1> oa;                                     // same members, the lazy instantiations of the row rule
    ApplyRowsT<2> ow;
    static_assert(sizeof(oa) == sizeof(op) && sizeof(ow) == sizeof(op), "the ApplyRowsT instantiations share one layout");
    if (op.ar.wd != 0.f) {
      memcpy(&ow, &op, sizeof(op));
      return launch_rows(ow, d, v4, op.W + (int64_t)n_small * op.small_rows, (hipStream_t)stream, name);
    }
    memcpy(&oa, &op, sizeof(op));
    return launch_rows(oa, d, v4, op.W + (int64_t)n_small * op.small_rows, (hipStream_t)stream, name);
  }
  return launch_rows(op, d, v4, op.W + (int64_t)n_small * op.small_rows, (hipStream_t)stream, name);
}


extern "C" int64_t ktup_shard_reduce_list_len(int64_t n_entries, int d) {
  if (n_entries <= 0 || d <= 0 || d % 4) return 0;
  return 2 * fused_grid(n_entries, d);
}

namespace {
// the replicas of ktup_train_rec_step_rows_ws and where their sums go: rep_dst = {gP, gPn, gR or NULL, gRn or NULL}
int fill_rep(const char* name, XNormArgs& x, float* rep, int n_rep, int64_t rep_elems, float* const* rep_dst) {
  if (!rep) return KTUP_OK;
  KTUP_REQUIRE(n_rep >= 1 && rep_elems > 0 && rep_dst && rep_dst[0] && rep_dst[1] && (rep_dst[2] == nullptr) == (rep_dst[3] == nullptr),
               "%s: the replicas need their count, their size and the gradients they fold into (gP, gPn, and gR / gRn together or not at all)", name);
  x.rep = rep; x.n_rep = n_rep; x.rep_elems = rep_elems;
  for (int k = 0; k < 4; ++k) x.rdst[k] = rep_dst[k];
  return KTUP_OK;
}
}  // namespace

extern "C" int ktup_shard_reduce_norm(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                                      int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, int32_t* xkeys, int n_small,
                                      float* const* small_grads, int64_t small_elems, float small_weight, double* sumsq, int n_slots,
                                      int dup_only, double* fold, int n_fold, int64_t* cursor, void* stream) {
  return ktup_shard_reduce_norm_fold(G, ldg, d, n_src, src_off, sort_ws, n_entries, n_wire_rows, gwire, ldw, xkeys, n_small, small_grads,
                                     small_elems, small_weight, sumsq, n_slots, dup_only, fold, n_fold, cursor, nullptr, 0, nullptr, stream);
}

extern "C" int ktup_shard_reduce_norm_fold(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                                           int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, int32_t* xkeys, int n_small,
                                           float* const* small_grads, int64_t small_elems, float small_weight, double* sumsq, int n_slots,
                                           int dup_only, double* fold, int n_fold, int64_t* cursor, float* rep, int n_rep,
                                           float* const* rep_dst, void* stream) {
  const char* name = "ktup_shard_reduce_norm";
  FusedArgs a{};
  if (int e = fill_fused(name, a, G, ldg, d, n_src, src_off, sort_ws, n_entries, n_wire_rows, gwire, ldw, xkeys)) return e;
  KTUP_REQUIRE(sumsq && (n_slots == 1 || n_slots == NSLOT), "%s: the sum of squares lives in 1 or %d words", name, NSLOT);
  KTUP_REQUIRE(n_small >= 0 && n_small <= MAXS && (n_small == 0 || (small_grads && small_elems > 0)), "%s: bad small-gradient list", name);
  a.sumsq = sumsq; a.slots = n_slots; a.dup_only = dup_only != 0;
  hipStream_t st = (hipStream_t)stream;
  const int64_t grid = fused_grid(n_entries, d);
  XNormArgs x{};
  x.n_small = n_small; x.small_elems = small_elems > 0 ? small_elems : 1; x.small_weight = small_weight;
  for (int k = 0; k < n_small; ++k) {
    KTUP_REQUIRE(small_grads[k], "%s: small gradient %d is null", name, k);
    x.sg[k] = small_grads[k];
  }
  x.sumsq = sumsq; x.slots = n_slots;
  KTUP_REQUIRE(n_fold >= 0 && (n_fold == 0 || fold), "%s: fold needs its array", name);
  x.fold = n_fold > 0 ? fold : nullptr; x.n_fold = n_fold; x.cursor = cursor;
  if (int e = fill_rep(name, x, rep, n_rep, small_elems, rep_dst)) return e;       // (the replicas' sums ARE the small gradients: same size)
  return launch_fused<0, 0>(a, grid, st, name, (const ApplyRows*)nullptr, 0, &x);     // the walk + (extra workgroups) the small gradients, the fold, the cursor
}

extern "C" int ktup_shard_reduce_apply(int kind, int n_tables, float* const* tables, const int64_t* ld, float* const* states,
                                       const int64_t* lds, const int64_t* cap, const int64_t* ids, int64_t n_blocks, const float* G,
                                       int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws, int64_t n_entries,
                                       float* gwire, int64_t ldw, const int32_t* xkeys, int n_small, int small_rows,
                                       float* const* small_grads, float* const* small_p0, float* const* small_s0, float* const* small_p1,
                                       float* const* small_s1, const double* small_g64, float lr, float eps, const double* sumsq,
                                       int sumsq_slots, float max_norm, const int32_t* skip_count, const double* skip_value,
                                       float* loss_step, int n_loss, float* loss_sum, int32_t* skipped_steps, const ktup_adam_t* adam_rule,
                                       void* stream) {
  const char* name = "ktup_shard_reduce_apply";
  KTUP_REQUIRE(n_loss >= 0 && (n_loss == 0 || (loss_step && loss_sum)), "%s: the loss slots need both arrays", name);
  if (int e = check_kind(name, kind, adam_rule, d)) return e;
  ApplyRows op{};
  if (int e = fill_wire(name, op.w, n_tables, tables, ld, states, lds, cap, d)) return e;
  KTUP_REQUIRE(ids && n_blocks > 0, "%s: null pointer argument or bad sizes", name);
  const int64_t W = n_blocks * op.w.capsum;
  FusedArgs a{};
  if (int e = fill_fused(name, a, G, ldg, d, n_src, src_off, sort_ws, n_entries, W, gwire, ldw, const_cast<int32_t*>(xkeys))) return e;
  KTUP_REQUIRE(n_small >= 0 && n_small <= MAXS && (n_small == 0 || (small_rows > 0 && small_grads && small_p0)), "%s: bad small-table list", name);
  KTUP_REQUIRE(max_norm <= 0.f || (sumsq && (sumsq_slots == 1 || sumsq_slots == NSLOT)), "%s: clipping needs the sum of squared gradients (1 or %d words)", name, NSLOT);
  const bool adam = kind == KTUP_OPT_ADAM;
  const bool adagrad = kind == KTUP_OPT_ADAGRAD || adam;          // "the rows have a state" (Adam: [m | v | last], pitch >= 2 d + 4)
  bool v4 = true;
  for (int t = 0; t < n_tables; ++t) {
    KTUP_REQUIRE(!adagrad || (states && states[t]), "%s: table %d: Adagrad state missing", name, t);
    v4 = v4 && aligned16(tables[t]) && ld[t] % 4 == 0 && (!adagrad || (aligned16(states[t]) && op.w.lds[t] % 4 == 0));
  }
  for (int k = 0; k < n_small; ++k) {
    KTUP_REQUIRE(small_grads[k] && small_p0[k] && (!adagrad || (small_s0 && small_s0[k])), "%s: small table %d: null pointer", name, k);
    op.sg[k] = small_grads[k]; op.sp0[k] = small_p0[k]; op.ss0[k] = small_s0 ? small_s0[k] : nullptr;
    op.sp1[k] = small_p1 ? small_p1[k] : nullptr; op.ss1[k] = small_s1 ? small_s1[k] : nullptr;
    KTUP_REQUIRE(!op.sp1[k] || !adagrad || op.ss1[k], "%s: small table %d: second table's Adagrad state missing", name, k);
    v4 = v4 && aligned16(op.sg[k]) && aligned16(op.sp0[k]) && aligned16(op.ss0[k]) && aligned16(op.sp1[k]) && aligned16(op.ss1[k]);
  }
  if (!v4) return set_error(KTUP_ERR_UNSUPPORTED, "%s: tables, states and gradients must be 16-byte aligned with pitches %% 4 == 0", name);
  a.sumsq = const_cast<double*>(sumsq); a.slots = sumsq_slots; a.w = op.w; a.ids = ids; a.lr = lr; a.eps = eps; a.max_norm = max_norm;
  a.adagrad = adagrad; a.skip_i = skip_count; a.skip_d = skip_value;
  a.adam = adam; op.adam = adam; op.small_lds = adam ? KTUP_SHARD_ADAM_STATE_PITCH(d) : d;
  if (adam) {
    a.ar = op.ar = make_rule(adam_rule);
    for (int t = 0; t < n_tables; ++t) KTUP_REQUIRE(op.w.lds[t] >= KTUP_SHARD_ADAM_STATE_PITCH(d), "%s: table %d: an Adam state row is [m | v | last]: pitch >= 2 d + 4", name, t);
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t grid = fused_grid(n_entries, d);
  // the listed boundary rows from gw (then zero-filled) and the small tables ride in the same launch
  op.ids = ids; op.W = 2 * grid; op.g = gwire; op.ldg = ldw; op.xkeys = xkeys;
  op.n_small = n_small; op.small_rows = small_rows > 0 ? small_rows : 1; op.small_g64 = small_g64; op.d = d;
  op.lr = lr; op.eps = eps; op.max_norm = max_norm; op.sumsq = sumsq; op.sumsq_slots = sumsq_slots; op.skip_i = skip_count; op.skip_d = skip_value;
  op.adagrad = adagrad;
  op.loss_step = loss_step; op.n_loss = n_loss; op.loss_sum = loss_sum; op.skipped = skipped_steps;
  if (adam) {
    if (op.ar.wd != 0.f) {
      ApplyRowsT<2> ow;
      memcpy(&ow, &op, sizeof(op));
      return launch_fused<1, 2>(a, grid, st, name, &ow, op.W + (int64_t)n_small * op.small_rows);
    }
    ApplyRowsT<1> oa;
    memcpy(&oa, &op, sizeof(op));
    return launch_fused<1, 1>(a, grid, st, name, &oa, op.W + (int64_t)n_small * op.small_rows);
  }
  return launch_fused<1>(a, grid, st, name, &op, op.W + (int64_t)n_small * op.small_rows);
}

extern "C" int ktup_shard_zero_shared_rows(const int32_t* sort_ws, int64_t n_entries, int64_t n_wire_rows, const int64_t* inverse, float* gwire,
                                           int64_t ldw, int d, void* stream) {
  const char* name = "ktup_shard_zero_shared_rows";
  KTUP_REQUIRE(sort_ws && inverse && gwire && n_entries > 0 && n_wire_rows > 0 && d > 0 && ldw >= d, "%s: null pointer argument or bad sizes", name);
  if (d % 4 || ldw % 4 || !aligned16(gwire)) return set_error(KTUP_ERR_UNSUPPORTED, "%s: needs d %% 4 == 0 and 16-byte aligned rows", name);
  const int32_t* rank = sort_ws + ((n_wire_rows + 2) & ~(int64_t)1);
  hipLaunchKernelGGL(zero_shared_rows_kernel, dim3(grid_for((n_entries + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream, rank, inverse,
                     n_entries, gwire, ldw, d / 4);
  return check_launch(name);
}

extern "C" int ktup_shard_reduce_store(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                                       int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, void* stream) {
  return ktup_shard_reduce_store_fold(G, ldg, d, n_src, src_off, sort_ws, n_entries, n_wire_rows, gwire, ldw, nullptr, 0, 0, nullptr, stream);
}

extern "C" int ktup_shard_reduce_store_fold(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* sort_ws,
                                            int64_t n_entries, int64_t n_wire_rows, float* gwire, int64_t ldw, float* rep, int n_rep,
                                            int64_t rep_elems, float* const* rep_dst, void* stream) {
  const char* name = "ktup_shard_reduce_store";
  FusedArgs a{};
  int32_t none = 0;
  if (int e = fill_fused(name, a, G, ldg, d, n_src, src_off, sort_ws, n_entries, n_wire_rows, gwire, ldw, &none)) return e;
  a.xkeys = nullptr;
  XNormArgs x{};
  x.small_weight = 1.f;
  if (int e = fill_rep(name, x, rep, n_rep, rep_elems, rep_dst)) return e;
  return launch_fused<2, 0>(a, fused_grid(n_entries, d), (hipStream_t)stream, name, (const ApplyRows*)nullptr, 0, rep ? &x : nullptr);
}

extern "C" int ktup_shard_bucket(int mode, int n_small, float* const* small_grads, int64_t small_elems, double* bucket,
                                 const double* sumsq_local, int sumsq_slots, const int32_t* overflow, double* sumsq_total, double small_weight,
                                 void* stream) {
  const char* name = "ktup_shard_bucket";
  KTUP_REQUIRE((mode == 0 || mode == 1) && n_small >= 0 && n_small <= MAXS && small_elems >= 0 && bucket, "%s: bad arguments", name);
  BucketArgs a{};
  a.n_small = n_small; a.small_elems = small_elems > 0 ? small_elems : 1; a.bucket = bucket;
  if (n_small == 0) a.small_elems = 1;
  for (int k = 0; k < n_small; ++k) {
    KTUP_REQUIRE(mode == 1 || (small_grads && small_grads[k]), "%s: small gradient %d is null", name, k);
    a.sg[k] = small_grads ? small_grads[k] : nullptr;
  }
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    KTUP_REQUIRE(sumsq_local && sumsq_slots >= 1 && overflow, "%s: mode 0 needs the local sum of squares and the overflow word", name);
    a.sumsq_in = sumsq_local; a.sumsq_slots = sumsq_slots; a.overflow = overflow;
    const int64_t N = (int64_t)n_small * a.small_elems + 2;
    hipLaunchKernelGGL(bucket_kernel<0>, dim3(grid_for((N + 1023) / 1024, 64)), dim3(1024), 0, st, a);
  } else {
    KTUP_REQUIRE(sumsq_total, "%s: mode 1 needs the output", name);
    a.sumsq_out = sumsq_total; a.small_weight = small_weight;
    hipLaunchKernelGGL(bucket_kernel<1>, dim3(1), dim3(1024), 0, st, a);
  }
  return check_launch(name);
}

extern "C" int ktup_zero_async(void* ptr, int64_t nbytes, void* stream) {
  KTUP_REQUIRE(nbytes >= 0 && nbytes % 16 == 0 && (nbytes == 0 || (ptr && aligned16(ptr))), "ktup_zero_async: needs a 16-byte aligned buffer of 16-byte multiples");
  if (nbytes == 0) return KTUP_OK;
  hipLaunchKernelGGL(zero_f4_kernel, dim3(grid_for((nbytes / 16 + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<float4*>(ptr), nbytes / 16);
  return check_launch("ktup_zero_async");
}
