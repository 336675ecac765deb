// Row pack / unpack for row-sharded embedding tables (config 5: tables partitioned by `row % world` across the GPUs of
// one node, lookups exchanged with RCCL all-to-all over xGMI).  The reference has no distributed code; these are the
// two device-side halves of the exchange:
//   pack  : out[k, :]  = table[ids[k], :]        (owner side: rows requested by a peer -> contiguous send buffer)
//   unpack: gtable[ids[k], :] += rows[k, :]      (owner side: row gradients coming back -> shard gradient, atomics)
// Same lane-group-per-row mapping as the scorers (contiguous 16-B pieces per row).
// Negative ids are padding (fixed-capacity id lists keep the host out of the loop: the number of distinct ids of a batch
// stays on the device): pack writes a zero row for them, unpack and the sparse step skip them.
//   dedupe: distinct ids of a batch + each entry's position among them, by an open-addressing hash table in caller scratch
//           (two launches; replaces sort + unique + a host sync for the size).
#include "ktup_rows.h"

using namespace ktup;

namespace {

struct PackRows {
  const float* T; int64_t ldt; const int64_t* ids; float* out; int64_t ldo;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    V x[CPL];
    const int64_t id = ids[row];
    if (id >= 0) {
      cx.load(x, T + id * ldt);
    } else {
#pragma unroll
      for (int j = 0; j < CPL; ++j) vzero(x[j]);
    }
    V* o = reinterpret_cast<V*>(out + row * ldo);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = cx.lane + j * G;
      if (c < cx.nch) o[c] = x[j];
    }
  }
};

struct UnpackAdd {
  const float* rows; int64_t ldr; const int64_t* ids; float* gT; int64_t ldg;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    V x[CPL];
    const int64_t id = ids[row];
    if (id < 0) return;
    cx.load(x, rows + row * ldr);
    cx.scatter_add(gT + id * ldg, x);
  }
};

// Owner-side sparse optimizer: update only the rows a step touched.
//   g' = coef * g,  coef = min(1, max_norm / (sqrt(*sumsq) + 1e-6))   (the global-norm clip; sumsq is the all-reduced sum)
//   SGD     : p -= lr * g'
//   Adagrad : sum += g'^2;  p -= lr * g' / (sqrt(sum) + eps)
// With weight_decay = 0 (and no momentum) this IS the reference's dense step: rows with a zero gradient do not move under
// either rule (utils/trainer.py:63-77 with l2_lambda = 0).  `ids` must be unique (one update per row).
KTUP_DEV float up1(float& p, float& st, float g, float lr, float eps, bool adagrad) {
  if (adagrad) {
    st = fmaf(g, g, st);
    p = p - lr * (g / (sqrtf(st) + eps));
  } else {
    p = fmaf(-lr, g, p);
  }
  return p;
}
KTUP_DEV void upv(float& p, float& st, float g, float lr, float eps, bool adagrad) { up1(p, st, g, lr, eps, adagrad); }
KTUP_DEV void upv(float4& p, float4& st, float4 g, float lr, float eps, bool adagrad) {
  up1(p.x, st.x, g.x, lr, eps, adagrad); up1(p.y, st.y, g.y, lr, eps, adagrad);
  up1(p.z, st.z, g.z, lr, eps, adagrad); up1(p.w, st.w, g.w, lr, eps, adagrad);
}

struct SparseRowStep {
  float* T; int64_t ldt; float* S; int64_t lds; const int64_t* ids; const float* g; int64_t ldg;
  float lr, eps, max_norm; const double* sumsq; bool adagrad;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    float coef = 1.f;
    if (max_norm > 0.f) {
      const float c = max_norm / ((float)sqrt(*sumsq) + 1e-6f);
      coef = c < 1.f ? c : 1.f;
    }
    const int64_t r = ids[row];
    if (r < 0) return;
    V p[CPL], st[CPL], gr[CPL];
    cx.load(p, T + r * ldt);
    cx.load(gr, g + row * ldg);
    if (adagrad) cx.load(st, S + r * lds);
    V* po = reinterpret_cast<V*>(T + r * ldt);
    V* so = reinterpret_cast<V*>(S + r * lds);
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
};

// ---- dedupe: open addressing, linear probing, keys = ids (>= 0), empty = -1
KTUP_DEV uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void dedupe_init_kernel(unsigned long long* __restrict__ keys, uint64_t slots, int64_t* __restrict__ uniq,
                                                          int64_t n, int32_t* __restrict__ count) {
  const int64_t total = (int64_t)slots + n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    if (i < (int64_t)slots) keys[i] = ~0ull; else uniq[i - (int64_t)slots] = -1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = 0;
}

__global__ __launch_bounds__(256) void dedupe_insert_kernel(const int64_t* __restrict__ ids, int64_t n, unsigned long long* keys,
                                                            int32_t* __restrict__ slot_idx, uint64_t mask, int32_t* count,
                                                            int64_t* __restrict__ uniq) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const unsigned long long id = (unsigned long long)ids[e];
    uint64_t h = mix64(id) & mask;
    bool created = false;
    for (;;) {
      const unsigned long long prev = atomicCAS(keys + h, ~0ull, id);
      if (prev == ~0ull) { created = true; break; }   // this thread created the entry: it names the compact row
      if (prev == id) break;
      h = (h + 1) & mask;
    }
    // compact row numbers: ONE counter atomic per wave (a per-thread atomicAdd on one address serialises the whole batch)
    const unsigned long long made = __ballot(created);
    if (created) {
      const int lane = threadIdx.x & 63;
      const int leader = __ffsll((long long)made) - 1;
      int32_t base = 0;
      if (lane == leader) base = atomicAdd(count, __popcll(made));
      base = __shfl(base, leader, 64);
      const int32_t idx = base + __popcll(made & ((1ull << lane) - 1ull));
      slot_idx[h] = idx;
      uniq[idx] = (int64_t)id;
    }
  }
}

__global__ __launch_bounds__(256) void dedupe_lookup_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                            const unsigned long long* __restrict__ keys,
                                                            const int32_t* __restrict__ slot_idx, uint64_t mask,
                                                            int64_t* __restrict__ inverse) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const unsigned long long id = (unsigned long long)ids[e];
    uint64_t h = mix64(id) & mask;
    while (keys[h] != id) h = (h + 1) & mask;
    inverse[e] = slot_idx[h];
  }
}

uint64_t dedupe_slots(int64_t n) {
  uint64_t s = 64;
  while (s < (uint64_t)(2 * n)) s <<= 1;
  return s;
}

}  // namespace

// Distinct ids of a batch without leaving the device: uniq[0 .. *n_unique) = the distinct ids (in no particular order),
// uniq[*n_unique .. n) = -1, inverse[e] = position of ids[e] in uniq.  ids must be >= 0.
extern "C" size_t ktup_shard_dedupe_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return (size_t)dedupe_slots(n) * (sizeof(unsigned long long) + sizeof(int32_t));
}

extern "C" int ktup_shard_dedupe(const int64_t* ids, int64_t n, int64_t* uniq, int64_t* inverse, int32_t* n_unique, void* ws,
                                 void* stream) {
  const char* name = "ktup_shard_dedupe";
  KTUP_REQUIRE(n >= 0 && n < (1ll << 30), "%s: bad size", name);
  KTUP_REQUIRE(n_unique, "%s: null pointer argument", name);
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (hipMemsetAsync(n_unique, 0, sizeof(int32_t), st) != hipSuccess) return check_launch(name);
    return KTUP_OK;
  }
  KTUP_REQUIRE(ids && uniq && inverse && ws, "%s: null pointer argument", name);
  KTUP_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7u) == 0, "%s: workspace must be 8-byte aligned", name);
  const uint64_t slots = dedupe_slots(n);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws);
  int32_t* slot_idx = reinterpret_cast<int32_t*>(keys + slots);
  hipLaunchKernelGGL(dedupe_init_kernel, dim3(grid_for(((int64_t)slots + n + 255) / 256, 1024)), dim3(256), 0, st, keys, slots, uniq, n,
                     n_unique);
  const int grid = grid_for((n + 255) / 256, 1024);
  hipLaunchKernelGGL(dedupe_insert_kernel, dim3(grid), dim3(256), 0, st, ids, n, keys, slot_idx, slots - 1, n_unique, uniq);
  hipLaunchKernelGGL(dedupe_lookup_kernel, dim3(grid), dim3(256), 0, st, ids, n, keys, slot_idx, slots - 1, inverse);
  return check_launch(name);
}

extern "C" int ktup_shard_sparse_step(int kind, float* table, int64_t ldt, float* state, int64_t lds, int d, const int64_t* ids,
                                      int64_t n, const float* grows, int64_t ldg, float lr, float eps, const double* sumsq,
                                      float max_norm, void* stream) {
  const char* name = "ktup_shard_sparse_step";
  KTUP_REQUIRE(kind == KTUP_OPT_SGD || kind == KTUP_OPT_ADAGRAD, "%s: only plain SGD and Adagrad have an exact row-sparse form", name);
  KTUP_REQUIRE(d > 0 && n >= 0 && ldg >= d, "%s: bad sizes", name);
  if (n == 0) return KTUP_OK;
  KTUP_REQUIRE(table && ids && grows && (kind == KTUP_OPT_SGD || state), "%s: null pointer argument", name);
  KTUP_REQUIRE(max_norm <= 0.f || sumsq, "%s: clipping needs the (all-reduced) sum of squared gradients", name);
  SparseRowStep op{table, ldt, state ? state : table, state ? lds : ldt, ids, grows, ldg, lr, eps, max_norm, sumsq, kind == KTUP_OPT_ADAGRAD};
  return launch_rows(op, d, can_vec4(d, {table, state, grows}, {ldt, state ? lds : 4, ldg}), n, (hipStream_t)stream, name);
}

extern "C" int ktup_shard_pack_rows(const float* table, int64_t ldt, int d, const int64_t* ids, int64_t n, float* out,
                                    int64_t ldo, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && ldo >= d, "ktup_shard_pack_rows: bad sizes");
  KTUP_REQUIRE(n == 0 || (table && ids && out), "ktup_shard_pack_rows: null pointer argument");
  PackRows op{table, ldt, ids, out, ldo};
  return launch_rows(op, d, can_vec4(d, {table, out}, {ldt, ldo}), n, (hipStream_t)stream, "ktup_shard_pack_rows");
}

extern "C" int ktup_shard_unpack_rows_add(const float* rows, int64_t ldr, int d, const int64_t* ids, int64_t n, float* gtable,
                                          int64_t ldg, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && ldr >= d, "ktup_shard_unpack_rows_add: bad sizes");
  KTUP_REQUIRE(n == 0 || (rows && ids && gtable), "ktup_shard_unpack_rows_add: null pointer argument");
  UnpackAdd op{rows, ldr, ids, gtable, ldg};
  return launch_rows(op, d, can_vec4(d, {rows, gtable}, {ldr, ldg}), n, (hipStream_t)stream, "ktup_shard_unpack_rows_add");
}
