// K19: negative sampling on the device, with the constraints of jTransUP/utils/data.py:12-85
//   rec (getNegRatings)        : uniform item  != the positive, not rated by the user in ANY split (per-user bitmap),
//                                and not already used as a negative in this batch;
//   kg  (getTrainTripleBatch)  : fair coin per triple -> corrupt head or tail with a uniform entity != the original that
//                                does not make a known-true triple (binary search in the sorted 64-bit keys of all splits).
// The reference draws from python's Mersenne Twister on the host, once per row, inside the training loop; RNG parity
// is not achievable, so parity here means the same constraints and the same distribution (Philox4x32-10 counters).
//
// Every output is a VALID row index, always: when the random tries are exhausted a deterministic scan looks for an admissible
// candidate, and when none exists at all the row gets an in-range stand-in and *fail_count is incremented -- the host checks
// the counter (DeviceSampler.check) instead of ever handing an out-of-range id to the scoring kernels.
//
// Batch uniqueness is DETERMINISTIC: (seed, offset, inputs) fix the batch, whatever the thread timing.  Rounds: in round t every
// still-open row proposes its t-th draw with atomicMin(owner[item], (t << 32) | row); after a barrier the row whose key stands
// owns the item (earlier rounds beat later ones, lower rows beat higher ones inside a round).  One workgroup runs all rounds
// (a batch is <= a few thousand rows; n <= n_items or uniqueness is impossible anyway), so the barrier is __syncthreads.
#include "ktup_common.h"

using namespace ktup;

namespace {

constexpr int MAX_TRIES = 4096;
constexpr int UNIQ_THREADS = 1024;

KTUP_DEV uint32_t draw(const Philox& ph, uint64_t ctr, int which) {
  const uint4 r = ph(ctr, 0x4e454753ull /* "NEGS" */);
  return which == 0 ? r.x : which == 1 ? r.y : which == 2 ? r.z : r.w;
}
KTUP_DEV int64_t bounded(uint32_t x, int64_t n) { return (int64_t)(((uint64_t)x * (uint64_t)n) >> 32); }
KTUP_DEV int64_t draw_item(const Philox& ph, uint64_t offset, int64_t row, int tries, int64_t n) {
  const uint64_t ctr = offset + (uint64_t)row * MAX_TRIES + tries;
  return bounded(draw(ph, ctr >> 2, (int)(ctr & 3)), n);
}
KTUP_DEV bool rated(const uint32_t* ubits, int64_t c) { return ubits && ((ubits[c >> 5] >> (c & 31)) & 1u); }

// one row without batch-uniqueness
KTUP_DEV int64_t rec_pick(const Philox& ph, uint64_t offset, int64_t b, int64_t user, int64_t p, int64_t n_items,
                          const uint32_t* __restrict__ bitmap, int64_t words, int32_t* __restrict__ fail) {
  const uint32_t* ubits = bitmap ? bitmap + user * words : nullptr;
  int64_t pick = -1;
  for (int tries = 0; tries < MAX_TRIES && pick < 0; ++tries) {
    const int64_t c = draw_item(ph, offset, b, tries, n_items);
    if (c != p && !rated(ubits, c)) pick = c;
  }
  if (pick < 0) {                                          // a user who rated (nearly) everything: scan from the last draw
    const int64_t s0 = draw_item(ph, offset, b, MAX_TRIES - 1, n_items);
    for (int64_t k = 0; k < n_items && pick < 0; ++k) {
      const int64_t c = s0 + k < n_items ? s0 + k : s0 + k - n_items;
      if (c != p && !rated(ubits, c)) pick = c;
    }
  }
  if (pick < 0) {                                          // no admissible item exists: in-range stand-in + error count
    if (fail) atomicAdd(fail, 1);
    pick = p + 1 < n_items ? p + 1 : 0;
  }
  return pick;
}

// no batch-uniqueness: rows are independent
__global__ __launch_bounds__(256) void negsample_rec_kernel(const int64_t* __restrict__ u, const int64_t* __restrict__ pos,
                                                            int64_t n, int64_t n_items, const uint32_t* __restrict__ bitmap,
                                                            int64_t words, uint64_t seed, uint64_t offset,
                                                            int64_t* __restrict__ neg, int32_t* __restrict__ fail) {
  const Philox ph(seed);
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n; b += (int64_t)gridDim.x * 256)
    neg[b] = rec_pick(ph, offset, b, u[b], pos[b], n_items, bitmap, words, fail);
}

// batch-unique negatives by ONE workgroup of UNIQ_THREADS threads, deterministic rounds (see the file comment).  owner[] is
// all-ones on entry; every thread of the workgroup must call this (barriers inside).
KTUP_DEV void rec_unique_rounds(const Philox& ph, const int64_t* __restrict__ u, const int64_t* __restrict__ pos, int64_t n,
                                int64_t n_items, const uint32_t* __restrict__ bitmap, int64_t words, uint64_t offset, int64_t* neg,
                                unsigned long long* owner, int32_t* __restrict__ fail, bool leave_clean) {
  const int tid = threadIdx.x;
  for (int64_t b = tid; b < n; b += UNIQ_THREADS) neg[b] = -1;        // a thread only ever touches its own rows of neg[]
  bool all_done = false;
  for (int t = 0; t < MAX_TRIES && !all_done; ++t) {
    for (int64_t b = tid; b < n; b += UNIQ_THREADS) {
      if (neg[b] >= 0) continue;
      const int64_t c = draw_item(ph, offset, b, t, n_items);
      const uint32_t* ubits = bitmap ? bitmap + u[b] * words : nullptr;
      if (c == pos[b] || rated(ubits, c)) { neg[b] = -1; continue; }
      atomicMin(owner + c, ((unsigned long long)t << 32) | (unsigned long long)b);
      neg[b] = -(c + 2);                                              // proposed c this round
    }
    __syncthreads();                                                  // all proposals of round t are in (atomics live in L2)
    int open = 0;
    for (int64_t b = tid; b < n; b += UNIQ_THREADS) {
      const int64_t v = neg[b];
      if (v >= 0) continue;
      if (v <= -2) {
        const int64_t c = -(v + 2);
        const unsigned long long key = ((unsigned long long)t << 32) | (unsigned long long)b;
        if (__hip_atomic_load(owner + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key) { neg[b] = c; continue; }
        neg[b] = -1;
      }
      open = 1;
    }
    all_done = !__syncthreads_or(open);
  }
  if (!all_done) {
    // tries exhausted for some rows (more rows than admissible items, or a nearly full user): thread 0 serves them in row order
    if (tid == 0) {
      for (int64_t b = 0; b < n; ++b) {
        if (neg[b] >= 0) continue;
        const uint32_t* ubits = bitmap ? bitmap + u[b] * words : nullptr;
        const int64_t p = pos[b], s0 = draw_item(ph, offset, b, MAX_TRIES - 1, n_items);
        int64_t pick = -1;
        for (int64_t k = 0; k < n_items && pick < 0; ++k) {
          const int64_t c = s0 + k < n_items ? s0 + k : s0 + k - n_items;
          if (c == p || rated(ubits, c)) continue;
          if (__hip_atomic_load(owner + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ~0ull) continue;
          pick = c;
        }
        if (pick >= 0) {
          __hip_atomic_store(owner + pick, (unsigned long long)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          if (fail) atomicAdd(fail, 1);
          pick = p + 1 < n_items ? p + 1 : 0;
        }
        neg[b] = pick;
      }
    }
    __syncthreads();
  }
  // leave owner[] all-ones again: every entry a proposal touched ended up with the key of the row that won it (earlier rounds
  // and lower rows win, nothing overwrites a smaller key), so clearing the batch's own negatives clears everything
  if (leave_clean)
    for (int64_t b = tid; b < n; b += UNIQ_THREADS) __hip_atomic_store(owner + neg[b], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same rounds for a batch of at most UNIQ_THREADS rows on a catalogue whose owner table fits LDS (n_items <= FAST_ITEMS):
// one row per thread with its state in registers, owner[] in LDS (ds_min_u64 / ds_read instead of L2 round trips), and the first
// PRE candidates of every row -- with the bitmap words that decide them -- fetched up front in ONE batch of independent loads.
// Same draws, same keys, same winners as rec_unique_rounds: 13 us -> ~6 us for B = 512 on 3240 items.
constexpr int KG_COARSE = 4096;                            // 32 KB of LDS
constexpr int FAST_ITEMS = 8000, PRE = 4;                  // 8000 x 8 B of LDS

KTUP_DEV void rec_unique_fast(const Philox& ph, const int64_t* __restrict__ u, const int64_t* __restrict__ pos, int64_t n,
                              int64_t n_items, const uint32_t* __restrict__ bitmap, int64_t words, uint64_t offset, int64_t* neg,
                              unsigned long long* lown, int32_t* __restrict__ fail) {
  const int tid = threadIdx.x;
  const bool has = tid < n;
  const int64_t user = has ? u[tid] : 0, p = has ? pos[tid] : 0;
  const uint32_t* ubits = bitmap ? bitmap + user * words : nullptr;
  for (int64_t i = tid; i < n_items; i += UNIQ_THREADS) lown[i] = ~0ull;
  int64_t c[PRE];
  uint32_t w[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    c[k] = draw_item(ph, offset, tid, k, n_items);
    w[k] = (has && ubits) ? ubits[c[k] >> 5] : 0u;
  }
  __syncthreads();
  int64_t mine = -1;
  bool all_done = false;
  auto round = [&](int t, int64_t cc, bool ok) {
    const unsigned long long key = ((unsigned long long)t << 32) | (unsigned long long)tid;
    const bool open_row = has && mine < 0;
    if (open_row && ok) atomicMin(lown + cc, key);
    __syncthreads();
    int open = 0;
    if (open_row) {
      if (ok && lown[cc] == key) mine = cc; else open = 1;
    }
    all_done = !__syncthreads_or(open);
  };
#pragma unroll
  for (int t = 0; t < PRE; ++t)
    if (!all_done) round(t, c[t], c[t] != p && !((w[t] >> (c[t] & 31)) & 1u));
  for (int t = PRE; t < MAX_TRIES && !all_done; ++t) {
    const int64_t cc = draw_item(ph, offset, tid, t, n_items);
    round(t, cc, has && mine < 0 && cc != p && !rated(ubits, cc));
  }
  if (has) neg[tid] = mine;
  if (!all_done) {                                         // tries exhausted for some rows: thread 0 serves them in row order
    __syncthreads();
    if (tid == 0) {
      for (int64_t b = 0; b < n; ++b) {
        if (neg[b] >= 0) continue;
        const uint32_t* bb = bitmap ? bitmap + u[b] * words : nullptr;
        const int64_t pb = pos[b], s0 = draw_item(ph, offset, b, MAX_TRIES - 1, n_items);
        int64_t pick = -1;
        for (int64_t k = 0; k < n_items && pick < 0; ++k) {
          const int64_t cand = s0 + k < n_items ? s0 + k : s0 + k - n_items;
          if (cand == pb || rated(bb, cand) || lown[cand] != ~0ull) continue;
          pick = cand;
        }
        if (pick >= 0) {
          lown[pick] = (unsigned long long)b;
        } else {
          if (fail) atomicAdd(fail, 1);
          pick = pb + 1 < n_items ? pb + 1 : 0;
        }
        neg[b] = pick;
      }
    }
  }
}

__global__ __launch_bounds__(UNIQ_THREADS) void negsample_rec_unique_kernel(const int64_t* __restrict__ u, const int64_t* __restrict__ pos,
                                                                            int64_t n, int64_t n_items,
                                                                            const uint32_t* __restrict__ bitmap, int64_t words,
                                                                            uint64_t seed, uint64_t offset, int64_t* neg,
                                                                            unsigned long long* owner, int32_t* __restrict__ fail) {
  extern __shared__ unsigned long long lown[];
  const Philox ph(seed);
  if (n <= UNIQ_THREADS && n_items <= FAST_ITEMS) rec_unique_fast(ph, u, pos, n, n_items, bitmap, words, offset, neg, lown, fail);
  else rec_unique_rounds(ph, u, pos, n, n_items, bitmap, words, offset, neg, owner, fail, true);
}

KTUP_DEV bool known(const uint64_t* __restrict__ keys, int64_t nk, uint64_t key) {
  int64_t lo = 0, hi = nk;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo < nk && keys[lo] == key;
}

// one triple: fair coin -> corrupt head or tail.  `is_known(key)`: membership in the sorted key list
template <typename Known>
KTUP_DEV void kg_pick(const Philox& ph, uint64_t offset, int64_t b, int64_t hh, int64_t tt, int64_t rr, int64_t n_ent, int64_t n_rel,
                      bool filter, Known is_known, int32_t* __restrict__ fail, int64_t& out_h, int64_t& out_t) {
  const uint64_t base = offset + (uint64_t)b * MAX_TRIES;
  const bool corrupt_head = (draw(ph, base >> 2, (int)(base & 3)) & 0x80000000u) != 0;   // fair coin (data.py:13)
  const int64_t orig = corrupt_head ? hh : tt;
  auto admissible = [&](int64_t c) {
    if (c == orig) return false;
    if (!filter) return true;
    const uint64_t key = corrupt_head ? ((uint64_t)c * n_rel + rr) * n_ent + tt : ((uint64_t)hh * n_rel + rr) * n_ent + c;
    return !is_known(key);
  };
  int64_t pick = -1, last = 0;
  for (int tries = 1; tries < MAX_TRIES && pick < 0; ++tries) {
    last = draw_item(ph, offset, b, tries, n_ent);
    if (admissible(last)) pick = last;
  }
  for (int64_t k = 0; k < n_ent && pick < 0; ++k) {        // tries exhausted: deterministic scan from the last draw
    const int64_t c = last + k < n_ent ? last + k : last + k - n_ent;
    if (admissible(c)) pick = c;
  }
  if (pick < 0) {
    if (fail) atomicAdd(fail, 1);
    pick = orig + 1 < n_ent ? orig + 1 : 0;
  }
  out_h = corrupt_head ? pick : hh;
  out_t = corrupt_head ? tt : pick;
}

__global__ __launch_bounds__(256) void negsample_kg_kernel(const int64_t* __restrict__ h, const int64_t* __restrict__ t,
                                                           const int64_t* __restrict__ r, int64_t n, int64_t n_ent, int64_t n_rel,
                                                           const uint64_t* __restrict__ keys, int64_t nk, uint64_t seed,
                                                           uint64_t offset, int64_t* __restrict__ nh, int64_t* __restrict__ nt,
                                                           int32_t* __restrict__ fail) {
  const Philox ph(seed);
  auto is_known = [&](uint64_t key) { return known(keys, nk, key); };
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n; b += (int64_t)gridDim.x * 256)
    kg_pick(ph, offset, b, h[b], t[b], r[b], n_ent, n_rel, keys != nullptr, is_known, fail, nh[b], nt[b]);
}

// ---- feed kernels: batch + negatives + the steppers' [pos ; neg] id layout in ONE launch whose every argument is static, so a
// training step replays from a HIP graph with no host work between steps (the reference builds each batch in python:
// utils/data.py:87-110 MakeTrainIterator + :64-85 / :12-56).  `cols`: this epoch's shuffled example columns (the host
// reshuffles them in place once per epoch); *cursor: first row of the next batch; *offset_dev: the Philox counter the
// host-driven samplers take as an argument -- the same (seed, offset) sequence, hence the same negatives.  One workgroup.
__global__ __launch_bounds__(UNIQ_THREADS) void feed_rec_kernel(const int64_t* __restrict__ col_u, const int64_t* __restrict__ col_i,
                                                                int64_t n_rows, int64_t B, int64_t* cursor, uint64_t* offset_dev,
                                                                int64_t n_items, const uint32_t* __restrict__ bitmap, int64_t words,
                                                                uint64_t seed, int unique, int64_t* u2, int64_t* i2,
                                                                unsigned long long* owner, int32_t* __restrict__ fail) {
  const Philox ph(seed);
  int64_t start = *cursor;
  const uint64_t offset = *offset_dev;
  if (start < 0 || start + B > n_rows) {                   // the host wraps before this can happen; never read out of bounds
    if (threadIdx.x == 0 && fail) atomicAdd(fail, 1);
    start = 0;
  }
  __syncthreads();                                         // everyone holds the cursor before thread 0 moves it
  const int64_t *u = col_u + start, *pos = col_i + start;
  int64_t* neg = i2 + B;
  extern __shared__ unsigned long long lown[];
  if (unique && B <= UNIQ_THREADS && n_items <= FAST_ITEMS) {
    rec_unique_fast(ph, u, pos, B, n_items, bitmap, words, offset, neg, lown, fail);
    __syncthreads();                                       // (the fallback's thread 0 may still have been writing neg[])
  } else if (unique) {
    rec_unique_rounds(ph, u, pos, B, n_items, bitmap, words, offset, neg, owner, fail, true);
  } else {
    for (int64_t b = threadIdx.x; b < B; b += UNIQ_THREADS) neg[b] = rec_pick(ph, offset, b, u[b], pos[b], n_items, bitmap, words, fail);
  }
  for (int64_t b = threadIdx.x; b < B; b += UNIQ_THREADS) {
    const int64_t uu = u[b];
    u2[b] = uu; u2[b + B] = uu; i2[b] = pos[b];
  }
  if (threadIdx.x == 0) { *cursor = start + B; *offset_dev = offset + (uint64_t)B * MAX_TRIES; }
}

__global__ __launch_bounds__(UNIQ_THREADS) void feed_kg_kernel(const int64_t* __restrict__ col_h, const int64_t* __restrict__ col_t,
                                                               const int64_t* __restrict__ col_r, int64_t n_rows, int64_t B,
                                                               int64_t* cursor, uint64_t* offset_dev, int64_t n_ent, int64_t n_rel,
                                                               const uint64_t* __restrict__ keys, int64_t nk, uint64_t seed,
                                                               int64_t* h2, int64_t* t2, int64_t* r2, int32_t* __restrict__ fail) {
  // every COARSE_STRIDE-th key in LDS: a membership test is a binary search in LDS + log2(stride) probes of the global list
  // instead of ~16 dependent L2 round trips (which were the whole 9 us of this launch)
  __shared__ uint64_t coarse[KG_COARSE];
  const int64_t stride = nk > 0 ? max((int64_t)16, (nk + KG_COARSE - 1) / KG_COARSE) : 1;
  const int64_t nc = nk > 0 ? (nk + stride - 1) / stride : 0;
  for (int64_t i = threadIdx.x; i < nc; i += UNIQ_THREADS) coarse[i] = keys[i * stride];
  const Philox ph(seed);
  int64_t start = *cursor;
  const uint64_t offset = *offset_dev;
  if (start < 0 || start + B > n_rows) {
    if (threadIdx.x == 0 && fail) atomicAdd(fail, 1);
    start = 0;
  }
  __syncthreads();
  auto is_known = [&](uint64_t key) {
    int64_t lo = 0, hi = nc;                               // first coarse entry > key
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (coarse[mid] <= key) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) return false;                             // below the smallest key
    int64_t a = (lo - 1) * stride, z = min(nk, lo * stride);
    while (a < z) {
      const int64_t mid = (a + z) >> 1;
      if (keys[mid] < key) a = mid + 1; else z = mid;
    }
    return a < nk && keys[a] == key;
  };
  for (int64_t b = threadIdx.x; b < B; b += UNIQ_THREADS) {
    const int64_t hh = col_h[start + b], tt = col_t[start + b], rr = col_r[start + b];
    int64_t nh, nt;
    kg_pick(ph, offset, b, hh, tt, rr, n_ent, n_rel, keys != nullptr, is_known, fail, nh, nt);
    h2[b] = hh; t2[b] = tt; r2[b] = rr;
    h2[b + B] = nh; t2[b + B] = nt; r2[b + B] = rr;
  }
  if (threadIdx.x == 0) { *cursor = start + B; *offset_dev = offset + (uint64_t)B * MAX_TRIES; }
}

}  // namespace

extern "C" size_t ktup_negsample_rec_workspace_bytes(int64_t n_items) { return (size_t)(n_items > 0 ? n_items : 0) * sizeof(unsigned long long); }

extern "C" int ktup_negsample_rec(const int64_t* u_ids, const int64_t* pos_items, int64_t n, int64_t n_items,
                                  const uint32_t* user_item_bitmap, int64_t words_per_user, uint64_t seed, uint64_t offset,
                                  int unique_in_batch, int64_t* neg_items, void* ws, int32_t* fail_count, void* stream) {
  const char* name = "ktup_negsample_rec";
  KTUP_REQUIRE(n >= 0 && n_items > 1, "%s: bad sizes", name);
  if (n == 0) return KTUP_OK;
  KTUP_REQUIRE(u_ids && pos_items && neg_items, "%s: null pointer argument", name);
  KTUP_REQUIRE(!user_item_bitmap || words_per_user * 32 >= n_items, "%s: bitmap rows too short", name);
  KTUP_REQUIRE(!unique_in_batch || ws, "%s: unique_in_batch needs the workspace", name);
  KTUP_REQUIRE(!unique_in_batch || (reinterpret_cast<uintptr_t>(ws) & 7u) == 0, "%s: workspace must be 8-byte aligned", name);
  KTUP_REQUIRE(n < (1ll << 32), "%s: batch too large", name);
  hipStream_t st = (hipStream_t)stream;
  if (unique_in_batch) {
    if (hipMemsetAsync(ws, 0xff, ktup_negsample_rec_workspace_bytes(n_items), st) != hipSuccess) return check_launch(name);
    const size_t lds = (n <= UNIQ_THREADS && n_items <= FAST_ITEMS) ? (size_t)n_items * sizeof(unsigned long long) : 0;
    hipLaunchKernelGGL(negsample_rec_unique_kernel, dim3(1), dim3(UNIQ_THREADS), lds, st, u_ids, pos_items, n, n_items, user_item_bitmap,
                       words_per_user, seed, offset, neg_items, (unsigned long long*)ws, fail_count);
  } else {
    hipLaunchKernelGGL(negsample_rec_kernel, dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, u_ids, pos_items, n, n_items,
                       user_item_bitmap, words_per_user, seed, offset, neg_items, fail_count);
  }
  return check_launch(name);
}

extern "C" int ktup_negsample_kg(const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int64_t n_ent, int64_t n_rel,
                                 const uint64_t* sorted_keys, int64_t n_keys, uint64_t seed, uint64_t offset, int64_t* neg_h,
                                 int64_t* neg_t, int32_t* fail_count, void* stream) {
  const char* name = "ktup_negsample_kg";
  KTUP_REQUIRE(n >= 0 && n_ent > 1 && n_rel > 0, "%s: bad sizes", name);
  if (n == 0) return KTUP_OK;
  KTUP_REQUIRE(h && t && r && neg_h && neg_t, "%s: null pointer argument", name);
  hipLaunchKernelGGL(negsample_kg_kernel, dim3(grid_for((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, t, r, n, n_ent, n_rel,
                     sorted_keys, n_keys, seed, offset, neg_h, neg_t, fail_count);
  return check_launch(name);
}

extern "C" int ktup_feed_rec(const int64_t* col_u, const int64_t* col_i, int64_t n_rows, int64_t B, int64_t* cursor,
                             uint64_t* offset_dev, int64_t n_items, const uint32_t* user_item_bitmap, int64_t words_per_user,
                             uint64_t seed, int unique_in_batch, int64_t* u2, int64_t* i2, void* ws, int32_t* fail_count,
                             void* stream) {
  const char* name = "ktup_feed_rec";
  KTUP_REQUIRE(B > 0 && n_rows >= B && n_items > 1, "%s: bad sizes", name);
  KTUP_REQUIRE(col_u && col_i && cursor && offset_dev && u2 && i2, "%s: null pointer argument", name);
  KTUP_REQUIRE(!user_item_bitmap || words_per_user * 32 >= n_items, "%s: bitmap rows too short", name);
  KTUP_REQUIRE(!unique_in_batch || (ws && (reinterpret_cast<uintptr_t>(ws) & 7u) == 0), "%s: unique_in_batch needs the 8-byte aligned workspace",
               name);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (unique_in_batch && B <= UNIQ_THREADS && n_items <= FAST_ITEMS) ? (size_t)n_items * sizeof(unsigned long long) : 0;
  hipLaunchKernelGGL(feed_rec_kernel, dim3(1), dim3(UNIQ_THREADS), lds, st, col_u, col_i, n_rows, B, cursor, offset_dev, n_items,
                     user_item_bitmap, words_per_user, seed, unique_in_batch, u2, i2, (unsigned long long*)ws, fail_count);
  return check_launch(name);
}

extern "C" int ktup_feed_kg(const int64_t* col_h, const int64_t* col_t, const int64_t* col_r, int64_t n_rows, int64_t B,
                            int64_t* cursor, uint64_t* offset_dev, int64_t n_ent, int64_t n_rel, const uint64_t* sorted_keys,
                            int64_t n_keys, uint64_t seed, int64_t* h2, int64_t* t2, int64_t* r2, int32_t* fail_count, void* stream) {
  const char* name = "ktup_feed_kg";
  KTUP_REQUIRE(B > 0 && n_rows >= B && n_ent > 1 && n_rel > 0, "%s: bad sizes", name);
  KTUP_REQUIRE(col_h && col_t && col_r && cursor && offset_dev && h2 && t2 && r2, "%s: null pointer argument", name);
  hipLaunchKernelGGL(feed_kg_kernel, dim3(1), dim3(UNIQ_THREADS), 0, (hipStream_t)stream, col_h, col_t, col_r, n_rows, B, cursor,
                     offset_dev, n_ent, n_rel, sorted_keys, n_keys, seed, h2, t2, r2, fail_count);
  return check_launch(name);
}
