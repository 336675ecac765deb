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

// no batch-uniqueness: rows are independent
__global__ __launch_bounds__(256) void negsample_rec_kernel(const int64_t* __restrict__ u, const int64_t* __restrict__ pos,
                                                            int64_t n, int64_t n_items, const uint32_t* __restrict__ bitmap,
                                                            int64_t words, uint64_t seed, uint64_t offset,
                                                            int64_t* __restrict__ neg, int32_t* __restrict__ fail) {
  const Philox ph(seed);
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n; b += (int64_t)gridDim.x * 256) {
    const uint32_t* ubits = bitmap ? bitmap + u[b] * words : nullptr;
    const int64_t p = pos[b];
    int64_t pick = -1;
    for (int tries = 0; tries < MAX_TRIES && pick < 0; ++tries) {
      const int64_t c = draw_item(ph, offset, b, tries, n_items);
      if (c != p && !rated(ubits, c)) pick = c;
    }
    if (pick < 0) {                                        // a user who rated (nearly) everything: scan from the last draw
      const int64_t s0 = draw_item(ph, offset, b, MAX_TRIES - 1, n_items);
      for (int64_t k = 0; k < n_items && pick < 0; ++k) {
        const int64_t c = s0 + k < n_items ? s0 + k : s0 + k - n_items;
        if (c != p && !rated(ubits, c)) pick = c;
      }
    }
    if (pick < 0) {                                        // no admissible item exists: in-range stand-in + error count
      if (fail) atomicAdd(fail, 1);
      pick = p + 1 < n_items ? p + 1 : 0;
    }
    neg[b] = pick;
  }
}

// batch-unique negatives, one workgroup, deterministic rounds (see the file comment).  owner[] is all-ones on entry.
__global__ __launch_bounds__(UNIQ_THREADS) void negsample_rec_unique_kernel(const int64_t* __restrict__ u, const int64_t* __restrict__ pos,
                                                                            int64_t n, int64_t n_items,
                                                                            const uint32_t* __restrict__ bitmap, int64_t words,
                                                                            uint64_t seed, uint64_t offset, int64_t* neg,
                                                                            unsigned long long* owner, int32_t* __restrict__ fail) {
  const Philox ph(seed);
  const int tid = threadIdx.x;
  for (int64_t b = tid; b < n; b += UNIQ_THREADS) neg[b] = -1;        // a thread only ever touches its own rows of neg[]
  for (int t = 0; t < MAX_TRIES; ++t) {
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
    if (!__syncthreads_or(open)) return;
  }
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
}

KTUP_DEV bool known(const uint64_t* __restrict__ keys, int64_t nk, uint64_t key) {
  int64_t lo = 0, hi = nk;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo < nk && keys[lo] == key;
}

__global__ __launch_bounds__(256) void negsample_kg_kernel(const int64_t* __restrict__ h, const int64_t* __restrict__ t,
                                                           const int64_t* __restrict__ r, int64_t n, int64_t n_ent, int64_t n_rel,
                                                           const uint64_t* __restrict__ keys, int64_t nk, uint64_t seed,
                                                           uint64_t offset, int64_t* __restrict__ nh, int64_t* __restrict__ nt,
                                                           int32_t* __restrict__ fail) {
  const Philox ph(seed);
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n; b += (int64_t)gridDim.x * 256) {
    const int64_t hh = h[b], tt = t[b], rr = r[b];
    const uint64_t base = offset + (uint64_t)b * MAX_TRIES;
    const bool corrupt_head = (draw(ph, base >> 2, (int)(base & 3)) & 0x80000000u) != 0;   // fair coin (data.py:13)
    const int64_t orig = corrupt_head ? hh : tt;
    auto admissible = [&](int64_t c) {
      if (c == orig) return false;
      if (!keys) return true;
      const uint64_t key = corrupt_head ? ((uint64_t)c * n_rel + rr) * n_ent + tt : ((uint64_t)hh * n_rel + rr) * n_ent + c;
      return !known(keys, nk, key);
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
    nh[b] = corrupt_head ? pick : hh;
    nt[b] = corrupt_head ? tt : pick;
  }
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
    hipLaunchKernelGGL(negsample_rec_unique_kernel, dim3(1), dim3(UNIQ_THREADS), 0, st, u_ids, pos_items, n, n_items, user_item_bitmap,
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
