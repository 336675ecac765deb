// K19: negative sampling on the device, with the constraints of jTransUP/utils/data.py:12-85
//   rec (getNegRatings)        : uniform item  != the positive, not rated by the user in ANY split (per-user bitmap),
//                                and not already used as a negative in this batch (claimed with an atomic OR);
//   kg  (getTrainTripleBatch)  : fair coin per triple -> corrupt head or tail with a uniform entity != the original that
//                                does not make a known-true triple (binary search in the sorted 64-bit keys of all splits).
// The reference draws from python's Mersenne Twister on the host, once per row, inside the training loop; RNG parity
// is not achievable, so parity here means the same constraints and the same distribution (Philox4x32-10 counters).
#include "ktup_common.h"

using namespace ktup;

namespace {

constexpr int MAX_TRIES = 4096;

KTUP_DEV uint32_t draw(const Philox& ph, uint64_t ctr, int which) {
  const uint4 r = ph(ctr, 0x4e454753ull /* "NEGS" */);
  return which == 0 ? r.x : which == 1 ? r.y : which == 2 ? r.z : r.w;
}
KTUP_DEV int64_t bounded(uint32_t x, int64_t n) { return (int64_t)(((uint64_t)x * (uint64_t)n) >> 32); }

__global__ __launch_bounds__(256) void negsample_rec_kernel(const int64_t* __restrict__ u, const int64_t* __restrict__ pos,
                                                            int64_t n, int64_t n_items, const uint32_t* __restrict__ bitmap,
                                                            int64_t words, uint64_t seed, uint64_t offset, int unique,
                                                            int64_t* __restrict__ neg, uint32_t* __restrict__ batch_bits) {
  const Philox ph(seed);
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n; b += (int64_t)gridDim.x * 256) {
    const uint32_t* ubits = bitmap ? bitmap + u[b] * words : nullptr;
    const int64_t p = pos[b];
    int64_t pick = -1;
    for (int tries = 0; tries < MAX_TRIES && pick < 0; ++tries) {
      const uint64_t ctr = offset + (uint64_t)b * MAX_TRIES + tries;
      const int64_t c = bounded(draw(ph, ctr >> 2, (int)(ctr & 3)), n_items);
      if (c == p) continue;
      if (ubits && ((ubits[c >> 5] >> (c & 31)) & 1u)) continue;
      if (unique) {
        const uint32_t bit = 1u << (c & 31);
        if (atomicOr(batch_bits + (c >> 5), bit) & bit) continue;   // someone in this batch already took it
      }
      pick = c;
    }
    neg[b] = pick;   // -1: the constraints could not be met (more rows than admissible items); the host raises
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
                                                           uint64_t offset, int64_t* __restrict__ nh, int64_t* __restrict__ nt) {
  const Philox ph(seed);
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n; b += (int64_t)gridDim.x * 256) {
    const int64_t hh = h[b], tt = t[b], rr = r[b];
    const uint64_t base = offset + (uint64_t)b * MAX_TRIES;
    const bool corrupt_head = (draw(ph, base >> 2, (int)(base & 3)) & 0x80000000u) != 0;   // fair coin (data.py:13)
    int64_t pick = -1;
    for (int tries = 1; tries < MAX_TRIES && pick < 0; ++tries) {
      const uint64_t ctr = base + tries;
      const int64_t c = bounded(draw(ph, ctr >> 2, (int)(ctr & 3)), n_ent);
      if (c == (corrupt_head ? hh : tt)) continue;
      if (keys) {
        const uint64_t key = corrupt_head ? ((uint64_t)c * n_rel + rr) * n_ent + tt : ((uint64_t)hh * n_rel + rr) * n_ent + c;
        if (known(keys, nk, key)) continue;
      }
      pick = c;
    }
    nh[b] = corrupt_head ? pick : hh;
    nt[b] = corrupt_head ? tt : pick;
  }
}

}  // namespace

extern "C" size_t ktup_negsample_rec_workspace_bytes(int64_t n_items) { return (size_t)((n_items + 31) / 32) * sizeof(uint32_t); }

extern "C" int ktup_negsample_rec(const int64_t* u_ids, const int64_t* pos_items, int64_t n, int64_t n_items,
                                  const uint32_t* user_item_bitmap, int64_t words_per_user, uint64_t seed, uint64_t offset,
                                  int unique_in_batch, int64_t* neg_items, void* ws, void* stream) {
  const char* name = "ktup_negsample_rec";
  KTUP_REQUIRE(n >= 0 && n_items > 1, "%s: bad sizes", name);
  if (n == 0) return KTUP_OK;
  KTUP_REQUIRE(u_ids && pos_items && neg_items, "%s: null pointer argument", name);
  KTUP_REQUIRE(!user_item_bitmap || words_per_user * 32 >= n_items, "%s: bitmap rows too short", name);
  KTUP_REQUIRE(!unique_in_batch || ws, "%s: unique_in_batch needs the workspace", name);
  hipStream_t st = (hipStream_t)stream;
  if (unique_in_batch && hipMemsetAsync(ws, 0, ktup_negsample_rec_workspace_bytes(n_items), st) != hipSuccess) return check_launch(name);
  hipLaunchKernelGGL(negsample_rec_kernel, dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, u_ids, pos_items, n, n_items,
                     user_item_bitmap, words_per_user, seed, offset, unique_in_batch, neg_items, (uint32_t*)ws);
  return check_launch(name);
}

extern "C" int ktup_negsample_kg(const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int64_t n_ent, int64_t n_rel,
                                 const uint64_t* sorted_keys, int64_t n_keys, uint64_t seed, uint64_t offset, int64_t* neg_h,
                                 int64_t* neg_t, void* stream) {
  const char* name = "ktup_negsample_kg";
  KTUP_REQUIRE(n >= 0 && n_ent > 1 && n_rel > 0, "%s: bad sizes", name);
  if (n == 0) return KTUP_OK;
  KTUP_REQUIRE(h && t && r && neg_h && neg_t, "%s: null pointer argument", name);
  hipLaunchKernelGGL(negsample_kg_kernel, dim3(grid_for((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, t, r, n, n_ent, n_rel,
                     sorted_keys, n_keys, seed, offset, neg_h, neg_t);
  return check_launch(name);
}
