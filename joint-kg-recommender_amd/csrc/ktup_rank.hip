// K17 / K18: the ranking walk of evaluation on the device.
//   reference: jTransUP/utils/misc.py:213-248 (getRecPerformance) and :125-146 (getKGPerformance), which
//   np.argsort the full score row on the host (after a 6.6 MB D2H copy per batch, pickled to worker processes)
//   and then walk it in python skipping filtered ids.
// Integer work, bit-exact by construction: every candidate gets the 64-bit key
//        key = (order-preserving image of the fp32 score) << 32 | candidate id
// so the order is total: ascending score, ties -> lower id first (the declared tie rule; np.argsort's default sort
// is not stable, so the reference leaves ties unspecified).  `descending` negates the score first, exactly like
// `per_scores = -pred` (misc.py:93,180).  Filtered candidates get key = 2^64-1 and never rank.
// One workgroup per query row; the row's keys live in LDS (8 B per candidate: 26 KB for ml1m's 3240 items,
// 118 KB for its 14709 entities), so the score row is read from HBM exactly once.
// Catalogues beyond 19,000 candidates (amazon-book / last-fm / yelp2018 entities, config 5's 1 M items) take the
// chunked kernels below: the row streams through LDS in chunks of 16 K keys, still one read of the score row for
// top-n and one per batch of 1024 golds for the ranks, no workspace.
#include <cstdlib>

#include "ktup_common.h"

using namespace ktup;

namespace {

constexpr uint64_t KEY_MAX = ~0ull;
constexpr int64_t MAX_LDS_CAND = 19000;  // 152 KB of keys + reduction scratch within the 160 KB LDS

KTUP_DEV uint64_t make_key(float s, bool descending, uint32_t id) {
  if (descending) s = -s;
  if (s == 0.f) s = 0.f;  // -0.0 and +0.0 compare equal in the reference's sort: one key for both
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | id;
}

KTUP_DEV uint64_t wave_min64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const uint64_t o = __shfl_xor(v, m, 64);
    v = o < v ? o : v;
  }
  return v;
}

KTUP_DEV int wave_sum_int(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

KTUP_DEV void load_keys(uint64_t* keys, const float* row, int64_t n_cand, bool descending, const int32_t* fids, int64_t nf) {
  for (int64_t j = threadIdx.x; j < n_cand; j += 256) keys[j] = make_key(row[j], descending, (uint32_t)j);
  __syncthreads();
  for (int64_t f = threadIdx.x; f < nf; f += 256) {
    const int32_t id = fids[f];
    if (id >= 0 && id < n_cand) keys[id] = KEY_MAX;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void topk_filtered_kernel(const float* __restrict__ scores, int64_t lds, int64_t n_cand,
                                                            int descending, const int64_t* __restrict__ filt_off,
                                                            const int32_t* __restrict__ filt_ids, int topn,
                                                            int32_t* __restrict__ top_ids, float* __restrict__ top_scores) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  __shared__ uint64_t red[4];
  const int64_t b = blockIdx.x;
  const float* row = scores + b * lds;
  const int64_t f0 = filt_off ? filt_off[b] : 0, f1 = filt_off ? filt_off[b + 1] : 0;
  load_keys(keys, row, n_cand, descending != 0, filt_ids + f0, f1 - f0);
  uint64_t prev = 0;
  bool first = true;
  for (int r = 0; r < topn; ++r) {
    uint64_t best = KEY_MAX;
    for (int64_t j = threadIdx.x; j < n_cand; j += 256) {
      const uint64_t k = keys[j];
      if ((first || k > prev) && k < best) best = k;
    }
    best = wave_min64(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    best = min(min(red[0], red[1]), min(red[2], red[3]));
    __syncthreads();
    if (threadIdx.x == 0) {
      const bool ok = best != KEY_MAX;
      const int32_t id = ok ? (int32_t)(uint32_t)best : -1;
      top_ids[b * topn + r] = id;
      if (top_scores) top_scores[b * topn + r] = ok ? row[id] : 0.f;
    }
    prev = best;
    first = false;
    if (best == KEY_MAX) {  // fewer than topn unfiltered candidates: pad the rest (uniform branch)
      for (int rr = r + 1 + threadIdx.x; rr < topn; rr += 256) {
        top_ids[b * topn + rr] = -1;
        if (top_scores) top_scores[b * topn + rr] = 0.f;
      }
      break;
    }
  }
}

// rank of gold g = #{unfiltered, non-gold candidates ordered before g}  (0-based; other golds do not advance the rank,
// misc.py:134-144).  A gold id that is itself filtered is never reached by the reference's walk: rank -1.
// No key array: a workgroup keeps two BITS per candidate in LDS (filtered; filtered or gold: not counted) and streams the score row
// from global memory, comparing every candidate against up to GOLD_B gold keys held in registers.  3.6 KB of LDS at 14,709 entities,
// so every query of a 512-query batch is resident at once (a 117 KB key array allowed one workgroup per CU: two rounds).
constexpr int GOLD_B = 8;
constexpr int STREAM_U = 8;                            // independent row loads in flight per thread
constexpr int64_t STREAM_MAX_CAND = 64 * 1024 * 8;     // two bitmaps of <= 64 KB

__global__ __launch_bounds__(256) void gold_ranks_stream_kernel(const float* __restrict__ scores, int64_t lds, int64_t n_cand,
                                                                int descending, const int64_t* __restrict__ filt_off,
                                                                const int32_t* __restrict__ filt_ids,
                                                                const int64_t* __restrict__ gold_off,
                                                                const int32_t* __restrict__ gold_ids, int32_t* __restrict__ ranks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int words = (int)((n_cand + 31) / 32);
  uint32_t* filt = reinterpret_cast<uint32_t*>(smem);                     // [words]  filtered ids
  uint32_t* skip = filt + words;                                          // [words]  filtered or gold: never counted
  __shared__ int red[4][GOLD_B];
  __shared__ uint64_t gkeys[GOLD_B];
  const int64_t b = blockIdx.x;
  const float* row = scores + b * lds;
  const bool desc = descending != 0;
  for (int i = threadIdx.x; i < words; i += 256) filt[i] = 0u;
  __syncthreads();
  const int64_t f0 = filt_off ? filt_off[b] : 0, f1 = filt_off ? filt_off[b + 1] : 0;
  for (int64_t f = f0 + threadIdx.x; f < f1; f += 256) {
    const int32_t id = filt_ids[f];
    if (id >= 0 && id < n_cand) atomicOr(filt + (id >> 5), 1u << (id & 31));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < words; i += 256) skip[i] = filt[i];
  __syncthreads();
  const int64_t g0 = gold_off[b], g1 = gold_off[b + 1];
  for (int64_t o = g0 + threadIdx.x; o < g1; o += 256) {
    const int32_t og = gold_ids[o];
    if (og >= 0 && og < n_cand) atomicOr(skip + (og >> 5), 1u << (og & 31));
  }
  __syncthreads();
  for (int64_t gb = g0; gb < g1; gb += GOLD_B) {
    if (threadIdx.x < GOLD_B) {
      const int64_t gi = gb + threadIdx.x;
      uint64_t k = KEY_MAX;
      if (gi < g1) {
        const int32_t g = gold_ids[gi];
        if (g >= 0 && g < n_cand && !((filt[g >> 5] >> (g & 31)) & 1u)) k = make_key(row[g], desc, (uint32_t)g);
      }
      gkeys[threadIdx.x] = k;
    }
    __syncthreads();
    uint64_t gk[GOLD_B];
    int cnt[GOLD_B];
#pragma unroll
    for (int i = 0; i < GOLD_B; ++i) { gk[i] = gkeys[i]; cnt[i] = 0; }
    for (int64_t j0 = threadIdx.x; j0 < n_cand; j0 += 256 * STREAM_U) {
      float v[STREAM_U];
#pragma unroll
      for (int u = 0; u < STREAM_U; ++u) {
        const int64_t j = j0 + 256 * u;
        v[u] = row[j < n_cand ? j : 0];
      }
#pragma unroll
      for (int u = 0; u < STREAM_U; ++u) {
        const int64_t j = j0 + 256 * u;
        const bool on = j < n_cand && !((skip[(j < n_cand ? j : 0) >> 5] >> (j & 31)) & 1u);
        const uint64_t k = on ? make_key(v[u], desc, (uint32_t)j) : KEY_MAX;     // KEY_MAX is below no gold key
#pragma unroll
        for (int i = 0; i < GOLD_B; ++i) cnt[i] += k < gk[i] ? 1 : 0;
      }
    }
#pragma unroll
    for (int i = 0; i < GOLD_B; ++i) {
      const int c = wave_sum_int(cnt[i]);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = c;
    }
    __syncthreads();
    if (threadIdx.x < GOLD_B && gb + threadIdx.x < g1)
      ranks[gb + threadIdx.x] = gkeys[threadIdx.x] == KEY_MAX ? -1 : red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- chunked (any n_cand)
KTUP_DEV uint64_t block_min64(uint64_t v, uint64_t* red) {
  v = wave_min64(v);
  __syncthreads();  // red[] may still be read from the previous round
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return min(min(red[0], red[1]), min(red[2], red[3]));
}

// keys[0, len) <- keys of candidates [c0, c0 + len), anything >= `limit` or filtered becomes KEY_MAX.
// Returns (block-uniform) whether any candidate was below `limit` before the filter was applied.
KTUP_DEV bool load_chunk(uint64_t* keys, const float* row, int64_t c0, int len, bool descending, uint64_t limit,
                         const int32_t* fids, int64_t nf) {
  int any = 0;
  for (int j = threadIdx.x; j < len; j += 256) {
    uint64_t k = make_key(row[c0 + j], descending, (uint32_t)(c0 + j));
    if (k >= limit) k = KEY_MAX;
    any |= k != KEY_MAX;
    keys[j] = k;
  }
  if (!__syncthreads_or(any)) return false;
  for (int64_t f = threadIdx.x; f < nf; f += 256) {
    const int64_t id = (int64_t)fids[f] - c0;
    if (id >= 0 && id < len) keys[id] = KEY_MAX;
  }
  __syncthreads();
  return true;
}

// top-n over a streamed row: slots [CH, CH + topn) of the key array hold the running list; a chunk that has a
// candidate below the list's current worst key is merged by extracting the topn smallest of chunk + list.
__global__ __launch_bounds__(256) void topk_chunked_kernel(const float* __restrict__ scores, int64_t lds, int64_t n_cand,
                                                           int descending, const int64_t* __restrict__ filt_off,
                                                           const int32_t* __restrict__ filt_ids, int topn, int CH,
                                                           int32_t* __restrict__ top_ids, float* __restrict__ top_scores) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  uint64_t* run = keys + CH;          // [topn] sorted ascending, KEY_MAX padded
  uint64_t* merged = run + topn;      // [topn]
  __shared__ uint64_t red[4];
  const int64_t b = blockIdx.x;
  const float* row = scores + b * lds;
  const int64_t f0 = filt_off ? filt_off[b] : 0, f1 = filt_off ? filt_off[b + 1] : 0;
  for (int r = threadIdx.x; r < topn; r += 256) run[r] = KEY_MAX;
  __syncthreads();
  for (int64_t c0 = 0; c0 < n_cand; c0 += CH) {
    const int len = (int)min((int64_t)CH, n_cand - c0);
    const uint64_t limit = run[topn - 1];
    __syncthreads();                  // everyone has read `limit` before the list is rewritten below
    if (!load_chunk(keys, row, c0, len, descending != 0, limit, filt_ids + f0, f1 - f0)) continue;
    for (int j = len + threadIdx.x; j < CH; j += 256) keys[j] = KEY_MAX;   // the tail chunk: neutral keys up to the list
    __syncthreads();
    uint64_t prev = 0;
    for (int r = 0; r < topn; ++r) {
      uint64_t best = KEY_MAX;
      for (int j = threadIdx.x; j < CH + topn; j += 256) {
        const uint64_t k = keys[j];
        if ((r == 0 || k > prev) && k < best) best = k;
      }
      best = block_min64(best, red);
      if (threadIdx.x == 0) merged[r] = best;
      prev = best;
      if (best == KEY_MAX) {          // uniform: fewer than topn unfiltered candidates so far
        for (int rr = r + 1 + threadIdx.x; rr < topn; rr += 256) merged[rr] = KEY_MAX;
        break;
      }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < topn; r += 256) run[r] = merged[r];
    __syncthreads();
  }
  for (int r = threadIdx.x; r < topn; r += 256) {
    const uint64_t k = run[r];
    const int32_t id = k != KEY_MAX ? (int32_t)(uint32_t)k : -1;
    top_ids[b * topn + r] = id;
    if (top_scores) top_scores[b * topn + r] = id >= 0 ? row[id] : 0.f;
  }
}

constexpr int GOLD_BATCH = 1024;

__global__ __launch_bounds__(256) void gold_ranks_chunked_kernel(const float* __restrict__ scores, int64_t lds, int64_t n_cand,
                                                                 int descending, const int64_t* __restrict__ filt_off,
                                                                 const int32_t* __restrict__ filt_ids,
                                                                 const int64_t* __restrict__ gold_off,
                                                                 const int32_t* __restrict__ gold_ids, int CH,
                                                                 int32_t* __restrict__ ranks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  __shared__ uint64_t gkey[GOLD_BATCH];
  __shared__ int gid[GOLD_BATCH];
  __shared__ int acc[GOLD_BATCH];
  const int64_t b = blockIdx.x;
  const float* row = scores + b * lds;
  const int32_t* fids = filt_ids + (filt_off ? filt_off[b] : 0);
  const int64_t nf = filt_off ? filt_off[b + 1] - filt_off[b] : 0;
  const int64_t g0 = gold_off[b], g1 = gold_off[b + 1];
  for (int64_t gb = g0; gb < g1; gb += GOLD_BATCH) {
    const int ng = (int)min((int64_t)GOLD_BATCH, g1 - gb);
    __syncthreads();
    for (int i = threadIdx.x; i < ng; i += 256) {
      const int32_t g = gold_ids[gb + i];
      const bool ok = g >= 0 && g < n_cand;
      gid[i] = ok ? g : -1;
      gkey[i] = ok ? make_key(row[g], descending != 0, (uint32_t)g) : KEY_MAX;
      acc[i] = 0;
    }
    __syncthreads();
    for (int64_t f = threadIdx.x; f < nf; f += 256) {   // a gold that is itself filtered is never reached: rank -1
      const int32_t id = fids[f];
      for (int i = 0; i < ng; ++i)
        if (gid[i] == id) gkey[i] = KEY_MAX;
    }
    __syncthreads();
    for (int64_t c0 = 0; c0 < n_cand; c0 += CH) {
      const int len = (int)min((int64_t)CH, n_cand - c0);
      __syncthreads();
      (void)load_chunk(keys, row, c0, len, descending != 0, KEY_MAX, fids, nf);
      for (int i = 0; i < ng; ++i) {
        const uint64_t gk = gkey[i];
        if (gk == KEY_MAX) continue;  // uniform (LDS broadcast)
        int cnt = 0;
        for (int j = threadIdx.x; j < len; j += 256) cnt += keys[j] < gk ? 1 : 0;
        for (int64_t o = g0 + threadIdx.x; o < g1; o += 256) {   // other golds do not advance the rank
          const int64_t og = (int64_t)gold_ids[o] - c0;
          if (og >= 0 && og < len && keys[og] < gk) cnt -= 1;
        }
        cnt = wave_sum_int(cnt);
        if ((threadIdx.x & 63) == 0 && cnt != 0) atomicAdd(&acc[i], cnt);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ng; i += 256) ranks[gb + i] = gkey[i] == KEY_MAX ? -1 : acc[i];
  }
}

// K18b  per-query recommendation metrics (utils/misc.py:232-248, utils/evaluation.py:80-110): hits of the ranked ids in
// the gold set, precision = hits / len(list), recall = hits / |gold|, F1, hit flag and NDCG with method-0 weights
// (1, 1, 1/log2(3), ...) whose ideal is the best ordering of the OBSERVED hits.  One thread per query, float64 like the
// reference; the gold ids of a query are ascending (binary search).
// Sixteen lanes per query (topn <= 16: lane r looks entry r up, the four dependent loads of its binary search beside the others');
// the float64 sums are then formed by the group's first lane in list order, as one thread per query formed them (that form was a chain of
// ~40 dependent loads per query: 10 us for the 6040 users of an ml1m pass); longer lists keep the one-thread loop.
__global__ __launch_bounds__(64) void rec_metrics_kernel(const int32_t* __restrict__ top_ids, int64_t nq, int topn,
                                                         const int64_t* __restrict__ gold_off,
                                                         const int32_t* __restrict__ gold_ids, double* __restrict__ out) {
  auto lookup = [&](int32_t id, int64_t g0, int64_t g1) {
    int64_t lo = g0, hi = g1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (gold_ids[mid] < id) lo = mid + 1; else hi = mid;
    }
    return lo < g1 && gold_ids[lo] == id;
  };
  auto finish = [&](int64_t b, int k, int hc, double dcg, int64_t ngold) {
    double f1 = 0.0, p = 0.0, rc = 0.0, ndcg = 0.0;
    if (hc > 0) {
      p = (double)hc / (double)k;
      rc = (double)hc / (double)ngold;
      f1 = 2.0 * p * rc / (p + rc);
      double ideal = 1.0;                         // hc hits in the first hc positions
      for (int c = 1; c < hc; ++c) ideal += 1.0 / log2((double)(c + 1));
      ndcg = dcg / ideal;
    }
    double* o = out + b * 5;
    o[0] = f1; o[1] = p; o[2] = rc; o[3] = hc > 0 ? 1.0 : 0.0; o[4] = ndcg;
  };
  if (topn <= 16) {
    const int lane = threadIdx.x, r = lane & 15, grp = lane >> 4;
    const int64_t b = (int64_t)blockIdx.x * 4 + grp;
    const bool on = b < nq;
    const int64_t g0 = on ? gold_off[b] : 0, g1 = on ? gold_off[b + 1] : 0;
    const int32_t id = (on && r < topn) ? top_ids[b * topn + r] : -1;
    const bool valid = id >= 0;                    // -1 padding: the reference's list is simply shorter
    const bool hit = valid && lookup(id, g0, g1);
    const uint32_t vmask = (uint32_t)(__builtin_amdgcn_ballot_w64(valid) >> (16 * grp)) & 0xffffu;
    uint32_t hmask = (uint32_t)(__builtin_amdgcn_ballot_w64(hit) >> (16 * grp)) & 0xffffu;
    if (on && r == 0) {
      int hc = 0;
      double dcg = 0.0;
      while (hmask) {                              // hits in list order; position = valid entries before it
        const int pos = __builtin_ctz(hmask);
        hmask &= hmask - 1;
        const int k = __popc(vmask & ((1u << pos) - 1u));
        hc += 1;
        dcg += k == 0 ? 1.0 : 1.0 / log2((double)(k + 1));
      }
      finish(b, __popc(vmask), hc, dcg, g1 - g0);
    }
    return;
  }
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= nq) return;
  const int64_t g0 = gold_off[b], g1 = gold_off[b + 1];
  int k = 0, hc = 0;
  double dcg = 0.0;
  for (int r = 0; r < topn; ++r) {
    const int32_t id = top_ids[b * topn + r];
    if (id < 0) continue;
    // position in the reference's (shorter) list = number of valid entries before it; padding only ever trails
    if (lookup(id, g0, g1)) {
      hc += 1;
      dcg += k == 0 ? 1.0 : 1.0 / log2((double)(k + 1));
    }
    k += 1;
  }
  finish(b, k, hc, dcg, g1 - g0);
}

constexpr int CHUNK_KEYS = 16384;  // 128 KB of keys per chunk

// 0 = take the single-workgroup path; otherwise the chunk size.  Option rank_chunk = <keys> forces the chunked path (tests).
int chunk_for(int64_t n_cand) {
  {
    const int v = ktup::opt_rank_chunk();
    if (v > 0) return min(max(v, 64), CHUNK_KEYS);
  }
  return n_cand > MAX_LDS_CAND ? CHUNK_KEYS : 0;
}

void allow_lds(const void* fn, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int prep_lds(const void* fn, int64_t n_cand, size_t* lds, const char* name) {
  if (n_cand > MAX_LDS_CAND)
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: %lld candidates exceed the single-workgroup LDS ranking path (max %lld)", name,
                     (long long)n_cand, (long long)MAX_LDS_CAND);
  *lds = (size_t)n_cand * 8;
  if (*lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)*lds);
  return KTUP_OK;
}

}  // namespace

extern "C" int ktup_eval_topk_filtered(const float* scores, int64_t lds, int64_t nq, int64_t n_cand, int descending,
                                       const int64_t* filt_off, const int32_t* filt_ids, int topn, int32_t* top_ids,
                                       float* top_scores, void* stream) {
  const char* name = "ktup_eval_topk_filtered";
  KTUP_REQUIRE(nq >= 0 && n_cand > 0 && topn > 0 && lds >= n_cand, "%s: bad sizes", name);
  if (nq == 0) return KTUP_OK;
  KTUP_REQUIRE(scores && top_ids && ((filt_off == nullptr) || filt_ids), "%s: null pointer argument", name);
  KTUP_REQUIRE(n_cand <= 0x7fffffffll, "%s: candidate ids are 32-bit", name);
  if (const int ch = chunk_for(n_cand)) {
    KTUP_REQUIRE(topn <= 1024, "%s: topn %d > 1024 on the chunked path", name, topn);
    const size_t lbytes = ((size_t)ch + 2 * (size_t)topn) * 8;
    allow_lds((const void*)topk_chunked_kernel, lbytes);
    hipLaunchKernelGGL(topk_chunked_kernel, dim3((unsigned)nq), dim3(256), lbytes, (hipStream_t)stream, scores, lds, n_cand,
                       descending, filt_off, filt_ids, topn, ch, top_ids, top_scores);
    return check_launch(name);
  }
  size_t bytes = 0;
  if (int e = prep_lds((const void*)topk_filtered_kernel, n_cand, &bytes, name)) return e;
  hipLaunchKernelGGL(topk_filtered_kernel, dim3((unsigned)nq), dim3(256), bytes, (hipStream_t)stream, scores, lds, n_cand,
                     descending, filt_off, filt_ids, topn, top_ids, top_scores);
  return check_launch(name);
}

extern "C" int ktup_eval_gold_ranks(const float* scores, int64_t lds, int64_t nq, int64_t n_cand, int descending,
                                    const int64_t* filt_off, const int32_t* filt_ids, const int64_t* gold_off,
                                    const int32_t* gold_ids, int32_t* ranks, void* stream) {
  const char* name = "ktup_eval_gold_ranks";
  KTUP_REQUIRE(nq >= 0 && n_cand > 0 && lds >= n_cand, "%s: bad sizes", name);
  if (nq == 0) return KTUP_OK;
  KTUP_REQUIRE(scores && gold_off && gold_ids && ranks && ((filt_off == nullptr) || filt_ids), "%s: null pointer argument", name);
  KTUP_REQUIRE(n_cand <= 0x7fffffffll, "%s: candidate ids are 32-bit", name);
  if (const int ch = (ktup::opt_rank_chunk() > 0 || n_cand > STREAM_MAX_CAND) ? chunk_for(n_cand) : 0) {
    const size_t lbytes = (size_t)ch * 8;
    allow_lds((const void*)gold_ranks_chunked_kernel, lbytes);
    hipLaunchKernelGGL(gold_ranks_chunked_kernel, dim3((unsigned)nq), dim3(256), lbytes, (hipStream_t)stream, scores, lds, n_cand,
                       descending, filt_off, filt_ids, gold_off, gold_ids, ch, ranks);
    return check_launch(name);
  }
  const size_t lbytes = (size_t)((n_cand + 31) / 32) * 8;
  allow_lds((const void*)gold_ranks_stream_kernel, lbytes);
  hipLaunchKernelGGL(gold_ranks_stream_kernel, dim3((unsigned)nq), dim3(256), lbytes, (hipStream_t)stream, scores, lds, n_cand, descending,
                     filt_off, filt_ids, gold_off, gold_ids, ranks);
  return check_launch(name);
}

// ---------------------------------------------------------------------------------------------- candidate shards
// The 0-based filtered rank of a gold id is a COUNT (misc.py:134-144: candidates walked before it, skipping filtered ids and
// other golds), so it is additive over disjoint candidate shards: this kernel counts, for one shard [c_lo, c_lo + n_local) of
// the catalogue, the unfiltered non-gold candidates ordered before each gold; the all-reduce(sum) of the shards' counts is the
// rank.  The gold's own score comes from the shard that owns it (gold_scores, exchanged by the caller).  Keys use GLOBAL ids, so
// ties break exactly as in the single-device kernels.  A gold that is itself filtered gets FILTERED_GOLD (negative on every
// shard, so the sum stays negative -> rank -1).
constexpr int32_t FILTERED_GOLD = -(1 << 20);

__global__ __launch_bounds__(256) void gold_rank_counts_kernel(const float* __restrict__ scores, int64_t lds, int64_t n_local, int64_t c_lo,
                                                               int64_t c_stride, int descending, const int64_t* __restrict__ filt_off,
                                                               const int32_t* __restrict__ filt_ids,
                                                               const int64_t* __restrict__ gold_off,
                                                               const int32_t* __restrict__ gold_ids,
                                                               const float* __restrict__ gold_scores, int CH,
                                                               int32_t* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  __shared__ int red[4];
  const int64_t b = blockIdx.x;
  const float* row = scores + b * lds;
  const int32_t* fids = filt_ids + (filt_off ? filt_off[b] : 0);
  const int64_t nf = filt_off ? filt_off[b + 1] - filt_off[b] : 0;
  const int64_t g0 = gold_off[b], g1 = gold_off[b + 1];
  for (int64_t gi = g0 + threadIdx.x; gi < g1; gi += 256) {            // filtered golds (every shard sees the whole filter list)
    const int32_t g = gold_ids[gi];
    int32_t v = 0;
    for (int64_t f = 0; f < nf; ++f) if (fids[f] == g) { v = FILTERED_GOLD; break; }
    counts[gi] = v;
  }
  for (int64_t c0 = 0; c0 < n_local; c0 += CH) {
    const int len = (int)min((int64_t)CH, n_local - c0);
    __syncthreads();
    // local candidate j has the global id c_lo + c_stride * j (c_stride = 1: a contiguous block; = world: rows g % world == rank of a
    // row-sharded table); a listed id lies in this shard when it is on that lattice
    for (int j = threadIdx.x; j < len; j += 256) keys[j] = make_key(row[c0 + j], descending != 0, (uint32_t)(c_lo + c_stride * (c0 + j)));
    __syncthreads();
    for (int64_t f = threadIdx.x; f < nf; f += 256) {                  // filtered candidates and the golds themselves do not count
      const int64_t off = (int64_t)fids[f] - c_lo;
      const int64_t id = off / c_stride - c0;
      if (off >= 0 && off % c_stride == 0 && id >= 0 && id < len) keys[id] = KEY_MAX;
    }
    for (int64_t o = g0 + threadIdx.x; o < g1; o += 256) {
      const int64_t off = (int64_t)gold_ids[o] - c_lo;
      const int64_t id = off / c_stride - c0;
      if (off >= 0 && off % c_stride == 0 && id >= 0 && id < len) keys[id] = KEY_MAX;
    }
    __syncthreads();
    for (int64_t gi = g0; gi < g1; ++gi) {
      const uint64_t gk = make_key(gold_scores[gi], descending != 0, (uint32_t)gold_ids[gi]);
      int cnt = 0;
      for (int j = threadIdx.x; j < len; j += 256) cnt += keys[j] < gk ? 1 : 0;
      cnt = wave_sum_int(cnt);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
      __syncthreads();
      if (threadIdx.x == 0) {
        const int32_t have = counts[gi];
        if (have >= 0) counts[gi] = have + red[0] + red[1] + red[2] + red[3];
      }
      __syncthreads();
    }
  }
}

extern "C" int ktup_eval_gold_rank_counts(const float* scores, int64_t lds, int64_t nq, int64_t n_local, int64_t cand_lo,
                                          int descending, const int64_t* filt_off, const int32_t* filt_ids,
                                          const int64_t* gold_off, const int32_t* gold_ids, const float* gold_scores,
                                          int32_t* counts, void* stream) {
  return ktup_eval_gold_rank_counts_strided(scores, lds, nq, n_local, cand_lo, 1, descending, filt_off, filt_ids, gold_off, gold_ids, gold_scores,
                                            counts, stream);
}

extern "C" int ktup_eval_gold_rank_counts_strided(const float* scores, int64_t lds, int64_t nq, int64_t n_local, int64_t cand_lo,
                                                  int64_t cand_stride, int descending, const int64_t* filt_off, const int32_t* filt_ids,
                                                  const int64_t* gold_off, const int32_t* gold_ids, const float* gold_scores,
                                                  int32_t* counts, void* stream) {
  const char* name = "ktup_eval_gold_rank_counts";
  KTUP_REQUIRE(nq >= 0 && n_local >= 0 && cand_lo >= 0 && cand_stride >= 1 && (n_local == 0 || lds >= n_local), "%s: bad sizes", name);
  if (nq == 0) return KTUP_OK;
  KTUP_REQUIRE((scores || n_local == 0) && gold_off && gold_ids && gold_scores && counts && ((filt_off == nullptr) || filt_ids),
               "%s: null pointer argument", name);
  KTUP_REQUIRE(cand_lo + cand_stride * n_local <= 0x7fffffffll + cand_stride, "%s: candidate ids are 32-bit", name);
  const int ch = (int)(n_local < 1 ? 64 : (n_local < CHUNK_KEYS ? ((n_local + 63) & ~63ll) : CHUNK_KEYS));
  const size_t lbytes = (size_t)ch * 8;
  allow_lds((const void*)gold_rank_counts_kernel, lbytes);
  hipLaunchKernelGGL(gold_rank_counts_kernel, dim3((unsigned)nq), dim3(256), lbytes, (hipStream_t)stream, scores, lds, n_local, cand_lo,
                     cand_stride, descending, filt_off, filt_ids, gold_off, gold_ids, gold_scores, ch, counts);
  return check_launch(name);
}

extern "C" int ktup_eval_rec_metrics(const int32_t* top_ids, int64_t nq, int topn, const int64_t* gold_off,
                                     const int32_t* gold_ids, double* out, void* stream) {
  const char* name = "ktup_eval_rec_metrics";
  KTUP_REQUIRE(nq >= 0 && topn > 0, "%s: bad sizes", name);
  if (nq == 0) return KTUP_OK;
  KTUP_REQUIRE(top_ids && gold_off && gold_ids && out, "%s: null pointer argument", name);
  hipLaunchKernelGGL(rec_metrics_kernel, dim3((unsigned)(topn <= 16 ? (nq + 3) / 4 : (nq + 63) / 64)), dim3(64), 0, (hipStream_t)stream, top_ids, nq,
                     topn, gold_off, gold_ids, out);
  return check_launch(name);
}
