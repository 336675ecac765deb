// Row-gradient reduction by sorted segments: the large-batch / hot-row alternative to per-element float atomics.
//
// The backward of every scorer on the path ends in  gT[ids[e]] += (+/-) G[e]  for e over the batch (the dense .grad of an
// nn.Embedding, e.g. transE.py:51-63 / jTransUP.py:122-143 differentiated).  With atomics that is d float atomics per gathered
// row, and rows that many samples share (716,800 pairs on 6,040 users; Zipf ids on a sharded table) serialise on the same L2
// lines: round 1 measured 5.2 ms for the 716,800-pair KTUP backward against a 0.11 ms forward.  Here the batch's ids are
// counting-sorted (histogram of int atomics -> exclusive scan -> scatter; the key range is the table's row count), the scorer's
// backward writes its per-sample row gradients G (n x d) with plain coalesced stores, and a lane group walks a chunk of the
// sorted order summing consecutive entries of one row in registers; runs cut by a chunk edge are joined inside the workgroup, so a
// row's sum reaches memory with a plain read-modify-write, or one float4 atomic per WORKGROUP it spans (see the kernel).  Optionally every flushed row is also added to a second table
// through an int32 map (KTUP: gE[item2ent[i]] += what goes to gI[i], skipping the pad entity, jTransUP.py:96,122-130).
//
// ws layout (int32): start[n_rows + 1] | rank[m] | perm[m] | skey[m]
#include "ktup_common.h"

namespace ktup {
namespace {

constexpr int SCAN_NARROW = 256;    // one wave per SIMD and few registers: the scan must fit on a CU beside a gradient kernel that holds
                               // most of its registers and LDS (it runs on the side stream, ktup_runtime.hip fork_side)
constexpr int SCAN_WIDE = 1024;    // the sharded step's sort runs alone (ktup_shard_step.hip): 8192 counters per sweep
constexpr int SCAN_V = 8;      // counters per thread per sweep

// entries [0, n_src) take their key from ids, entries [n_src, m) from ids2 (the two roles of a triple's entities: one sort)
__global__ __launch_bounds__(256) void seg_hist_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ ids2, int64_t n_src,
                                                       int64_t m, int32_t* __restrict__ start, int32_t* __restrict__ rank) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (int64_t)gridDim.x * 256)
    rank[e] = atomicAdd(start + (e < n_src ? ids[e] : ids2[e - n_src]), 1);
}

// exclusive scan in place over K + 1 counters (the last one receives the total); one workgroup, SCAN_T x SCAN_V items per sweep
template <int SCAN_T>
__global__ __launch_bounds__(SCAN_T) void seg_scan_kernel(int32_t* __restrict__ start, int64_t K) {
  __shared__ int32_t wsum[SCAN_T / 64];
  __shared__ int32_t carry_s;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base <= K; base += (int64_t)SCAN_T * SCAN_V) {
    const int64_t i0 = base + (int64_t)t * SCAN_V;
    int32_t v[SCAN_V];
    int32_t mine = 0;
#pragma unroll
    for (int c = 0; c < SCAN_V; ++c) { v[c] = (i0 + c < K) ? start[i0 + c] : 0; mine += v[c]; }
    int32_t inc = mine;                                         // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    int32_t total = 0;
    for (int k = 0; k < SCAN_T / 64; ++k) total += wsum[k];
    int32_t run = carry_s + woff + inc - mine;
#pragma unroll
    for (int c = 0; c < SCAN_V; ++c) {
      if (i0 + c <= K) start[i0 + c] = run;                     // index K gets the grand total (its own count is 0)
      run += v[c];
    }
    __syncthreads();
    if (t == 0) carry_s += total;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void seg_scatter_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ ids2, int64_t n_src,
                                                          int64_t m, const int32_t* __restrict__ start, const int32_t* __restrict__ rank,
                                                          int32_t* __restrict__ perm, int32_t* __restrict__ skey) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (int64_t)gridDim.x * 256) {
    const int64_t id = e < n_src ? ids[e] : ids2[e - n_src];
    const int32_t pos = start[id] + rank[e];
    perm[pos] = (int32_t)e;
    skey[pos] = (int32_t)id;
  }
}

struct SegArgs {
  const float4* G; int64_t ldg4; int nch; int64_t n_src;
  const int32_t *perm, *skey; int64_t m, sign_split;
  int64_t src_off;                 // entry e >= n_src reads row e - src_off of G (n_src: the second half re-reads the first)
  const int32_t* m_dev;            // non-null: the number of sorted entries lives on the device (<= m; entries with a negative key were skipped)
  float* gT; int64_t ldt;
  const int32_t* map2; int64_t pad2; float* gT2; int64_t ldt2;
  int chunk;
};

// GL lanes own one chunk of `chunk` consecutive sorted entries; lane l holds float4 chunks l, l + GL, ... of the running row sum.
// A row whose entries all lie inside the chunk belongs to this lane group alone: plain read-modify-write.  The runs cut by the
// chunk's edges (a user's 119 pairs span several 64-entry chunks) go to LDS instead of to float atomics: the workgroup's
// 256 / GL chunks are consecutive in the sorted order, so after a barrier one lane group walks the 2 x 256 / GL edge partials in
// order, joins equal keys, and only the rows that continue into the NEIGHBOURING workgroups need atomics -- 2 rows per
// 256 / GL chunks instead of 2 per chunk (2.2 M float atomics -> 0.28 M at 716,800 pairs; ~41 G atomics/s is the chip's rate).
// Rows of the mapped second table (several rows may share one) always take atomics.
template <int GL, int CPL>
__global__ __launch_bounds__(256) void seg_reduce_kernel(SegArgs a) {
  const int lane = threadIdx.x % GL, grp = threadIdx.x / GL;
  constexpr int GPB = 256 / GL;
  constexpr int ROW4 = GL * CPL;                                  // float4 per edge partial
  __shared__ float4 edge[2 * GPB * ROW4];
  __shared__ int32_t ekey[2 * GPB];
  if (a.m_dev) a.m = *a.m_dev;
  const int64_t nchunks = (a.m + a.chunk - 1) / a.chunk;
  auto to_memory = [&](int32_t key, const float4* acc, bool atomic) {
    float* row = a.gT + (int64_t)key * a.ldt;
    float* row2 = nullptr;
    if (a.map2) {
      const int64_t t2 = a.map2[key];
      if (t2 != a.pad2) row2 = a.gT2 + t2 * a.ldt2;
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int ch = lane + j * GL;
      if (ch < a.nch) {
        if (atomic) {
          atomic_add4(row + 4 * ch, acc[j]);
        } else {
          float4* dst = reinterpret_cast<float4*>(row + 4 * ch);
          *dst = *dst + acc[j];
        }
        if (row2) atomic_add4(row2 + 4 * ch, acc[j]);
      }
    }
  };
  for (int64_t c0 = (int64_t)blockIdx.x * GPB; c0 < nchunks; c0 += (int64_t)gridDim.x * GPB) {
    const int64_t c = c0 + grp;
    const bool active = c < nchunks;
    if (lane == 0) { ekey[2 * grp] = -1; ekey[2 * grp + 1] = -1; }
    if (active) {
      const int64_t k0 = c * a.chunk, k1 = min(a.m, k0 + a.chunk);
      float4 acc[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) acc[j] = f4zero();
      int32_t cur = a.skey[k0];
      const int32_t before = k0 > 0 ? a.skey[k0 - 1] : -1, after = k1 < a.m ? a.skey[k1] : -1;
      bool head_open = true;                                      // no run has been closed yet: `cur` may continue the previous chunk
      auto flush = [&](int32_t key, bool last) {
        const bool is_head = head_open && key == before, is_tail = last && key == after;
        if (is_head || is_tail) {
          const int slot = is_head ? 2 * grp : 2 * grp + 1;       // a run that is both (the whole chunk is one row) is a head
          if (lane == 0) ekey[slot] = key;
#pragma unroll
          for (int j = 0; j < CPL; ++j) edge[slot * ROW4 + lane + j * GL] = acc[j];
        } else {
          to_memory(key, acc, false);
        }
        head_open = false;
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] = f4zero();
      };
      constexpr int UNR = CPL == 1 ? 8 : 4;                       // independent row reads in flight per lane
      for (int64_t k = k0; k < k1; k += UNR) {
        int32_t key[UNR], e[UNR];
        float4 v[UNR][CPL];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const bool on = k + u < k1;
          key[u] = on ? a.skey[k + u] : -1;
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
          const float sg = e[u] < a.sign_split ? 1.f : -1.f;
#pragma unroll
          for (int j = 0; j < CPL; ++j) acc[j] = fma4(sg, v[u][j], acc[j]);
        }
      }
      flush(cur, true);
    }
    __syncthreads();
    if (grp == 0) {                                               // join the edge partials of this workgroup's consecutive chunks
      const int64_t w0 = c0 * a.chunk, w1 = min(a.m, (c0 + GPB) * a.chunk);
      const int32_t wbefore = w0 > 0 ? a.skey[w0 - 1] : -1, wafter = w1 < a.m ? a.skey[w1] : -1;
      float4 acc[CPL];
      int32_t cur = -1;
      for (int sl = 0; sl < 2 * GPB; ++sl) {
        const int32_t key = ekey[sl];
        if (key < 0) continue;
        if (key != cur) {
          if (cur >= 0) to_memory(cur, acc, cur == wbefore || cur == wafter);
          cur = key;
#pragma unroll
          for (int j = 0; j < CPL; ++j) acc[j] = f4zero();
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] = acc[j] + edge[sl * ROW4 + lane + j * GL];
      }
      if (cur >= 0) to_memory(cur, acc, cur == wbefore || cur == wafter);
    }
    __syncthreads();                                              // the next trip rewrites the edge slots
  }
}

}  // namespace

size_t seg_ws_bytes(int64_t m, int64_t n_rows) {
  if (m <= 0 || n_rows <= 0) return 0;
  return (((size_t)(n_rows + 1) + (size_t)3 * m) * sizeof(int32_t) + 15) & ~(size_t)15;
}

bool seg_covers(const float* G, int64_t ldg, int d, int64_t n_src, int64_t m, int64_t n_rows, const float* gT, int64_t ldt,
                const int32_t* map2, const float* gT2, int64_t ldt2, const void* ws) {
  if (!ws || d % 4 || d > 1024 || ldg % 4 || ldt % 4 || (map2 && ldt2 % 4) || !aligned16(G) || !aligned16(gT) || (map2 && !aligned16(gT2)))
    return false;
  return m < (1ll << 31) && n_rows < (1ll << 31) && (m == n_src || m == 2 * n_src);
}

int seg_sort(const int64_t* ids, const int64_t* ids2, int64_t n_src, int64_t m, int64_t n_rows, void* ws, hipStream_t st, const char* name) {
  if (m == 0) return KTUP_OK;
  int32_t* start = reinterpret_cast<int32_t*>(ws);
  int32_t* rank = start + (n_rows + 1);
  int32_t* perm = rank + m;
  int32_t* skey = perm + m;
  if (hipMemsetAsync(start, 0, (size_t)(n_rows + 1) * sizeof(int32_t), st) != hipSuccess) return check_launch(name);
  const int gm = grid_for((m + 255) / 256, 2048);
  hipLaunchKernelGGL(seg_hist_kernel, dim3(gm), dim3(256), 0, st, ids, ids2, n_src, m, start, rank);
  hipLaunchKernelGGL(seg_scan_kernel<SCAN_NARROW>, dim3(1), dim3(SCAN_NARROW), 0, st, start, n_rows);
  hipLaunchKernelGGL(seg_scatter_kernel, dim3(gm), dim3(256), 0, st, ids, ids2, n_src, m, start, rank, perm, skey);
  return check_launch(name);
}

int seg_apply(const float* G, int64_t ldg, int d, int64_t n_src, int64_t m, int64_t sign_split, int64_t n_rows, float* gT, int64_t ldt,
              const int32_t* map2, int64_t pad2, float* gT2, int64_t ldt2, const void* ws, hipStream_t st, const char* name) {
  if (m == 0) return KTUP_OK;
  const int32_t* perm = reinterpret_cast<const int32_t*>(ws) + (n_rows + 1) + m;
  const int32_t* skey = perm + m;
  SegArgs a;
  a.G = reinterpret_cast<const float4*>(G); a.ldg4 = ldg / 4; a.nch = d / 4; a.n_src = n_src;
  a.perm = perm; a.skey = skey; a.m = m; a.sign_split = sign_split; a.src_off = n_src; a.m_dev = nullptr;
  a.gT = gT; a.ldt = ldt; a.map2 = map2; a.pad2 = pad2; a.gT2 = gT2; a.ldt2 = ldt2;
  int64_t ch = m / 8192;
  a.chunk = (int)(ch < 8 ? 8 : ch > 64 ? 64 : ch);
  a.chunk = (a.chunk + 3) & ~3;
  const int64_t nchunks = (m + a.chunk - 1) / a.chunk;
#define KTUP_SEG(GL, CPL)                                                                                         \
  {                                                                                                               \
    const int grid = grid_for((nchunks + (256 / GL) - 1) / (256 / GL), 4096);                                     \
    hipLaunchKernelGGL((seg_reduce_kernel<GL, CPL>), dim3(grid), dim3(256), 0, st, a);                            \
    return check_launch(name);                                                                                    \
  }
  if (a.nch <= 16) KTUP_SEG(16, 1)
  if (a.nch <= 32) KTUP_SEG(32, 1)
  if (a.nch <= 64) KTUP_SEG(64, 1)
  if (a.nch <= 128) KTUP_SEG(64, 2)
  KTUP_SEG(64, 4)
#undef KTUP_SEG
}

// ---- pieces for callers that build the sorted order themselves (ktup_shard_step.hip: the histogram is a by-product of its
// id routing).  seg_scan_wide: exclusive scan of K + 1 counters in place (index K receives the total) with a 1024-thread
// workgroup; seg_apply_sorted: the reduction over perm / skey with the entry count read from the device (*m_dev <= m_max).
int seg_scan_wide(int32_t* start, int64_t K, hipStream_t st, const char* name) {
  hipLaunchKernelGGL(seg_scan_kernel<SCAN_WIDE>, dim3(1), dim3(SCAN_WIDE), 0, st, start, K);
  return check_launch(name);
}

int seg_apply_sorted(const float* G, int64_t ldg, int d, int64_t n_src, int64_t src_off, const int32_t* perm, const int32_t* skey,
                     int64_t m_max, const int32_t* m_dev, float* gT, int64_t ldt, hipStream_t st, const char* name) {
  if (m_max == 0) return KTUP_OK;
  if (d % 4 || d > 1024 || ldg % 4 || ldt % 4 || !aligned16(G) || !aligned16(gT) || m_max >= (1ll << 31)) return 1;
  SegArgs a;
  a.G = reinterpret_cast<const float4*>(G); a.ldg4 = ldg / 4; a.nch = d / 4; a.n_src = n_src;
  a.perm = perm; a.skey = skey; a.m = m_max; a.sign_split = m_max; a.src_off = src_off; a.m_dev = m_dev;
  a.gT = gT; a.ldt = ldt; a.map2 = nullptr; a.pad2 = -1; a.gT2 = nullptr; a.ldt2 = 0;
  int64_t ch = m_max / 8192;
  a.chunk = (int)(ch < 8 ? 8 : ch > 64 ? 64 : ch);
  a.chunk = (a.chunk + 3) & ~3;
  const int64_t nchunks = (m_max + a.chunk - 1) / a.chunk;
#define KTUP_SEG(GL, CPL)                                                                                         \
  {                                                                                                               \
    const int grid = grid_for((nchunks + (256 / GL) - 1) / (256 / GL), 4096);                                     \
    hipLaunchKernelGGL((seg_reduce_kernel<GL, CPL>), dim3(grid), dim3(256), 0, st, a);                            \
    return check_launch(name);                                                                                    \
  }
  if (a.nch <= 16) KTUP_SEG(16, 1)
  if (a.nch <= 32) KTUP_SEG(32, 1)
  if (a.nch <= 64) KTUP_SEG(64, 1)
  if (a.nch <= 128) KTUP_SEG(64, 2)
  KTUP_SEG(64, 4)
#undef KTUP_SEG
}

// gT[ids[e]] += sign(e) * G[e mod n_src]  for e in [0, m); sign = + for e < sign_split, - otherwise (m = n_src or 2 n_src).
// Returns KTUP_OK / an error, or 1 when the shape is not covered (d % 4, alignment, sizes): the caller keeps its atomics.
int seg_reduce(const float* G, int64_t ldg, int d, int64_t n_src, const int64_t* ids, int64_t m, int64_t sign_split, int64_t n_rows,
               float* gT, int64_t ldt, const int32_t* map2, int64_t pad2, float* gT2, int64_t ldt2, void* ws, hipStream_t st,
               const char* name) {
  if (m == 0) return KTUP_OK;
  if (!seg_covers(G, ldg, d, n_src, m, n_rows, gT, ldt, map2, gT2, ldt2, ws)) return 1;
  if (int e = seg_sort(ids, ids + n_src, n_src, m, n_rows, ws, st, name)) return e;
  return seg_apply(G, ldg, d, n_src, m, sign_split, n_rows, gT, ldt, map2, pad2, gT2, ldt2, ws, st, name);
}

}  // namespace ktup

extern "C" size_t ktup_segment_workspace_bytes(int64_t m, int64_t n_rows) { return ktup::seg_ws_bytes(m, n_rows); }

extern "C" int ktup_segment_reduce_rows(const float* G, int64_t ldg, int d, int64_t n_src, const int64_t* ids, int64_t m,
                                        int64_t sign_split, int64_t n_rows, float* gT, int64_t ldt, const int32_t* map2, int64_t pad2,
                                        float* gT2, int64_t ldt2, void* ws, void* stream) {
  const char* name = "ktup_segment_reduce_rows";
  KTUP_REQUIRE(m >= 0 && n_rows > 0 && d > 0, "%s: bad sizes", name);
  if (m == 0) return KTUP_OK;
  KTUP_REQUIRE(G && ids && gT && ws, "%s: null pointer argument", name);
  KTUP_REQUIRE(!map2 || gT2, "%s: map2 needs the second table", name);
  const int rc = ktup::seg_reduce(G, ldg, d, n_src, ids, m, sign_split, n_rows, gT, ldt, map2, pad2, gT2, ldt2, ws, (hipStream_t)stream, name);
  if (rc == 1)
    return ktup::set_error(KTUP_ERR_UNSUPPORTED, "%s: needs d %% 4 == 0, 16-byte aligned rows, m == n_src or 2 n_src, sizes < 2^31", name);
  return rc;
}
