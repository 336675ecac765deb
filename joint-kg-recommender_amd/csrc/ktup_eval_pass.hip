// K16 + K17 fused for a whole evaluation pass (soft gate, squared L2): all-item TUP / KTUP scores AND the filtered top-n of
// every user in ONE launch that never writes the (users x items) score matrix.
//
// Reference: transUP.py:84-102 / jTransUP.py:163-191 (evaluate / evaluateRec) produce a (B x N) matrix per batch of 512 users,
// which utils/misc.py:186-248 copies to the host, argsorts and walks.  Round 1 kept that shape on the device: per batch a
// six-GEMM score kernel wrote B x N x 4 bytes (78 MB over an ml1m pass) and a ranking kernel read them back -- 12 batches x
// ~5 launches, 1.05 ms, the GEMMs at 18 % of the fp32 matrix peak.  Here one persistent kernel walks the pass:
//   * scores: the same six bilinear terms as pairs_l2_mc_kernel (ktup_eval_mc.hip: |a|^2 - 2 s (a.n) + s^2 |n|^2 with every term a
//     (users x d).(d x items) product on v_mfma_f32_16x16x4_f32), the same k order and the same epilogue arithmetic, so a score
//     is bit-identical to the matrix route's;
//   * a workgroup owns 64 users (4 waves x 16) and a contiguous split of the catalogue.  A wave keeps its 16 users' three
//     operand vectors in REGISTERS for the whole pass (the A operands: 18 float4 per lane at d = 100); items stream through a
//     double-buffered 16-item LDS tile (the B operands, 19 KB per buffer) that the next tile's global loads refill under the
//     MFMAs -- LDS stays at ~60 KB, two workgroups per CU;
//   * ranking: a user's sorted top-n list (64-bit keys = order-preserving score image << 32 | item id, the order of
//     ktup_rank.hip: ascending score, ties -> lower id) lives in REGISTERS, one element per lane of the 16-lane row that owns
//     the user in the MFMA output layout (lane (kq, j) holds element j of users 4 kq + reg).  A score is a candidate only if its
//     key beats the user's current n-th key and its bit in the wave's filter bitmap (built once from the CSR filter lists, for
//     this workgroup's item split only: 16 users x (split / 32) words of LDS) is clear; candidates are rare after the first
//     tiles (~n ln(N / n) per user).  The four rows insert their candidates in parallel: the shift of the sorted list is one DPP
//     row_shr per half key, the position a popcount of a ballot -- no LDS round trip (the first version kept the lists in LDS
//     and inserted one candidate per wave at a time: 830 us per ml1m pass, 3 LDS latencies per candidate);
//   * item scalars (v.NV, |C0|^2, C0.NV, |NV|^2) are computed once per pass by a small kernel and travel with the tile;
//   * the splits' partial lists are merged by a second, tiny launch (topk_merge_kernel).
// L1 distance and the ST-Gumbel gate do not decompose into GEMMs: they keep the per-batch kernels of ktup_eval.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ktup_common.h"
#include "ktup_lane_swap.h"
#include "ktup_pref_geom.h"

namespace ktup {
namespace {

constexpr uint64_t PKEY_MAX = ~0ull;
constexpr int IBT = 16;        // items per LDS tile
constexpr int TOPN_MAX = 16;   // top-n list capacity per user: one element per lane of a 16-lane row

KTUP_DEV uint64_t pass_key(float s, uint32_t id) {   // ktup_rank.hip make_key, ascending (lower score = better)
  if (s == 0.f) s = 0.f;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | id;
}

template <int NCH_>
struct PGeom {
  static constexpr int NCH = NCH_, D = 4 * NCH;
  static constexpr int KG = (D + 15) / 16;
  static constexpr bool TAIL1 = NCH - 4 * (KG - 1) == 1;       // d % 16 == 4: the last chunk goes through one b32-operand MFMA
  static constexpr int KGF = TAIL1 ? KG - 1 : KG;
  static_assert(TAIL1 || NCH % 4 == 0, "k groups must be whole (d % 16 in {0, 4})");
  static constexpr int P4 = NCH | 1;                           // odd float4 row pitch of the item tile
  static constexpr int TILE_F4 = IBT * 3 * P4;
  static constexpr int LPT = (IBT * 3 * NCH + IBT + 255) / 256;   // float4 loads per thread and tile (+ one scalar quad per item)
};

struct PassArgs {
  const float *QW, *C0, *C1, *C2;   // users: rows of 3 d floats [AU | u | NU]; items: three (n_items x d) arrays
  const float* ISC;                 // [n_items][4] item scalars (item_scalars_kernel)
  int64_t nq, n_items;
  const int64_t* filt_off;          // CSR filter sets per user (global item ids); null = no filter
  const int32_t* filt_ids;
  int topn, nsplit;
  int64_t split_items;              // items per split (multiple of IBT)
  uint64_t* part;                   // [nq][nsplit][topn] partial lists
  int bm_words;                     // filter bitmap words per user (covers one split)
};

template <typename G>
__global__ __launch_bounds__(256) void eval_pass_kernel(PassArgs a) {
  constexpr int NCH = G::NCH, D = G::D, KGF = G::KGF, P4 = G::P4, LPT = G::LPT;
  constexpr bool TAIL1 = G::TAIL1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* Xb = reinterpret_cast<v4*>(smem);                                   // [2][IBT][3][P4] item tiles
  float* isc = reinterpret_cast<float*>(Xb + 2 * G::TILE_F4);             // [2][IBT][4] item scalars
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* wbase = reinterpret_cast<char*>(isc + 2 * IBT * 4) + (size_t)w * ((size_t)16 * 4 * 4 + (size_t)16 * a.bm_words * 4);
  float* usc = reinterpret_cast<float*>(wbase);                           // [16][4] user scalars
  uint32_t* bm = reinterpret_cast<uint32_t*>(usc + 64);                   // [16][bm_words] filter bits of this split
  uint64_t tkr[4] = {PKEY_MAX, PKEY_MAX, PKEY_MAX, PKEY_MAX};             // element j of the sorted top-n lists of users 4 kq + reg
  const int64_t u0 = (int64_t)blockIdx.x * 64 + 16 * w;                   // this wave's first user
  const int64_t i_lo = (int64_t)blockIdx.y * a.split_items;
  const int64_t i_hi = min(a.n_items, i_lo + a.split_items);
  const int topn = a.topn;
  // ---- per-wave setup: top-n lists, filter bitmap of the split, user scalars, A operands
  for (int idx = lane; idx < 16 * a.bm_words; idx += 64) bm[idx] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (a.filt_off) {
    for (int r = 0; r < 16; ++r) {
      if (u0 + r >= a.nq) break;
      const int64_t f0 = a.filt_off[u0 + r], f1 = a.filt_off[u0 + r + 1];
      for (int64_t f = f0 + lane; f < f1; f += 64) {
        const int64_t id = (int64_t)a.filt_ids[f] - i_lo;
        if (id >= 0 && id < i_hi - i_lo) atomicOr(bm + r * a.bm_words + (id >> 5), 1u << (id & 31));
      }
    }
  }
  // user scalars with the arithmetic of pairs_l2_mc_kernel (8 lanes per row, chunks l, l + 8, ..., then xor-shuffles):
  // u.NU, |AU|^2, AU.NU, |NU|^2
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = 8 * pass + (lane >> 3);
    const bool ok = u0 + r < a.nq;
    const v4* r0 = reinterpret_cast<const v4*>(a.QW + (ok ? u0 + r : 0) * 3 * D);
    v4 s0 = (v4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    for (int c = lane & 7; c < NCH; c += 8) {
      const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
      const v4 x0 = ok ? r0[c] : zero, x1 = ok ? r0[NCH + c] : zero, x2 = ok ? r0[2 * NCH + c] : zero;
      s0 += x1 * x2; s1 += x0 * x0; s2 += x0 * x2; s3 += x2 * x2;
    }
    float f0 = (s0[0] + s0[1]) + (s0[2] + s0[3]), f1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    float f2 = (s2[0] + s2[1]) + (s2[2] + s2[3]), f3 = (s3[0] + s3[1]) + (s3[2] + s3[3]);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      f0 += __shfl_xor(f0, m, 64); f1 += __shfl_xor(f1, m, 64); f2 += __shfl_xor(f2, m, 64); f3 += __shfl_xor(f3, m, 64);
    }
    if ((lane & 7) == 0) { usc[r * 4 + 0] = f0; usc[r * 4 + 1] = f1; usc[r * 4 + 2] = f2; usc[r * 4 + 3] = f3; }
  }
  // A operands of the whole pass: lane (kq, j) = user row j, chunks 4 g + kq of AU / u / NU (rows past nq are zero)
  v4 aAU[KGF], au[KGF], aNU[KGF];
  float tAU = 0.f, tu = 0.f, tNU = 0.f;
  {
    const bool ok = u0 + j < a.nq;
    const v4* r0 = reinterpret_cast<const v4*>(a.QW + (ok ? u0 + j : 0) * 3 * D);
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < KGF; ++g) {
      aAU[g] = ok ? r0[4 * g + kq] : zero;
      au[g] = ok ? r0[NCH + 4 * g + kq] : zero;
      aNU[g] = ok ? r0[2 * NCH + 4 * g + kq] : zero;
    }
    if (TAIL1) {
      const float* rf = reinterpret_cast<const float*>(r0);
      tAU = ok ? rf[16 * KGF + kq] : 0.f;
      tu = ok ? rf[D + 16 * KGF + kq] : 0.f;
      tNU = ok ? rf[2 * D + 16 * KGF + kq] : 0.f;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- the item stream: tile t of this split, double buffered
  const int64_t ntile = (i_hi - i_lo + IBT - 1) / IBT;
  v4 pre[LPT];
  // loop-invariant geometry of this thread's LPT float4 of a tile: source element, row inside the tile, LDS destination
  const float* fsrc[LPT];
  int frow[LPT], fdst[LPT];
#pragma unroll
  for (int l = 0; l < LPT; ++l) {
    const int idx = tid + 256 * l;
    if (idx < IBT * 3 * NCH) {
      const int row = idx / (3 * NCH), rem = idx - row * (3 * NCH), vec = rem / NCH, c = rem - vec * NCH;
      fsrc[l] = (vec == 0 ? a.C0 : vec == 1 ? a.C1 : a.C2) + (i_lo + row) * D + 4 * c;
      frow[l] = row;
      fdst[l] = (row * 3 + vec) * P4 + c;
    } else if (idx < IBT * 3 * NCH + IBT) {
      const int row = idx - IBT * 3 * NCH;
      fsrc[l] = a.ISC + (i_lo + row) * 4;
      frow[l] = row;
      fdst[l] = -1 - row;                                                 // scalar quad of item `row`
    } else {
      fsrc[l] = nullptr; frow[l] = IBT; fdst[l] = 0;
    }
  }
  auto fetch = [&](int64_t t) {                                           // global loads of tile t into registers
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      v4 val = (v4){0.f, 0.f, 0.f, 0.f};
      if (fsrc[l] && i_lo + t * IBT + frow[l] < i_hi)
        val = *reinterpret_cast<const v4*>(fsrc[l] + t * IBT * (fdst[l] >= 0 ? D : 4));
      pre[l] = val;
    }
  };
  auto stash = [&](int buf) {                                             // registers -> LDS tile (+ its item scalars)
    v4* X = Xb + buf * G::TILE_F4;
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      if (fsrc[l]) {
        if (fdst[l] >= 0) X[fdst[l]] = pre[l];
        else *reinterpret_cast<v4*>(isc + (buf * IBT + (-1 - fdst[l])) * 4) = pre[l];
      }
    }
  };
  if (ntile > 0) {
    fetch(0);
    stash(0);
    __syncthreads();
  }
  const int rowbase = 16 * kq;
  uint64_t thr[4] = {PKEY_MAX, PKEY_MAX, PKEY_MAX, PKEY_MAX};             // the users' current n-th keys (lane topn - 1 of the row)
  for (int64_t t = 0; t < ntile; ++t) {
    const int buf = (int)(t & 1);
    if (t + 1 < ntile) fetch(t + 1);                                      // in flight under the MFMAs below
    const v4* ib = Xb + buf * G::TILE_F4 + (j * 3) * P4 + kq;             // lane (kq, col j): C0, v, NV of item j of the tile
    v4 uNV = (v4){0.f, 0.f, 0.f, 0.f}, NUv = uNV, AUC0 = uNV, AUNV = uNV, NUC0 = uNV, NUNV = uNV;
#pragma unroll
    for (int g = 0; g < KGF; ++g) {
      const v4 bC0 = ib[4 * g], bv = ib[P4 + 4 * g], bNV = ib[2 * P4 + 4 * g];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uNV = __builtin_amdgcn_mfma_f32_16x16x4f32(au[g][c], bNV[c], uNV, 0, 0, 0);
        NUv = __builtin_amdgcn_mfma_f32_16x16x4f32(aNU[g][c], bv[c], NUv, 0, 0, 0);
        AUC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aAU[g][c], bC0[c], AUC0, 0, 0, 0);
        AUNV = __builtin_amdgcn_mfma_f32_16x16x4f32(aAU[g][c], bNV[c], AUNV, 0, 0, 0);
        NUC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aNU[g][c], bC0[c], NUC0, 0, 0, 0);
        NUNV = __builtin_amdgcn_mfma_f32_16x16x4f32(aNU[g][c], bNV[c], NUNV, 0, 0, 0);
      }
    }
    if (TAIL1) {                                                          // coordinates 16 KGF + kq
      const float* jf = reinterpret_cast<const float*>(Xb + buf * G::TILE_F4 + (j * 3) * P4 + 4 * KGF) + kq;
      const float bC0 = jf[0], bv = jf[4 * P4], bNV = jf[8 * P4];
      uNV = __builtin_amdgcn_mfma_f32_16x16x4f32(tu, bNV, uNV, 0, 0, 0);
      NUv = __builtin_amdgcn_mfma_f32_16x16x4f32(tNU, bv, NUv, 0, 0, 0);
      AUC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(tAU, bC0, AUC0, 0, 0, 0);
      AUNV = __builtin_amdgcn_mfma_f32_16x16x4f32(tAU, bNV, AUNV, 0, 0, 0);
      NUC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(tNU, bC0, NUC0, 0, 0, 0);
      NUNV = __builtin_amdgcn_mfma_f32_16x16x4f32(tNU, bNV, NUNV, 0, 0, 0);
    }
    // ---- epilogue: lane (kq, j) holds users 4 kq + reg (reg = 0..3) x item j of the tile -- pairs_l2_mc_kernel's arithmetic
    const v4 is4 = *reinterpret_cast<const v4*>(isc + (buf * IBT + j) * 4);
    const int64_t item = i_lo + t * IBT + j;
    const int64_t lid = item - i_lo;                                      // id inside the split (bitmap index)
    uint64_t ck[4];
    bool cand[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int ur = 4 * kq + reg;
      const v4 us4 = *reinterpret_cast<const v4*>(usc + ur * 4);
      const float s = (us4[0] + uNV[reg]) - (NUv[reg] + is4[0]);
      const float aa = fmaf(-2.f, AUC0[reg], us4[1] + is4[1]);
      const float an = (us4[2] + AUNV[reg]) - (NUC0[reg] + is4[2]);
      const float nn = fmaf(2.f, NUNV[reg], us4[3] + is4[3]);
      const float score = fmaf(s * s, nn, fmaf(-2.f * s, an, aa));
      ck[reg] = pass_key(score, (uint32_t)item);
      bool c = item < i_hi && u0 + ur < a.nq && ck[reg] < thr[reg];
      if (c) c = ((bm[ur * a.bm_words + (lid >> 5)] >> (lid & 31)) & 1u) == 0u;
      cand[reg] = c;
    }
    // ---- insertions: every 16-lane row serves its own users, one candidate per row and round, all four rows in parallel
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const unsigned long long any = __ballot(cand[reg]);
      if (!any) continue;                                                                    // the common case after the first tiles
      uint32_t rowmask = (uint32_t)((any >> rowbase) & 0xffffull);                           // this row's candidate lanes
      while (__ballot(rowmask != 0u)) {                                                      // (wave-uniform trip count)
        const bool act = rowmask != 0u;
        const int src = rowbase + (act ? __ffs((int)rowmask) - 1 : 0);
        rowmask &= rowmask - 1u;
        const uint64_t key = ((uint64_t)(uint32_t)__shfl((int)(ck[reg] >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)ck[reg], src, 64);
        const uint64_t mine = tkr[reg];
        // sorted row: element j - 1 moves to j behind the insertion point (DPP row_shr:1; lane 0 of a row keeps its own value)
        const uint32_t llo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)mine, (int)(uint32_t)mine, 0x111, 0xf, 0xf, false);
        const uint32_t lhi = (uint32_t)__builtin_amdgcn_update_dpp((int)(mine >> 32), (int)(mine >> 32), 0x111, 0xf, 0xf, false);
        const uint64_t left = ((uint64_t)lhi << 32) | llo;
        const int pos = __popc((uint32_t)((__ballot(j < topn && mine < key) >> rowbase) & 0xffffull));
        if (act && pos < topn) tkr[reg] = j < pos ? mine : (j == pos ? key : left);           // pos == topn: an earlier insertion raised the bar
      }
      thr[reg] = ((uint64_t)(uint32_t)__shfl((int)(tkr[reg] >> 32), rowbase + topn - 1, 64) << 32) |
                 (uint32_t)__shfl((int)(uint32_t)tkr[reg], rowbase + topn - 1, 64);
    }
    // ---- next tile: registers -> the other buffer (its last readers passed the barrier of the previous iteration)
    if (t + 1 < ntile) stash(buf ^ 1);
    __syncthreads();
  }
  // ---- this split's lists: lane (kq, j) holds element j of users 4 kq + reg
  if (j < topn) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t ur = u0 + 4 * kq + reg;
      if (ur < a.nq) a.part[(ur * a.nsplit + blockIdx.y) * topn + j] = tkr[reg];
    }
  }
}

// v.NV, |C0|^2, C0.NV, |NV|^2 per item with the arithmetic of pairs_l2_mc_kernel (8 lanes per row, chunks l, l + 8, ..., then
// xor-shuffles), once per pass
template <int NCH>
__global__ __launch_bounds__(256) void item_scalars_kernel(const float* __restrict__ C0, const float* __restrict__ C1,
                                                           const float* __restrict__ C2, int64_t n_items, float* __restrict__ ISC) {
  constexpr int D = 4 * NCH;
  const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool ok = row < n_items;
  const v4* r0 = reinterpret_cast<const v4*>(C0 + (ok ? row : 0) * D);
  const v4* r1 = reinterpret_cast<const v4*>(C1 + (ok ? row : 0) * D);
  const v4* r2 = reinterpret_cast<const v4*>(C2 + (ok ? row : 0) * D);
  v4 s0 = (v4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  for (int c = threadIdx.x & 7; c < NCH; c += 8) {
    const v4 x0 = r0[c], x1 = r1[c], x2 = r2[c];
    s0 += x1 * x2; s1 += x0 * x0; s2 += x0 * x2; s3 += x2 * x2;
  }
  float f0 = (s0[0] + s0[1]) + (s0[2] + s0[3]), f1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
  float f2 = (s2[0] + s2[1]) + (s2[2] + s2[3]), f3 = (s3[0] + s3[1]) + (s3[2] + s3[3]);
#pragma unroll
  for (int m = 1; m < 8; m <<= 1) {
    f0 += __shfl_xor(f0, m, 64); f1 += __shfl_xor(f1, m, 64); f2 += __shfl_xor(f2, m, 64); f3 += __shfl_xor(f3, m, 64);
  }
  if (ok && (threadIdx.x & 7) == 0) *reinterpret_cast<v4*>(ISC + row * 4) = (v4){f0, f1, f2, f3};
}

// partial lists of the splits -> the topn smallest keys per user; one thread per user (nsplit * topn <= a few hundred keys)
__global__ __launch_bounds__(256) void topk_merge_kernel(const uint64_t* __restrict__ part, int64_t nq, int nsplit, int topn,
                                                         const float* __restrict__ unused, int32_t* __restrict__ top_ids,
                                                         float* __restrict__ top_scores) {
  (void)unused;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= nq) return;
  const uint64_t* p = part + b * nsplit * topn;
  int head[8];                                                            // every split's list is sorted: a k-way merge
  for (int s = 0; s < nsplit; ++s) head[s] = 0;
  for (int r = 0; r < topn; ++r) {
    uint64_t best = PKEY_MAX;
    int bs = -1;
    for (int s = 0; s < nsplit; ++s) {
      if (head[s] < topn) {
        const uint64_t k = p[s * topn + head[s]];
        if (k < best) { best = k; bs = s; }
      }
    }
    if (bs >= 0) head[bs] += 1;
    const bool ok = best != PKEY_MAX;
    top_ids[b * topn + r] = ok ? (int32_t)(uint32_t)best : -1;
    if (top_scores) {
      uint32_t u = (uint32_t)(best >> 32);
      u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;                     // inverse of the order-preserving image
      top_scores[b * topn + r] = ok ? __uint_as_float(u) : 0.f;
    }
  }
}

template <typename G>
int launch_pass(PassArgs a, int32_t* top_ids, float* top_scores, hipStream_t st, const char* name) {
  const int64_t ublocks = (a.nq + 63) / 64;
  int nsplit = (int)(512 / ublocks);                                      // 2 workgroups per CU are resident: ONE round of <= 512
  if (nsplit > 8) nsplit = 8;
  const int64_t tiles = (a.n_items + IBT - 1) / IBT;
  if (nsplit > tiles) nsplit = (int)tiles;
  if (nsplit < 1) nsplit = 1;
  a.split_items = ((tiles + nsplit - 1) / nsplit) * IBT;
  nsplit = (int)((a.n_items + a.split_items - 1) / a.split_items);
  a.nsplit = nsplit;
  a.bm_words = (int)((a.split_items + 31) / 32);
  const size_t wave_bytes = (size_t)16 * 4 * 4 + (size_t)16 * a.bm_words * 4;
  const size_t lds = (size_t)2 * G::TILE_F4 * 16 + (size_t)2 * IBT * 4 * 4 + 4 * wave_bytes;
  if (lds > 160 * 1024) return 1;
  hipLaunchKernelGGL((item_scalars_kernel<G::NCH>), dim3((unsigned)((a.n_items + 31) / 32)), dim3(256), 0, st, a.C0, a.C1, a.C2, a.n_items,
                     const_cast<float*>(a.ISC));
  (void)hipFuncSetAttribute((const void*)eval_pass_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((eval_pass_kernel<G>), dim3((unsigned)ublocks, (unsigned)nsplit), dim3(256), lds, st, a);
  if (int e = check_launch(name)) return e;
  hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((a.nq + 255) / 256)), dim3(256), 0, st, a.part, a.nq, nsplit, a.topn,
                     (const float*)nullptr, top_ids, top_scores);
  return check_launch(name);
}

}  // namespace

// scratch behind `part`: the splits' partial lists, then the per-item scalars
size_t eval_pass_part_bytes(int64_t nq, int topn, int64_t n_items) {
  return (size_t)nq * 8 * topn * sizeof(uint64_t) + (size_t)n_items * 4 * sizeof(float);
}

// Returns KTUP_OK / an error, or 1 for shapes the fused pass does not cover (d, topn): the caller keeps the per-batch route.
int eval_pass_l2_mc(const float* QW, const float* C0, const float* C1, const float* C2, int d, int64_t nq, int64_t n_items,
                    const int64_t* filt_off, const int32_t* filt_ids, int topn, uint64_t* part, int32_t* top_ids, float* top_scores,
                    hipStream_t st, const char* name) {
  if ((d != 64 && d != 100 && d != 128) || topn < 1 || topn > TOPN_MAX || n_items >= (1ll << 31)) return 1;
  PassArgs a{};
  a.QW = QW; a.C0 = C0; a.C1 = C1; a.C2 = C2; a.nq = nq; a.n_items = n_items;
  a.filt_off = filt_off; a.filt_ids = filt_ids; a.topn = topn; a.part = part;
  a.ISC = reinterpret_cast<const float*>(part + (size_t)nq * 8 * topn);
  if (d == 64) return launch_pass<PGeom<16>>(a, top_ids, top_scores, st, name);
  if (d == 100) return launch_pass<PGeom<25>>(a, top_ids, top_scores, st, name);
  return launch_pass<PGeom<32>>(a, top_ids, top_scores, st, name);
}

}  // namespace ktup
