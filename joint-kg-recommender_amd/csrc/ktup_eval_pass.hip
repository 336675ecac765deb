// K16 + K17 fused for a whole evaluation pass (soft gate, squared L2): all-item TUP / KTUP scores AND the filtered top-n of
// every user in ONE sweep that never writes the (users x items) score matrix.
//
// Reference: transUP.py:84-102 / jTransUP.py:163-191 (evaluate / evaluateRec) produce a (B x N) matrix per batch of 512 users,
// which utils/misc.py:186-248 copies to the host, argsorts and walks.  Round 1 kept that shape on the device: per batch a
// six-GEMM score kernel wrote B x N x 4 bytes (78 MB over an ml1m pass) and a ranking kernel read them back -- 12 batches x
// ~5 launches, 1.05 ms.  Here:
//   * scores: |a|^2 - 2 s (a.n) + s^2 |n|^2 as in pairs_l2_mc_kernel (ktup_eval_mc.hip), but with every user x item cross term
//     except u.v contracted in PREFERENCE space (see QGeom below): 68 instead of 156 MFMAs per 16 x 16 tile at d = 100, P = 20.
//     Scores agree with the matrix route to fp32 rounding (the same sums in another association);
//   * a workgroup owns 64 users (4 waves x 16) and a contiguous split of the catalogue.  A wave keeps its 16 users' operand
//     rows in REGISTERS for the whole pass (17 float4 per lane); items stream through a double-buffered LDS buffer of
//     16-item tiles (the B operands, 160 floats per item) that the next tile's global loads refill under the MFMAs -- three
//     workgroups per CU;
//   * ranking: a user's sorted top-n list (64-bit keys = order-preserving score image << 32 | item id, the order of
//     ktup_rank.hip: ascending score, ties -> lower id) lives in the wave's LDS and is touched only when 16 candidates are pending
//     for it.  A score is a candidate if it is below the user's n-th score -- compared as floats, the keys deciding equality --
//     and its bit in the wave's filter bitmap (built once from the CSR filter lists, for this workgroup's item split only: 16 users
//     x (split / 32) words of LDS) is clear; candidates go to the user's pending row at positions taken from a ballot, and a row
//     that holds 16 is merged with the list by a fixed 16-lane network (bitonic sort of the candidates + merge: DPP exchanges on
//     the four rows of a register slot at once).  (History: lists in LDS with one insertion per wave at a time: 830 us per ml1m
//     pass; lists in registers with the network run per tile and slot: 212 us, 400 of its ~670 VALU instructions per tile in the
//     network; deferred merges: ~380 per tile, 192 us.  What is left is the shared MFMA + VALU pipe at the clock the chip holds
//     under this load (~2.1 GHz): option dbg_eval switches phases off to measure them -- 120 us without the ranking, 23 us of
//     prologue per workgroup.);
//   * both operand tables and the per-row scalars are built per pass by one small launch (pspace_rows_kernel) after the three
//     P x P Gram matrices (pspace_gram_kernel); the splits' partial lists are merged by a last, tiny launch (topk_merge_kernel).
// L1 distance and the ST-Gumbel gate do not decompose into bilinear terms: the hard gate has its own sweep (ktup_eval.hip
// sweep_hard_kernel, either distance); the soft gate with L1 keeps the per-batch kernels of ktup_eval.hip.
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
constexpr int PCAP = 32;       // pending candidates per user between two merges (a merge is due at 16; one tile adds at most 16)
// LDS per wave of the sweep: user scalars [16][4] | filter bits [16][bm_words] | pending candidates [16][PCAP] u64 | lists [16][16] u64 |
// n-th keys [16] u64
constexpr size_t WAVE_TAIL = (size_t)16 * PCAP * 8 + (size_t)16 * 16 * 8 + (size_t)16 * 8;
constexpr size_t wave_lds_bytes(int bm_words) { return (((size_t)16 * 4 * 4 + (size_t)16 * bm_words * 4 + 7) & ~(size_t)7) + WAVE_TAIL; }

KTUP_DEV uint64_t pass_key(float s, uint32_t id) {   // ktup_rank.hip make_key, ascending (lower score = better)
  if (s == 0.f) s = 0.f;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | id;
}

// ---- 16-lane row networks on 64-bit keys (lane j of a row = element j).  Partner j ^ K through DPP: quad_perm for 1 and 2,
// row_half_mirror . quad_perm[3,2,1,0] for 4 (7 - i then i ^ 3), row_mirror . row_half_mirror for 8.
template <int K>
KTUP_DEV uint32_t row_xor32(uint32_t v) {
  const int x = (int)v;
  if constexpr (K == 1) return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);
  else if constexpr (K == 2) return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);
  else if constexpr (K == 4) {
    const int h = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(h, h, 0x1B, 0xf, 0xf, false);
  } else {
    const int m = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(m, m, 0x141, 0xf, 0xf, false);
  }
}
// compare-exchange with lane j ^ K: keep the smaller key if keep_min, else the larger
template <int K>
KTUP_DEV void row_cmpx(uint64_t& v, bool keep_min) {
  const uint64_t o = ((uint64_t)row_xor32<K>((uint32_t)(v >> 32)) << 32) | row_xor32<K>((uint32_t)v);
  if ((o < v) == keep_min) v = o;
}
KTUP_DEV uint64_t row_mirror64(uint64_t v) {
  const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
  return ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false) << 32) |
         (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false);
}
// list: a row's ascending 16 keys; cand: up to 16 more in any order (PKEY_MAX = none).  Returns the 16 smallest of the 32, ascending:
// bitonic sort of the candidates (10 exchanges), elementwise min against their mirror (a bitonic row holding the 16 smallest),
// bitonic merge (4 exchanges) -- a fixed 14 exchanges instead of one dependent ballot / bpermute round per candidate.
KTUP_DEV uint64_t row_merge16(uint64_t list, uint64_t cand, int j) {
  const bool b1 = (j & 1) == 0, b2 = (j & 2) == 0, b4 = (j & 4) == 0, b8 = (j & 8) == 0;
  row_cmpx<1>(cand, b1 == b2);
  row_cmpx<2>(cand, b2 == b4); row_cmpx<1>(cand, b1 == b4);
  row_cmpx<4>(cand, b4 == b8); row_cmpx<2>(cand, b2 == b8); row_cmpx<1>(cand, b1 == b8);
  row_cmpx<8>(cand, b8); row_cmpx<4>(cand, b4); row_cmpx<2>(cand, b2); row_cmpx<1>(cand, b1);
  const uint64_t r = row_mirror64(cand);
  uint64_t m = r < list ? r : list;
  row_cmpx<8>(m, b8); row_cmpx<4>(m, b4); row_cmpx<2>(m, b2); row_cmpx<1>(m, b1);
  return m;
}

// partial lists of the splits -> the topn smallest keys per user.  One WAVE per user: the <= 128 keys sit two per lane, every lane
// ranks its keys against all of them (keys are distinct: the item id is their low half) and the lanes whose rank is below topn
// write their key's slot -- no serial k-way merge, no dependent memory round trips (that version: 22 us for 6040 users).
constexpr int MERGE_T = 256;
__global__ __launch_bounds__(MERGE_T) void topk_merge_kernel(const uint64_t* __restrict__ part, int64_t nq, int nsplit, int topn,
                                                             const float* __restrict__ unused, int32_t* __restrict__ top_ids,
                                                             float* __restrict__ top_scores) {
  (void)unused;
  __shared__ uint64_t wk[MERGE_T / 64][128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * (MERGE_T / 64) + w;
  if (b >= nq) return;                                                    // (whole waves leave: no workgroup barrier below)
  const int per = nsplit * topn;
  const uint64_t* p = part + b * per;
  const uint64_t k0 = lane < per ? p[lane] : PKEY_MAX, k1 = 64 + lane < per ? p[64 + lane] : PKEY_MAX;
  wk[w][lane] = k0; wk[w][64 + lane] = k1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int r0 = 0, r1 = 0;
  for (int i = 0; i < per; ++i) {
    const uint64_t k = wk[w][i];
    r0 += k < k0; r1 += k < k1;
  }
  const int valid = __popcll(__ballot(k0 != PKEY_MAX)) + __popcll(__ballot(k1 != PKEY_MAX));
  auto put = [&](uint64_t key, int r) {
    if (key == PKEY_MAX || r >= topn) return;
    top_ids[b * topn + r] = (int32_t)(uint32_t)key;
    if (top_scores) {
      uint32_t u = (uint32_t)(key >> 32);
      u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;                     // inverse of the order-preserving image
      top_scores[b * topn + r] = __uint_as_float(u);
    }
  };
  put(k0, r0); put(k1, r1);
  if (lane >= valid && lane < topn) {                                     // fewer candidates than topn: pad
    top_ids[b * topn + lane] = -1;
    if (top_scores) top_scores[b * topn + lane] = 0.f;
  }
}

// The filter lists of a pass as bits, built ONCE per pass instead of once per catalogue split inside the sweep (where the walk was 15 of
// the sweep's 180 us: every split's workgroup of a user block repeated it, latency-bound, before its first tile): one wave per 16
// consecutive users walks their contiguous run of the CSR ids -- all 64 lanes together, sixteen loads in flight per lane, an entry's
// row found by comparing its position with the 15 inner offsets --, sets bit (id - split's first item) of word [row][split][.] in
// LDS and stores the 16 rows as one contiguous run: out[(user * nsplit + split) * bm_words + word].  The waves ride as extra
// workgroups of the Gram launch (which the sweep waits for anyway).
struct FiltBm {
  const int64_t* off; const int32_t* ids; int64_t nq, n_items; uint32_t* out; int nsplit, bm_words; uint32_t split_items; int gram_blocks;
};

KTUP_DEV void filter_bitmap_wave(const FiltBm& f, int64_t u0, uint32_t* lbm, int lane) {
  const int wpu = f.nsplit * f.bm_words;
  for (int idx = lane; idx < 16 * wpu; idx += 64) lbm[idx] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t uo = u0 + (lane < 16 ? lane : 16);
  const int64_t myoff = f.off[uo < f.nq ? uo : f.nq];
  const int64_t f_begin = __shfl(myoff, 0, 64), f_end = __shfl(myoff, 16, 64);
  uint32_t rel[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) rel[k] = (uint32_t)(__shfl(myoff, k + 1, 64) - f_begin);
  constexpr int FB = 16;                                               // loads in flight per lane
  for (int64_t base = f_begin; base < f_end; base += 64 * FB) {
    int32_t ids[FB];
#pragma unroll
    for (int k = 0; k < FB; ++k) {
      const int64_t e = base + lane + 64 * k;
      ids[k] = e < f_end ? f.ids[e] : -1;
    }
#pragma unroll
    for (int k = 0; k < FB; ++k) {
      const uint32_t pos = (uint32_t)(base - f_begin) + lane + 64 * k;
      int r = 0;
#pragma unroll
      for (int q = 0; q < 15; ++q) r += pos >= rel[q] ? 1 : 0;
      if (ids[k] >= 0 && ids[k] < f.n_items) {
        const uint32_t id = (uint32_t)ids[k], sp = id / f.split_items, lid = id - sp * f.split_items;
        atomicOr(lbm + r * wpu + sp * f.bm_words + (lid >> 5), 1u << (lid & 31));
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t* o = f.out + u0 * wpu;                                      // (sized for whole 64-user blocks: rows past nq are all-zero)
  for (int idx = lane; idx < 16 * wpu; idx += 64) o[idx] = lbm[idx];
}

// ================================================================================================ preference-space formulation
// Six d-long products per pair (u.NV, NU.v, AU.C0, AU.NV, NU.C0, NU.NV: round 2's first fused pass, 0.39 ms per ml1m sweep) are
// more work than the model needs.  With the soft gate every vector of the score lives in
//   span{x} + span{Ar_p} + span{Cn_p}   (x = u or v;  RU = sum_p L_p Ar_p,  NU = sum_p L_p Cn_p,  L = Alog x, P <= 32 preferences)
// so every user x item cross term except u.v is a P-long contraction of per-row P-vectors (L, Rx = Ar x, Nx = Cn x) through the
// three P x P Gram matrices G_RR = Ar Ar^T, G_RN = Ar Cn^T, G_NN = Cn Cn^T:
//   s  cross = u.NV - v.NU            = [ Nu ; -L ]                          . [ LV ; Nv ]          K = 2 P
//   an cross = AU.NV - C0.NU          = [ Nu + G_RN^T L + G_RN L ; -L ]      . [ LV ; Nv ]          K = 2 P   (same B)
//   nn cross = 2 NU.NV                = [ 2 G_NN L ]                         . [ LV ]               K = P     (same B)
//   aa cross = -2 AU.C0               = [ -2u ; -2L ; 2 (Ru + G_RR L) ]      . [ v ; Rv ; LV ]      K = d + 2 P
// with the Gram products folded into the USER operand once per pass.  At d = 100, P = 20: 68 MFMAs (K padded to 16-blocks) and
// 12 ds_read_b128 per 16 x 16 tile instead of 156 and 75; an item row is 160 floats instead of 300.  Scores agree with the
// d-space kernels to fp32 rounding (a different association of the same sums), not bit for bit.
template <int NCH_, int NP_>
struct QGeom {
  static constexpr int NCH = NCH_, D = 4 * NCH, NP = NP_, P4 = 4 * NP;
  static constexpr int ROWB = D + 3 * P4;                       // item row: [x ; Rx ; L ; Nx]
  static constexpr int RB4 = ROWB / 4;
  // An item row is [x ; Rx ; L ; Nx | its 4 scalars (| one zero float4)] at an ODD float4 pitch (16 rows x 4 k-quads tile the LDS banks)
  // -- in memory exactly as in LDS, so a tile of 16 consecutive items is ONE contiguous run of IBT * ROW4 float4 and travels by
  // LDS-DMA (global_load_lds_dwordx4: destination = a wave-uniform LDS address + 16 B x lane): no staging registers (they were 12 of
  // the lane's 168, with 9 more for the lane-constant source / destination addresses: 18 spilled), no ds_write pass (whose 16-lane
  // groups straddled the row pads: ~16 two-way bank conflicts per tile)
  static constexpr int ROW4 = (RB4 + 1) | 1;
  static constexpr int GROW = 4 * ROW4;                         // floats per item row
  static constexpr int TSLOTS = IBT * ROW4;                     // float4 per tile
  static constexpr int DPW = ((TSLOTS + 63) / 64 + 3) / 4;      // DMA instructions per wave and tile (wave w issues pieces w, w + 4, ...)
  static constexpr int SOFF4 = (D + P4) / 4;                    // float4 offset of [L ; Nx] inside a row
  static constexpr int KA = (D + 2 * P4 + 15) / 16, KS = (2 * P4 + 15) / 16, KN = (P4 + 15) / 16;   // 16-blocks of K
  static constexpr int NA = KA + 2 * KS + KN;                   // float4 per lane and 16-block of a user row in memory: [AA | S | AN | NN]
  static constexpr int AROW = 16 * NA;                          // floats per user row
  // What the sweep multiplies is the REAL K of each product: whole 16-blocks (one ds_read_b128 and four MFMAs each, k = 16 g + 4 kq + c)
  // and then the remaining k-quads one MFMA each (one ds_read_b32, k = 16 F + 4 m + kq) -- the zero padding of the last 16-blocks
  // was 8 of 68 MFMAs per tile at d = 100, P = 20
  static constexpr int FA = (D + 2 * P4) / 16, TA = ((D + 2 * P4) % 16) / 4;     // AA: [x ; Rx ; L]
  static constexpr int FS = (2 * P4) / 16, TS = ((2 * P4) % 16) / 4;             // S and AN: [L ; Nx]
  static constexpr int FN = P4 / 16, TN = (P4 % 16) / 4;                         // NN: [L]
  static constexpr int NREG = 4 * FA + TA + 2 * (4 * FS + TS) + 4 * FN + TN;     // A-operand registers per lane
  static constexpr int SUB = 1;                                 // 16-item sub-tiles per buffer = per workgroup barrier (2: no faster, and the
                                                                // pending-candidate buffers below need the LDS for three workgroups per CU)
  static constexpr int TILE_F4 = SUB * IBT * ROW4 + 4;          // + pad: the padded K blocks read a little past the last row
#ifndef KTUP_EVAL_MINW3
#define KTUP_EVAL_MINW3 3
#endif
  static constexpr int MINW = NREG <= 64 ? KTUP_EVAL_MINW3 : 2;                 // waves per SIMD the register budget allows (= workgroups per CU)
};

struct QArgs {
  const float *A, *SCU;          // users: AROW floats + 4 scalars each
  const float* B;                // items: rows of GROW floats (operands, then the 4 scalars)
  int64_t nq, n_items;
  const int64_t* filt_off; const int32_t* filt_ids;
  int topn, nsplit; int64_t split_items; uint64_t* part; int bm_words;
  int32_t* gthr;                 // [nq (+ pad to whole 64-user blocks)] the best n-th score any split of a user has reached, as the bits of a
                                 // non-negative float (0x7fffffff: none yet); see the tile loop
  const uint32_t* bmg;           // the filter lists as bits, [user][split][bm_words] (filter_bitmap_wave); NULL: the sweep walks the lists itself
  int dbg;                       // MEASUREMENT ONLY (option dbg_eval): 1 no ranking epilogue, 2 no item loads, 4 no workgroup barrier per tile, 8 no tiles at all
};

template <typename G>
__global__ __launch_bounds__(256, G::MINW) void eval_pass_q_kernel(QArgs a) {
  constexpr int RB4 = G::RB4, ROW4 = G::ROW4, KA = G::KA, KS = G::KS, DPW = G::DPW;
  constexpr int FA = G::FA, TA = G::TA, FS = G::FS, TS = G::TS, FN = G::FN, TN = G::TN;
  static_assert(G::SUB == 1, "one 16-item tile per buffer");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* Xb = reinterpret_cast<v4*>(smem);                                   // [2][TILE_F4] item tiles
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* wbase = reinterpret_cast<char*>(Xb + 2 * G::TILE_F4) + (size_t)w * wave_lds_bytes(a.bm_words);
  float* usc = reinterpret_cast<float*>(wbase);                           // [16][4] user scalars
  uint32_t* bm = reinterpret_cast<uint32_t*>(usc + 64);                   // [16][bm_words] filter bits of this split
  uint64_t* pbuf = reinterpret_cast<uint64_t*>(wbase + wave_lds_bytes(a.bm_words) - WAVE_TAIL);   // [16][PCAP] pending candidates
  uint64_t* tk = pbuf + 16 * PCAP;                                        // [16][16] the users' sorted lists (touched by merges only)
  uint64_t* thrk = tk + 16 * 16;                                          // [16] their n-th keys (read where floats cannot decide)
  // Which (user block, split) this workgroup takes: workgroups go to the 8 XCDs round-robin in launch order and every XCD has an L2 of
  // its own, so with (blockIdx.x, blockIdx.y) taken as they come the 8 splits of a user block sit on 8 different XCDs and EVERY L2
  // fetches EVERY user row (8 x 6.6 MB over the fabric at ml1m size, ~10 us of prologue).  XCD x (launch-order ids x, x + 8, ...) takes
  // a contiguous run of the pairs ordered user block major instead: its L2 then holds ~1/8 of the user rows (the item rows it needs
  // entirely either way).
  int ub, sp;
  {
    const uint32_t L = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y;
    const uint32_t x = L & 7u, per = total >> 3, rem = total & 7u;
    const uint32_t q = x * per + (x < rem ? x : rem) + (L >> 3);
    ub = (int)(q / (uint32_t)a.nsplit);
    sp = (int)(q - (uint32_t)ub * (uint32_t)a.nsplit);
  }
  const int64_t u0 = (int64_t)ub * 64 + 16 * w;
  const int64_t i_lo = (int64_t)sp * a.split_items;
  const int64_t i_hi = min(a.n_items, i_lo + a.split_items);
  const int topn = a.topn;
  // the wave's filter bits from the pass's bitmap (filter_bitmap_wave): requested first, stored after the other set-up work below
  constexpr int BMR = 8;                                                  // words per lane held in flight (more: a second trip)
  const bool use_bmg = a.bmg != nullptr && a.filt_off && !(a.dbg & 16);
  uint32_t bmv[BMR];
  if (use_bmg) {
    const int nw = 16 * a.bm_words;
#pragma unroll
    for (int k = 0; k < BMR; ++k) {
      const int idx = lane + 64 * k, r = idx / a.bm_words;
      bmv[k] = idx < nw ? a.bmg[((u0 + r) * a.nsplit + sp) * a.bm_words + (idx - r * a.bm_words)] : 0u;
    }
  }
  for (int idx = tid; idx < 2 * G::TILE_F4; idx += 256) Xb[idx] = (v4){0.f, 0.f, 0.f, 0.f};   // row pads stay finite (x 0 operands)
  if (!use_bmg)
    for (int idx = lane; idx < 16 * a.bm_words; idx += 64) bm[idx] = 0u;
  for (int idx = lane; idx < 16 * 16; idx += 64) tk[idx] = PKEY_MAX;
  if (lane < 16) thrk[lane] = u0 + lane < a.nq ? PKEY_MAX : 0;            // rows past the end: nothing is ever below
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (a.filt_off && !use_bmg && !(a.dbg & 16)) {
    // (No bitmap of the pass -- it would not fit the scratch cap: every split's workgroup walks the lists itself.)
    // The 16 users of a wave are consecutive, so their filter lists are ONE contiguous run of the CSR ids: all 64 lanes walk it together
    // (sixteen independent loads in flight per lane) and find an entry's row by comparing its position with the 15 inner offsets.  (One list
    // after the other -- 48 dependent round trips -- was 18 of the sweep's 200 us, every split's workgroups repeating it; now 11.)
    const int64_t uo = u0 + (lane < 16 ? lane : 16);
    const int64_t myoff = a.filt_off[uo < a.nq ? uo : a.nq];
    const int64_t f_begin = __shfl(myoff, 0, 64), f_end = __shfl(myoff, 16, 64);
    uint32_t rel[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) rel[k] = (uint32_t)(__shfl(myoff, k + 1, 64) - f_begin);
    const int64_t span = i_hi - i_lo;
    constexpr int FB = 16;                                               // loads in flight per lane
    for (int64_t base = f_begin; base < f_end; base += 64 * FB) {
      int32_t ids[FB];
#pragma unroll
      for (int k = 0; k < FB; ++k) {
        const int64_t f = base + lane + 64 * k;
        ids[k] = f < f_end ? a.filt_ids[f] : -1;
      }
#pragma unroll
      for (int k = 0; k < FB; ++k) {
        const int64_t id = (int64_t)ids[k] - i_lo;
        const uint32_t pos = (uint32_t)(base - f_begin) + lane + 64 * k;
        int r = 0;
#pragma unroll
        for (int q = 0; q < 15; ++q) r += pos >= rel[q] ? 1 : 0;
        if (ids[k] >= 0 && id >= 0 && id < span) atomicOr(bm + r * a.bm_words + (id >> 5), 1u << (id & 31));
      }
    }
  }
  if (lane < 16) *reinterpret_cast<v4*>(usc + lane * 4) =
      u0 + lane < a.nq ? *reinterpret_cast<const v4*>(a.SCU + (u0 + lane) * 4) : (v4){0.f, 0.f, 0.f, 0.f};
  // A operands of the whole pass (user j): k-quad kq of every whole 16-block, then element kq of every remaining k-quad
  v4 aAA[FA > 0 ? FA : 1], aS[FS > 0 ? FS : 1], aAN[FS > 0 ? FS : 1], aNN[FN > 0 ? FN : 1];
  float tAA[TA > 0 ? TA : 1], tS[TS > 0 ? TS : 1], tAN[TS > 0 ? TS : 1], tNN[TN > 0 ? TN : 1];
  {
    const bool ok = u0 + j < a.nq && !(a.dbg & 32);
    const float* rf = a.A + (ok ? u0 + j : 0) * G::AROW;
    const v4* r0 = reinterpret_cast<const v4*>(rf);
    constexpr int oS = 16 * KA, oAN = oS + 16 * KS, oNN = oAN + 16 * KS;
    const v4 z4 = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < FA; ++g) aAA[g] = ok ? r0[4 * g + kq] : z4;
#pragma unroll
    for (int m = 0; m < TA; ++m) tAA[m] = ok ? rf[16 * FA + 4 * m + kq] : 0.f;
#pragma unroll
    for (int g = 0; g < FS; ++g) { aS[g] = ok ? r0[oS / 4 + 4 * g + kq] : z4; aAN[g] = ok ? r0[oAN / 4 + 4 * g + kq] : z4; }
#pragma unroll
    for (int m = 0; m < TS; ++m) { tS[m] = ok ? rf[oS + 16 * FS + 4 * m + kq] : 0.f; tAN[m] = ok ? rf[oAN + 16 * FS + 4 * m + kq] : 0.f; }
#pragma unroll
    for (int g = 0; g < FN; ++g) aNN[g] = ok ? r0[oNN / 4 + 4 * g + kq] : z4;
#pragma unroll
    for (int m = 0; m < TN; ++m) tNN[m] = ok ? rf[oNN + 16 * FN + 4 * m + kq] : 0.f;
  }
  if (use_bmg) {
    const int nw = 16 * a.bm_words;
#pragma unroll
    for (int k = 0; k < BMR; ++k)
      if (lane + 64 * k < nw) bm[lane + 64 * k] = bmv[k];
    for (int idx = lane + 64 * BMR; idx < nw; idx += 64) {                // (splits of more than 1024 items)
      const int r = idx / a.bm_words;
      bm[idx] = a.bmg[((u0 + r) * a.nsplit + sp) * a.bm_words + (idx - r * a.bm_words)];
    }
  }
  __syncthreads();                                                        // tiles zeroed, bitmaps and scalars in place
  const int64_t ntile = (a.dbg & 8) ? 0 : (i_hi - i_lo + IBT - 1) / IBT;    // (dbg 8: prologue and epilogue only)
  // this lane's float4 of every DMA piece of its wave: byte offset inside a tile (tiles past the table's end re-read its last row:
  // their items are masked out below, the operands only have to be finite)
  uint32_t soff[DPW];
#pragma unroll
  for (int k = 0; k < DPW; ++k) {
    const int slot = 64 * (w + 4 * k) + lane;
    soff[k] = slot < G::TSLOTS ? (uint32_t)slot * 16u : 0xffffffffu;
  }
  auto dma = [&](int64_t t, int buf) {                                    // tile t of this split -> LDS buffer `buf`
    const int64_t row0 = i_lo + t * IBT;
    const char* src = reinterpret_cast<const char*>(a.B + row0 * G::GROW);
    v4* dst = Xb + buf * G::TILE_F4 + 64 * w;
    const bool tail = row0 + IBT > a.n_items;                             // (uniform)
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
      if (soff[k] == 0xffffffffu) continue;
      const char* from = src + soff[k];
      if (tail) {
        const uint32_t slot = soff[k] >> 4, row = slot / ROW4, c = slot - row * ROW4;
        const int64_t r = row0 + row < a.n_items ? row0 + row : a.n_items - 1;
        from = reinterpret_cast<const char*>(a.B + r * G::GROW + 4 * c);
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)from,
                                       (__attribute__((address_space(3))) void*)(dst + 256 * k), 16, 0, 0);
    }
  };
  const int rowbase = 16 * kq;
  // A user's list is touched only when 16 candidates are pending for it.  A score is a candidate if it is below the user's n-th score
  // -- compared as FLOATS (thrf; NaN while the list is short: every score then takes the key compare below); equality or a NaN sends
  // the wave through the 64-bit key compare, so the order is that of the keys in every case.  Candidates are appended to the user's
  // pending row in LDS at positions taken from a ballot (no atomics; `pend` is replicated over the row's 16 lanes); once a row of a
  // register slot holds 16, that slot's four rows go through the merge network -- 16 real candidates per network pass instead of the
  // ~2 a tile yields -- and the thresholds are renewed.  (The network per tile and slot was 400 of this kernel's ~670 VALU
  // instructions per tile.)
  // The splits of a user share their bounds (a.gthr): any split's n-th score bounds the user's final n-th score from above, so a
  // score above it is no candidate anywhere (a split's list may then end short: the merge pads).  Published at every merge, re-read
  // once per tile -- requested before the tile's MFMAs (past the L1: the line changes under the kernel), used after them.  A stale
  // value is an older bound, never a wrong one.  With the XCD-aware mapping above a user's splits share one L2.  Each split alone
  // sends ~47 candidates per user through the append / merge path below, together ~4x fewer.
  float thrf[4];
  int pend[4] = {0, 0, 0, 0};
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) thrf[reg] = u0 + 4 * kq + reg < a.nq ? __uint_as_float(0x7fffffffu) : -__builtin_inff();
  const uint32_t lt_j = (1u << j) - 1u;
  const int rsh = 16 * (kq & 1);
  const bool rhi = (kq & 2) != 0;
  auto flush = [&](bool all) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int n = pend[reg];
      if (!__builtin_amdgcn_ballot_w64(all ? n > 0 : n >= 16)) continue;
      const int ur = 4 * kq + reg;
      const uint64_t* row = pbuf + ur * PCAP;
      uint64_t merged = row_merge16(tk[ur * 16 + j], j < n ? row[j] : PKEY_MAX, j);   // all four rows of the slot at once
      if (__builtin_amdgcn_ballot_w64(n > 16)) merged = row_merge16(j < topn ? merged : PKEY_MAX, 16 + j < n ? row[16 + j] : PKEY_MAX, j);
      merged = j < topn ? merged : PKEY_MAX;
      tk[ur * 16 + j] = merged;
      pend[reg] = 0;
      if (u0 + ur < a.nq) {                                                       // (rows past the end keep thrf = -inf, thrk = 0)
        const uint32_t hi = (uint32_t)__shfl((int)(merged >> 32), rowbase + topn - 1, 64);
        const float own = __uint_as_float((hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi);   // inverse of the order-preserving image (NaN: list short)
        thrf[reg] = own;                                                          // (the next tile's refresh brings the shared bound back)
        if (j == topn - 1) {
          thrk[ur] = merged;
          // the splits of a user tell each other: non-negative floats order like their bits (a negative n-th score -- rounding of a
          // squared distance -- is published as 0: a weaker bound, still one)
          if (a.gthr && own == own) atomicMin(a.gthr + u0 + ur, __float_as_int(__builtin_fmaxf(own, 0.f)));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto compute = [&](int buf, int sub, int64_t t, const int (&gl)[4]) {   // 16 users x the 16 items of tile t
    const v4* ib = Xb + buf * G::TILE_F4 + (sub * IBT + j) * ROW4 + kq;   // lane (kq, item j): k-quad kq of every 16-block
    v4 accAA = (v4){0.f, 0.f, 0.f, 0.f}, accS = accAA, accAN = accAA, accNN = accAA;
    // B operands one step ahead of the MFMAs that use them (two register sets): left to itself the compiler re-uses ONE set and
    // every group of MFMAs waits out an LDS round trip (measured: the MFMAs + reads alone took 120 us of a 68 us pipe time).
    // Steps: AA's whole blocks, AA's remaining quads, the [L ; Nx] blocks (S, AN and -- the first FN -- NN), their remaining quads
    // (S, AN), NN's remaining quads.
    const float* ibf = reinterpret_cast<const float*>(Xb + buf * G::TILE_F4 + (sub * IBT + j) * ROW4) + kq;   // element kq of a k-quad
    constexpr int S0 = FA, S1 = S0 + TA, S2 = S1 + FS, S3 = S2 + TS, NSTEP = S3 + TN;
    constexpr int SF = 4 * G::SOFF4;                                       // float offset of [L ; Nx] inside a row
    auto bsrc = [&](int s) -> v4 {
      if (s < S0) return ib[4 * s];
      if (s < S1) return (v4){ibf[16 * FA + 4 * (s - S0)], 0.f, 0.f, 0.f};
      if (s < S2) return ib[G::SOFF4 + 4 * (s - S1)];
      if (s < S3) return (v4){ibf[SF + 16 * FS + 4 * (s - S2)], 0.f, 0.f, 0.f};
      return (v4){ibf[SF + 16 * FN + 4 * (s - S3)], 0.f, 0.f, 0.f};
    };
    v4 bcur = bsrc(0);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      v4 bnext = bcur;
      if (s + 1 < NSTEP) bnext = bsrc(s + 1);
      __builtin_amdgcn_sched_barrier(0);                                   // the read stays ahead of the MFMAs below
      if (s < S0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) accAA = __builtin_amdgcn_mfma_f32_16x16x4f32(aAA[s][c], bcur[c], accAA, 0, 0, 0);
      } else if (s < S1) {
        accAA = __builtin_amdgcn_mfma_f32_16x16x4f32(tAA[s - S0], bcur[0], accAA, 0, 0, 0);
      } else if (s < S2) {
        const int g = s - S1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          accS = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[g][c], bcur[c], accS, 0, 0, 0);
          accAN = __builtin_amdgcn_mfma_f32_16x16x4f32(aAN[g][c], bcur[c], accAN, 0, 0, 0);
          if (g < FN) accNN = __builtin_amdgcn_mfma_f32_16x16x4f32(aNN[g][c], bcur[c], accNN, 0, 0, 0);
        }
      } else if (s < S3) {
        accS = __builtin_amdgcn_mfma_f32_16x16x4f32(tS[s - S2], bcur[0], accS, 0, 0, 0);
        accAN = __builtin_amdgcn_mfma_f32_16x16x4f32(tAN[s - S2], bcur[0], accAN, 0, 0, 0);
      } else {
        accNN = __builtin_amdgcn_mfma_f32_16x16x4f32(tNN[s - S3], bcur[0], accNN, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      bcur = bnext;
    }
    const v4 is4 = Xb[buf * G::TILE_F4 + (sub * IBT + j) * ROW4 + RB4];      // the row's scalars ride in its tile slot
    const int64_t item = i_lo + t * IBT + j;
    const int64_t lid = item - i_lo;
    const bool iok = item < i_hi;
    bool full = false;
    if (a.dbg & 1) {                                                       // measurement: keep the MFMAs alive, skip the ranking
      if (accS[0] + accAA[1] + accAN[2] + accNN[3] == 12345.f) thrf[0] = is4[0];
      return;
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int ur = 4 * kq + reg;
      // (as integers: gl is a non-negative float's bits or 0x7fffffff, and a bound of this split that is negative, -inf or a NaN of
      //  either sign keeps the meaning it has without the shared word)
      thrf[reg] = __int_as_float(min(__float_as_int(thrf[reg]), gl[reg]));
      const v4 us4 = *reinterpret_cast<const v4*>(usc + ur * 4);
      const float sv = (us4[0] + accS[reg]) - is4[0];
      const float aa = (us4[1] + is4[1]) + accAA[reg];
      const float an = (us4[2] + accAN[reg]) - is4[2];
      const float nn = (us4[3] + is4[3]) + accNN[reg];
      const float score = fmaf(sv * sv, nn, fmaf(-2.f * sv, an, aa));
      // one compare whose result stays a wave mask; with the shared bounds many 64-score slots have no candidate and leave here
      const uint64_t m_ngt = __builtin_amdgcn_ballot_w64(!(score > thrf[reg]));   // below the bound, equal to it, or unordered
      if (m_ngt == 0) continue;
      bool c = score < thrf[reg];
      if (m_ngt != __builtin_amdgcn_ballot_w64(c)) {                       // the keys decide (always while the list is short)
        if (!c && !(score > thrf[reg])) c = pass_key(score, (uint32_t)item) < thrk[ur];
      }
      c = c && iok;
      if (c) c = ((bm[ur * a.bm_words + (lid >> 5)] >> (lid & 31)) & 1u) == 0u;
      const uint64_t m = __builtin_amdgcn_ballot_w64(c);
      if (m) {
        const uint32_t rb = ((rhi ? (uint32_t)(m >> 32) : (uint32_t)m) >> rsh) & 0xffffu;   // the candidates of this lane's row
        if (c) pbuf[ur * PCAP + pend[reg] + __popc(rb & lt_j)] = pass_key(score, (uint32_t)item);
        pend[reg] += __popc(rb);
        full |= pend[reg] >= 16;
      }
    }
    if (__builtin_amdgcn_ballot_w64(full)) flush(false);
  };
  if (ntile > 0) {
    if (!(a.dbg & 2)) dma(0, 0);
    __syncthreads();                                                       // (carries the vmcnt(0) that lands the DMA)
  }
  // one tile per workgroup barrier; the next tile's DMA is in flight under this tile's MFMAs.  Buffer buf ^ 1 was last read in the
  // previous iteration, whose closing barrier every wave has passed; its new contents are read after this iteration's barrier.
  for (int64_t t0 = 0; t0 < ntile; ++t0) {
    const int buf = (int)(t0 & 1);
    if (t0 + 1 < ntile && !(a.dbg & 2)) dma(t0 + 1, buf ^ 1);
    int gl[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    if (a.gthr) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) gl[reg] = __hip_atomic_load(a.gthr + u0 + 4 * kq + reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    compute((a.dbg & 2) ? 0 : buf, 0, t0, gl);
    if (!(a.dbg & 4)) __syncthreads();
  }
  flush(true);
  if (j < topn) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t ur = u0 + 4 * kq + reg;
      if (ur < a.nq) a.part[(ur * a.nsplit + sp) * topn + j] = tk[(4 * kq + reg) * 16 + j];
    }
  }
}

// G_RR | G_RN | G_NN, each [P4][P4] (zero padded), from the prepared tables (row pitch dp): one wave per entry.  gs (optional):
// the same numbers as the A operands of pspace_rows_mc_kernel's fold products, [matrix G_RR, G_RN, G_RN^T, G_NN][pt][k-step m][lane]
// with lane (k, i) <-> row 16 pt + 4 (i & 3) + (i >> 2), column 16 (m >> 2) + 4 (m & 3) + k; rows >= P4 are zero.
KTUP_DEV int gs_index(int mtx, int p, int q, int PT, int NP) {
  const int pt = p >> 4, pr = p & 15, i = (pr >> 2) + 4 * (pr & 3);
  const int m = (q >> 4) * 4 + ((q & 15) >> 2), k = q & 3;
  return ((mtx * PT + pt) * NP + m) * 64 + k * 16 + i;
}

__global__ __launch_bounds__(256) void pspace_gram_kernel(const float* __restrict__ Ar, const float* __restrict__ Cn, int dp, int d, int P, int P4,
                                                          float* __restrict__ grams, float* __restrict__ gs, int PT, int NP, FiltBm fb) {
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x >= fb.gram_blocks) {                       // the pass's filter bitmap: four waves x 16 users per extra workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    filter_bitmap_wave(fb, ((int64_t)((int)blockIdx.x - fb.gram_blocks) * 4 + w) * 16,
                       reinterpret_cast<uint32_t*>(smem) + (size_t)w * 16 * fb.nsplit * fb.bm_words, lane);
    return;
  }
  const int gridx = fb.gram_blocks;
  for (int idx = blockIdx.x * 4 + (threadIdx.x >> 6); idx < 3 * P4 * P4; idx += gridx * 4) {
    const int t = idx / (P4 * P4), rem = idx - t * P4 * P4, p = rem / P4, q = rem - p * P4;
    float acc = 0.f;
    if (p < P && q < P) {
      const float* x = (t == 2 ? Cn : Ar) + (size_t)p * dp;
      const float* y = (t == 0 ? Ar : Cn) + (size_t)q * dp;
      for (int k = lane; k < d; k += 64) acc = fmaf(x[k], y[k], acc);
    }
    acc = group_sum<64>(acc);
    if (lane == 0) {
      grams[idx] = acc;
      if (gs) {
        if (t == 1) { gs[gs_index(1, p, q, PT, NP)] = acc; gs[gs_index(2, q, p, PT, NP)] = acc; }
        else gs[gs_index(t == 0 ? 0 : 3, p, q, PT, NP)] = acc;
      }
    }
  }
  if (gs && 16 * PT > P4)                                        // rows P4 .. 16 PT - 1 of the last row tile: zero
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < 4 * (16 * PT - P4) * P4; idx += gridx * 256) {
      const int mtx = idx / ((16 * PT - P4) * P4), rem = idx - mtx * (16 * PT - P4) * P4;
      gs[gs_index(mtx, P4 + rem / P4, rem % P4, PT, NP)] = 0.f;
    }
}

// One wave per row x (user: U[ids[row]]; item: X[row]): L = Alog x, Rx = Ar x, Nx = Cn x (lane = one of the 3 P4 table rows, the
// tables staged transposed in LDS), then the operand row and the four scalars of the formulation above.
struct RowsSide {
  const float* X; int64_t ldx; const int64_t* ids;   // rows X[ids[row]] (ids == NULL: X[row])
  const float* E; int64_t lde; const int32_t* map;    // + E[map[row]] (KTUP items: the aligned entity, the pad row being zero); NULL = none
  int64_t nrows; float* out; int orow; float* scal; int blocks;
  int opitch, spitch;                                 // floats between two rows of out / of scal (items: both G::GROW, scal = out + ROWB)
  int32_t* gthr;                                      // users: the sweep's shared n-th score per user, reset here (NULL: none / items)
};

template <bool IS_USER, int NCH, int NP>
KTUP_DEV void pspace_rows(const RowsSide& sd, int block, int P, const float* __restrict__ Alog, const float* __restrict__ Ar,
                          const float* __restrict__ Cn, int dp, const float* __restrict__ grams, int ka16, int ks16) {
  const float* __restrict__ X = sd.X;
  const int64_t ldx = sd.ldx, nrows = sd.nrows;
  const int64_t* __restrict__ ids = sd.ids;
  float* __restrict__ out = sd.out;
  float* __restrict__ scal = sd.scal;
  const int orow = sd.orow;
  constexpr int d = 4 * NCH, P4 = 4 * NP;                        // compile-time: the k loops unroll and their LDS reads batch
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // the tables (3 P x d floats = 24 KB at P = 20, d = 100) and the Gram matrices are read straight from global memory: they stay in
  // the CU's vector L1, and staging them in LDS per workgroup cost more than the rows' own work
  float* wv = lds;                                               // per wave: x [d] | res [3 P4] | fold [4 P4]
  constexpr int wstride = d + 7 * P4;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* G = grams;
  float* xs = wv + w * wstride;
  float* res = xs + d;
  float* fold = res + 3 * P4;
  for (int64_t row = (int64_t)block * 4 + w; row < nrows; row += (int64_t)sd.blocks * 4) {
    const float* x = X + (ids ? ids[row] : row) * ldx;
    const float* e = sd.E ? sd.E + (int64_t)sd.map[row] * sd.lde : nullptr;
    float sq = 0.f;
    for (int k = lane; k < d; k += 64) { const float v = e ? x[k] + e[k] : x[k]; xs[k] = v; sq = fmaf(v, v, sq); }
    sq = group_sum<64>(sq);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int col = lane; col < 3 * P4; col += 64) {
      const int t = col / P4, p = col - t * P4;
      float4 acc = f4zero();
      if (p < P) {
        const float4* tr = reinterpret_cast<const float4*>((t == 0 ? Alog : t == 1 ? Ar : Cn) + (size_t)p * dp);
        const float4* xr = reinterpret_cast<const float4*>(xs);
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc = acc + xr[c] * tr[c];
      }
      res[col] = (acc.x + acc.y) + (acc.z + acc.w);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* L = res;
    const float* Rx = res + P4;
    const float* Nx = res + 2 * P4;
    // Gram products of L: fold[0] = G_RR L, [1] = G_RN L, [2] = G_RN^T L, [3] = G_NN L; quadratic forms by wave sums
    float q_rr = 0.f, q_rn = 0.f, q_nn = 0.f, d_nl = 0.f, d_rl = 0.f;
    if (lane < P4) {
      float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
#pragma unroll
      for (int q = 0; q < P4; ++q) {
        const float lq = L[q];
        g0 = fmaf(G[lane * P4 + q], lq, g0);
        g1 = fmaf(G[P4 * P4 + lane * P4 + q], lq, g1);
        g2 = fmaf(G[P4 * P4 + q * P4 + lane], lq, g2);
        g3 = fmaf(G[2 * P4 * P4 + lane * P4 + q], lq, g3);
      }
      fold[lane] = g0; fold[P4 + lane] = g1; fold[2 * P4 + lane] = g2; fold[3 * P4 + lane] = g3;
      const float l = L[lane];
      q_rr = l * g0; q_rn = l * g1; q_nn = l * g3; d_nl = Nx[lane] * l; d_rl = Rx[lane] * l;
    }
    q_rr = group_sum<64>(q_rr); q_rn = group_sum<64>(q_rn); q_nn = group_sum<64>(q_nn);
    d_nl = group_sum<64>(d_nl); d_rl = group_sum<64>(d_rl);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* o = out + row * (int64_t)sd.opitch;
    if (IS_USER) {
      // [AA: -2u ; -2L ; 2 (Ru + G_RR L) ; 0] [S: Nu ; -L ; 0] [AN: Nu + G_RN^T L + G_RN L ; -L ; 0] [NN: 2 G_NN L ; 0]
      const int oS = 16 * ka16, oAN = oS + 16 * ks16, oNN = oAN + 16 * ks16;
      for (int k = lane; k < orow; k += 64) {
        float v = 0.f;
        if (k < oS) {
          if (k < d) v = -2.f * xs[k];
          else if (k < d + P4) v = -2.f * L[k - d];
          else if (k < d + 2 * P4) v = 2.f * (Rx[k - d - P4] + fold[k - d - P4]);
        } else if (k < oAN) {
          const int e = k - oS;
          if (e < P4) v = Nx[e]; else if (e < 2 * P4) v = -L[e - P4];
        } else if (k < oNN) {
          const int e = k - oAN;
          if (e < P4) v = Nx[e] + fold[2 * P4 + e] + fold[P4 + e]; else if (e < 2 * P4) v = -L[e - P4];
        } else {
          const int e = k - oNN;
          if (e < P4) v = 2.f * fold[3 * P4 + e];
        }
        o[k] = v;
      }
      // u.NU, |AU|^2, AU.NU, |NU|^2
      if (lane == 0) *reinterpret_cast<float4*>(scal + row * (int64_t)sd.spitch) = make_float4(d_nl, sq + 2.f * d_rl + q_rr, d_nl + q_rn, q_nn);
      if (lane == 0 && sd.gthr) sd.gthr[row] = 0x7fffffff;
    } else {
      // [x ; Rx ; L ; Nx]
      for (int k = lane; k < orow; k += 64) o[k] = k < d ? xs[k] : k < d + P4 ? Rx[k - d] : k < d + 2 * P4 ? L[k - d - P4] : Nx[k - d - 2 * P4];
      // v.NV, |C0|^2, C0.NV, |NV|^2
      if (lane == 0) {
        *reinterpret_cast<float4*>(scal + row * (int64_t)sd.spitch) = make_float4(d_nl, sq - 2.f * d_rl + q_rr, d_nl - q_rn, q_nn);
        for (int k = orow + 4; k < sd.opitch; ++k) o[k] = 0.f;              // the pitch's spare float4 (padded K blocks may read it)
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// users and items in ONE launch: the first `users.blocks` workgroups take the user rows, the rest the item rows
template <int NCH, int NP>
__global__ __launch_bounds__(256) void pspace_rows_kernel(RowsSide users, RowsSide items, int P, const float* __restrict__ Alog,
                                                          const float* __restrict__ Ar, const float* __restrict__ Cn, int dp,
                                                          const float* __restrict__ grams, int ka16, int ks16) {
  if ((int)blockIdx.x < users.blocks) pspace_rows<true, NCH, NP>(users, blockIdx.x, P, Alog, Ar, Cn, dp, grams, ka16, ks16);
  else pspace_rows<false, NCH, NP>(items, blockIdx.x - users.blocks, P, Alog, Ar, Cn, dp, grams, ka16, ks16);
}

// The same rows on the matrix cores: a wave owns 16 rows.  L^T, Rx^T, Nx^T = (Alog | Ar | Cn) . x^T with the tables staged
// slot-ordered in LDS exactly as in pref_fwd_mc (lane (kq, j = row) ends up with preference 16 tt + 4 reg + kq of its row), the
// four Gram folds G_RR L, G_RN L, G_RN^T L, G_NN L as K = P products whose B operands are those registers (the Gram matrices
// staged in the matching slot / k order), the row scalars by lane-swap sums over the four kq lanes, and the operand rows written
// straight from the registers (16 B per (row, preference quad)).  One wave per row (above) is a chain of ~10 dependent round
// trips per row -- 31 us for the 9,746 rows of an ml1m pass; 16 rows per wave are ~210 MFMAs.
template <int NCH_, int NP_>
struct RGeom {
  static constexpr int NCH = NCH_, NP = NP_, D = 4 * NCH, P4 = 4 * NP;
  static constexpr int PT = (NP + 3) / 4, KG = (D + 15) / 16;
  static constexpr int PITCHA4 = 4 * KG + 1;
  static constexpr int SLOT_F4 = PT * 16 * PITCHA4;              // one table, float4
  static constexpr int GS_F = 4 * PT * NP * 64;                  // Gram A operands: [matrix][pt][k-step][lane]
  static constexpr int J = (16 * NCH + 63) / 64;
  static constexpr int TILE_F4 = 16 * NCH + 3;
  static constexpr size_t WAVE_BYTES = (size_t)TILE_F4 * 16 + 2 * 16 * 4;
  static constexpr size_t LDS = (size_t)3 * SLOT_F4 * 16 + (size_t)GS_F * 4 + 4 * WAVE_BYTES;
};

template <int NCH, int NP>
__global__ __launch_bounds__(256) void pspace_rows_mc_kernel(RowsSide users, RowsSide items, int P, const float* __restrict__ Alog,
                                                             const float* __restrict__ Ar, const float* __restrict__ Cn, int dp,
                                                             const float* __restrict__ gs, int ka16, int ks16, FiltBm fb) {
  using R = RGeom<NCH, NP>;
  constexpr int D = R::D, P4 = R::P4, PT = R::PT, KG = R::KG, PITCHA4 = R::PITCHA4, J = R::J;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x >= users.blocks + items.blocks) {          // the pass's filter bitmap rides here: the walk is one wave's chain of
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // round trips (~10 us), as long as this launch and on CUs it leaves idle
    filter_bitmap_wave(fb, ((int64_t)((int)blockIdx.x - users.blocks - items.blocks) * 4 + wv) * 16,
                       reinterpret_cast<uint32_t*>(smem) + (size_t)wv * 16 * fb.nsplit * fb.bm_words, threadIdx.x & 63);
    return;
  }
  v4* Slot = reinterpret_cast<v4*>(smem);                         // [3][PT * 16 slots][PITCHA4]
  float* GS = reinterpret_cast<float*>(Slot + 3 * R::SLOT_F4);
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* wbase = reinterpret_cast<char*>(GS + R::GS_F) + (size_t)w * R::WAVE_BYTES;
  v4* XT = reinterpret_cast<v4*>(wbase);                          // x  [16][NCH] (+ 3 chunks never used unmasked)
  int32_t* sid = reinterpret_cast<int32_t*>(XT + R::TILE_F4);     // [2][16]
  const bool is_user = (int)blockIdx.x < users.blocks;
  const RowsSide& sd = is_user ? users : items;
  const int block = is_user ? (int)blockIdx.x : (int)blockIdx.x - users.blocks;
  const int64_t row0 = ((int64_t)block * 4 + w) * 16;
  const bool active = row0 < sd.nrows;
  const int orow = sd.orow;
  // the wave's rows are requested first; the tables are staged while they are on their way
  if (active && lane < 16) {
    const int64_t gr = row0 + lane;
    const bool ok = gr < sd.nrows;
    sid[lane] = ok ? (int32_t)(sd.ids ? sd.ids[gr] : gr) : 0;
    sid[16 + lane] = (ok && sd.E) ? sd.map[gr] : 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  v4 xv[J], ev[J];
  if (active) {
    const v4* X4 = reinterpret_cast<const v4*>(sd.X);
    const v4* E4 = reinterpret_cast<const v4*>(sd.E);
    const uint32_t ldx4 = (uint32_t)(sd.ldx >> 2), lde4 = (uint32_t)(sd.lde >> 2);
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const int e = lane + 64 * jj;
      const int r = e < 16 * NCH ? e / NCH : 0, c = e < 16 * NCH ? e % NCH : 0;
      xv[jj] = X4[(uint64_t)(uint32_t)sid[r] * ldx4 + (uint32_t)c];
      ev[jj] = E4 ? E4[(uint64_t)(uint32_t)sid[16 + r] * lde4 + (uint32_t)c] : (v4){0.f, 0.f, 0.f, 0.f};
    }
  }
  {   // every load of the staging is issued before the first LDS write (a loop of load -> write round trips cost 6 us here)
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
    constexpr int SIT = (R::SLOT_F4 + 255) / 256, GIT = (R::GS_F / 4 + 255) / 256;
    v4 ta[SIT], tr[SIT], tc[SIT], tg[GIT];
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int idx = tid + 256 * it;
      const int srow = idx / PITCHA4, c = idx - srow * PITCHA4;
      const int tt = srow >> 4, i = srow & 15;
      const int p = 16 * tt + 4 * (i & 3) + (i >> 2);             // slot -> preference (block transposed, as in pref_fwd_mc)
      const bool ok = idx < R::SLOT_F4 && p < P && c < NCH;
      ta[it] = ok ? *reinterpret_cast<const v4*>(Alog + (size_t)p * dp + 4 * c) : zero;
      tr[it] = ok ? *reinterpret_cast<const v4*>(Ar + (size_t)p * dp + 4 * c) : zero;
      tc[it] = ok ? *reinterpret_cast<const v4*>(Cn + (size_t)p * dp + 4 * c) : zero;
    }
    const v4* g4 = reinterpret_cast<const v4*>(gs);               // already in operand order (pspace_gram_kernel)
#pragma unroll
    for (int it = 0; it < GIT; ++it) tg[it] = tid + 256 * it < R::GS_F / 4 ? g4[tid + 256 * it] : zero;
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int idx = tid + 256 * it;
      if (idx < R::SLOT_F4) { Slot[idx] = ta[it]; Slot[R::SLOT_F4 + idx] = tr[it]; Slot[2 * R::SLOT_F4 + idx] = tc[it]; }
    }
#pragma unroll
    for (int it = 0; it < GIT; ++it)
      if (tid + 256 * it < R::GS_F / 4) reinterpret_cast<v4*>(GS)[tid + 256 * it] = tg[it];
  }
  if (active) {
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const int e = lane + 64 * jj;
      if (e < 16 * NCH) {
        const int r = e / NCH, c = e % NCH;
        const v4 v = xv[jj] + ev[jj];
        XT[e] = v;
        if (row0 + r < sd.nrows)                                  // the x part of the operand row: [-2u ...] | [x ...]
          *reinterpret_cast<v4*>(sd.out + (row0 + r) * (int64_t)sd.opitch + 4 * c) = is_user ? -2.f * v : v;
      }
    }
  }
  __syncthreads();
  if (!active) return;
  // ---- L^T, Rx^T, Nx^T: acc[t][tt][reg] of lane (kq, row j) = table t's product for preference 16 tt + 4 reg + kq
  v4 acc[3][PT];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int tt = 0; tt < PT; ++tt) acc[t][tt] = (v4){0.f, 0.f, 0.f, 0.f};
  float sq = 0.f;
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    v4 bv = XT[j * NCH + 4 * g + kq];
    if (4 * g + 3 >= NCH) {
      if (4 * g + kq >= NCH) bv = (v4){0.f, 0.f, 0.f, 0.f};
    }
    sq += (bv[0] * bv[0] + bv[1] * bv[1]) + (bv[2] * bv[2] + bv[3] * bv[3]);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) {
        const v4 av = Slot[t * R::SLOT_F4 + (tt * 16 + j) * PITCHA4 + 4 * g + kq];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[c], acc[t][tt], 0, 0, 0);
      }
  }
  sq = allsum_kq(sq);
  // ---- Gram folds: fold[0] = G_RR L, [1] = G_RN L, [2] = G_RN^T L, [3] = G_NN L   (same lane layout as L)
  v4 fold[4][PT];
#pragma unroll
  for (int mtx = 0; mtx < 4; ++mtx)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      fold[mtx][pt] = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < NP; ++m)
        fold[mtx][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(GS[((mtx * PT + pt) * NP + m) * 64 + lane], acc[0][m >> 2][m & 3], fold[mtx][pt], 0, 0, 0);
    }
  float q_rr = 0.f, q_rn = 0.f, q_nn = 0.f, d_nl = 0.f, d_rl = 0.f;
#pragma unroll
  for (int tt = 0; tt < PT; ++tt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float l = acc[0][tt][reg];
      q_rr = fmaf(l, fold[0][tt][reg], q_rr); q_rn = fmaf(l, fold[1][tt][reg], q_rn); q_nn = fmaf(l, fold[3][tt][reg], q_nn);
      d_nl = fmaf(acc[2][tt][reg], l, d_nl); d_rl = fmaf(acc[1][tt][reg], l, d_rl);
    }
  q_rr = allsum_kq(q_rr); q_rn = allsum_kq(q_rn); q_nn = allsum_kq(q_nn); d_nl = allsum_kq(d_nl); d_rl = allsum_kq(d_rl);
  const int64_t row = row0 + j;
  if (row >= sd.nrows) return;
  float* o = sd.out + row * (int64_t)sd.opitch;
  if (is_user) {
    // [AA: -2u ; -2L ; 2 (Ru + G_RR L) ; 0] [S: Nu ; -L ; 0] [AN: Nu + G_RN^T L + G_RN L ; -L ; 0] [NN: 2 G_NN L ; 0]
    const int oS = 16 * ka16, oAN = oS + 16 * ks16, oNN = oAN + 16 * ks16;
#pragma unroll
    for (int tt = 0; tt < PT; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int p = 16 * tt + 4 * reg + kq;
        if (p < P4) {
          const float l = acc[0][tt][reg], rx = acc[1][tt][reg], nx = acc[2][tt][reg];
          o[D + p] = -2.f * l;
          o[D + P4 + p] = 2.f * (rx + fold[0][tt][reg]);
          o[oS + p] = nx;
          o[oS + P4 + p] = -l;
          o[oAN + p] = nx + fold[2][tt][reg] + fold[1][tt][reg];
          o[oAN + P4 + p] = -l;
          o[oNN + p] = 2.f * fold[3][tt][reg];
        }
      }
    for (int k = D + 2 * P4 + kq; k < oS; k += 4) o[k] = 0.f;
    for (int k = oS + 2 * P4 + kq; k < oAN; k += 4) o[k] = 0.f;
    for (int k = oAN + 2 * P4 + kq; k < oNN; k += 4) o[k] = 0.f;
    for (int k = oNN + P4 + kq; k < orow; k += 4) o[k] = 0.f;
    // u.NU, |AU|^2, AU.NU, |NU|^2
    if (kq == 0) *reinterpret_cast<float4*>(sd.scal + row * (int64_t)sd.spitch) = make_float4(d_nl, sq + 2.f * d_rl + q_rr, d_nl + q_rn, q_nn);
    if (kq == 0 && sd.gthr) sd.gthr[row] = 0x7fffffff;
  } else {
    // [x ; Rx ; L ; Nx]
#pragma unroll
    for (int tt = 0; tt < PT; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int p = 16 * tt + 4 * reg + kq;
        if (p < P4) {
          o[D + p] = acc[1][tt][reg];
          o[D + P4 + p] = acc[0][tt][reg];
          o[D + 2 * P4 + p] = acc[2][tt][reg];
        }
      }
    // v.NV, |C0|^2, C0.NV, |NV|^2
    if (kq == 0) {
      *reinterpret_cast<float4*>(sd.scal + row * (int64_t)sd.spitch) = make_float4(d_nl, sq - 2.f * d_rl + q_rr, d_nl - q_rn, q_nn);
      for (int k = orow + 4; k < sd.opitch; ++k) o[k] = 0.f;                // the pitch's spare float4 (padded K blocks may read it)
    }
  }
}

struct QScratch { float *grams, *gs, *A, *SCU, *B; uint64_t* part; uint32_t* bm; int32_t* gthr; };

// Words of the pass's filter bitmap per user for ANY split count <= 8 (nsplit x ceil(split_items / 32) with split_items = 16 x
// ceil(tiles / nsplit)), and the cap under which the pass builds one (beyond it every split's workgroup walks the CSR lists itself)
inline size_t filt_bm_words_per_user(int64_t n_items) { return (size_t)((n_items + IBT - 1) / IBT) / 2 + 13; }
inline size_t filt_bm_bytes(int64_t nq, int64_t n_items) {
  const size_t b = (size_t)((nq + 63) / 64) * 64 * filt_bm_words_per_user(n_items) * sizeof(uint32_t);
  return b <= ((size_t)128 << 20) ? b : 0;
}

template <typename G>
QScratch q_carve(void* scratch, int64_t nq, int64_t n_items) {
  QScratch s;
  float* p = reinterpret_cast<float*>(scratch);
  s.grams = p; p += 3 * 32 * 32;
  s.gs = p; p += 4 * 2 * 8 * 64;                                 // RGeom::GS_F at its largest (PT = 2, NP = 8)
  s.A = p; p += (size_t)nq * G::AROW;
  s.SCU = p; p += (size_t)nq * 4;
  s.B = p; p += (size_t)n_items * G::GROW + 64;                  // (eval_pass_pspace_bytes counts the same)
  s.part = reinterpret_cast<uint64_t*>(p);
  s.gthr = reinterpret_cast<int32_t*>(s.part + (size_t)nq * 8 * TOPN_MAX);  // (eval_pass_pspace_bytes: nq x 8 x topn keys, topn <= TOPN_MAX)
  s.bm = reinterpret_cast<uint32_t*>(s.gthr + ((nq + 63) / 64) * 64);
  return s;
}

template <typename G>
int launch_q(const float* U, int64_t ldu, const int64_t* u_ids, int64_t nq, const float* I, int64_t ldi, const float* E, int64_t lde,
             const int32_t* item2ent, int64_t n_items, const float* pref_ws,
             int ppad, int dp, int n_pref, const int64_t* filt_off, const int32_t* filt_ids, int topn, void* scratch, int32_t* top_ids,
             float* top_scores, hipStream_t st, const char* name) {
  const QScratch q = q_carve<G>(scratch, nq, n_items);
  const float* Alog = pref_ws;
  const float* Ar = pref_ws + (size_t)ppad * dp;
  const float* Cn = pref_ws + (size_t)(ppad + n_pref) * dp;
  const bool rows_mc = opt_eval_mc() && !((ldu | ldi | lde | dp) & 3) && aligned16(U) && aligned16(I) && (!E || aligned16(E)) && aligned16(pref_ws) &&
                       nq < (1ll << 31) && n_items < (1ll << 31);
  using R = RGeom<G::NCH, G::NP>;
  // the catalogue splits of the sweep (decided here: the filter bitmap is laid out by split)
  const int64_t ublocks = (nq + 63) / 64;
  int nsplit = (int)(256 * G::MINW / ublocks);                            // MINW workgroups per CU are resident: ONE round
  if (nsplit > 8) nsplit = 8;
  if (opt_eval_nsplit() > 0 && opt_eval_nsplit() <= 8) nsplit = opt_eval_nsplit();   // measurement knob
  const int64_t tiles = (n_items + IBT - 1) / IBT;
  if (nsplit > tiles) nsplit = (int)tiles;
  if (nsplit < 1) nsplit = 1;
  const int64_t split_items = ((tiles + nsplit - 1) / nsplit) * IBT;
  nsplit = (int)((n_items + split_items - 1) / split_items);
  const int bm_words = (int)((split_items + 31) / 32);
  const size_t lds = (size_t)2 * G::TILE_F4 * 16 + 4 * wave_lds_bytes(bm_words);
  if (lds > 160 * 1024) return 1;
  const int gram_blocks = (3 * G::P4 * G::P4 + 3) / 4;
  const size_t bm_lds = (size_t)4 * 16 * nsplit * bm_words * sizeof(uint32_t);
  const bool use_bm = filt_off && filt_ids && filt_bm_bytes(nq, n_items) && (size_t)nsplit * bm_words <= filt_bm_words_per_user(n_items) &&
                      bm_lds <= 64 * 1024;
  FiltBm fb{filt_off, filt_ids, nq, n_items, q.bm, nsplit, bm_words, (uint32_t)split_items, gram_blocks};
  // the bitmap's waves ride in the operand-row launch (16 us, 146 workgroups at ml1m size); without the matrix-core row kernel, in the Gram launch
  const bool bm_in_gram = use_bm && !rows_mc;
  hipLaunchKernelGGL(pspace_gram_kernel, dim3(gram_blocks + (bm_in_gram ? (unsigned)ublocks : 0u)), dim3(256), bm_in_gram ? bm_lds : 0, st, Ar, Cn, dp,
                     G::D, n_pref, G::P4, q.grams, rows_mc ? q.gs : nullptr, R::PT, G::NP, fb);
  const size_t lds_rows = (size_t)4 * (G::D + 7 * G::P4) * sizeof(float);
  // many small workgroups: a row is a chain of dependent round trips (id -> row -> products -> store), hidden only by occupancy
  RowsSide us{U, ldu, u_ids, nullptr, 0, nullptr, nq, q.A, G::AROW, q.SCU, grid_for((nq + 3) / 4, 2048), G::AROW, 4, q.gthr};
  RowsSide is{I, ldi, nullptr, E, lde, item2ent, n_items, q.B, G::ROWB, q.B + G::ROWB, grid_for((n_items + 3) / 4, 2048), G::GROW, G::GROW, nullptr};
  if (rows_mc) {    // 16 rows per wave on the matrix cores
    us.blocks = (int)((nq + 63) / 64); is.blocks = (int)((n_items + 63) / 64);
    const size_t rows_lds = use_bm && bm_lds > R::LDS ? bm_lds : R::LDS;
    (void)hipFuncSetAttribute((const void*)pspace_rows_mc_kernel<G::NCH, G::NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_lds);
    hipLaunchKernelGGL((pspace_rows_mc_kernel<G::NCH, G::NP>), dim3(us.blocks + is.blocks + (use_bm ? (unsigned)ublocks : 0u)), dim3(256), rows_lds, st,
                       us, is, n_pref, Alog, Ar, Cn, dp, q.gs, G::KA, G::KS, fb);
  } else {
    hipLaunchKernelGGL((pspace_rows_kernel<G::NCH, G::NP>), dim3(us.blocks + is.blocks), dim3(256), lds_rows, st, us, is, n_pref, Alog, Ar, Cn, dp,
                       q.grams, G::KA, G::KS);
  }
  if (int e = check_launch(name)) return e;
  QArgs a{};
  a.A = q.A; a.SCU = q.SCU; a.B = q.B; a.nq = nq; a.n_items = n_items;
  a.filt_off = filt_off; a.filt_ids = filt_ids; a.topn = topn; a.part = q.part; a.dbg = opt_dbg_eval();
  a.split_items = split_items; a.nsplit = nsplit; a.bm_words = bm_words; a.bmg = use_bm ? q.bm : nullptr;
  a.gthr = nsplit > 1 && !(a.dbg & 64) ? q.gthr : nullptr;
  (void)hipFuncSetAttribute((const void*)eval_pass_q_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((eval_pass_q_kernel<G>), dim3((unsigned)ublocks, (unsigned)nsplit), dim3(256), lds, st, a);
  if (int e = check_launch(name)) return e;
  hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(MERGE_T), 0, st, q.part, nq, nsplit, topn, (const float*)nullptr, top_ids,
                     top_scores);
  return check_launch(name);
}

template <int NCH>
int launch_q_p(int n_pref, const float* U, int64_t ldu, const int64_t* u_ids, int64_t nq, const float* I, int64_t ldi, const float* E,
               int64_t lde, const int32_t* item2ent, int64_t n_items,
               const float* pref_ws, int ppad, int dp, const int64_t* filt_off, const int32_t* filt_ids, int topn, void* scratch,
               int32_t* top_ids, float* top_scores, hipStream_t st, const char* name) {
  if (n_pref <= 4)
    return launch_q<QGeom<NCH, 1>>(U, ldu, u_ids, nq, I, ldi, E, lde, item2ent, n_items, pref_ws, ppad, dp, n_pref, filt_off, filt_ids, topn, scratch, top_ids, top_scores, st, name);
  if (n_pref <= 20)
    return launch_q<QGeom<NCH, 5>>(U, ldu, u_ids, nq, I, ldi, E, lde, item2ent, n_items, pref_ws, ppad, dp, n_pref, filt_off, filt_ids, topn, scratch, top_ids, top_scores, st, name);
  return launch_q<QGeom<NCH, 8>>(U, ldu, u_ids, nq, I, ldi, E, lde, item2ent, n_items, pref_ws, ppad, dp, n_pref, filt_off, filt_ids, topn, scratch, top_ids, top_scores, st, name);
}

}  // namespace

}  // namespace ktup

namespace ktup {

// the merge of the splits' partial lists, for the other sweeps (ktup_eval.hip: the hard gate's pass)
int launch_topk_merge(const uint64_t* part, int64_t nq, int nsplit, int topn, int32_t* top_ids, float* top_scores, hipStream_t st, const char* name) {
  hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(MERGE_T), 0, st, part, nq, nsplit, topn, (const float*)nullptr, top_ids,
                     top_scores);
  return check_launch(name);
}

// The preference-space pass (see QGeom): scratch for the Gram matrices, both operand tables, the scalars and the partial lists.
size_t eval_pass_pspace_bytes(int d, int n_pref, int64_t nq, int64_t n_items, int topn) {
  const size_t p4 = n_pref <= 4 ? 4 : n_pref <= 20 ? 20 : 32;
  const size_t ka = (d + 2 * p4 + 15) / 16, ks = (2 * p4 + 15) / 16, kn = (p4 + 15) / 16;
  const size_t arow = 16 * (ka + 2 * ks + kn), rowb = d + 3 * p4, grow = 4 * ((rowb / 4 + 1) | 1);      // QGeom::AROW, ROWB, GROW
  (void)topn;
  return (3 * 32 * 32 + 4 * 2 * 8 * 64 + (size_t)nq * (arow + 4) + (size_t)n_items * grow + 64) * sizeof(float) +
         (size_t)nq * 8 * TOPN_MAX * sizeof(uint64_t) + (size_t)((nq + 63) / 64) * 64 * sizeof(int32_t) + filt_bm_bytes(nq, n_items);
}

// Items: I[row] (+ E[item2ent[row]] for KTUP; E == NULL for TUP); pref_ws: the prepared tables (ktup_pref_prepare; ppad / dp its
// geometry).  Returns KTUP_OK / an error, or 1 for shapes the pass does not cover.
int eval_pass_pspace(const float* U, int64_t ldu, const int64_t* u_ids, int64_t nq, const float* I, int64_t ldi, const float* E, int64_t lde,
                     const int32_t* item2ent, int64_t n_items, const float* pref_ws,
                     int ppad, int dp, int n_pref, int d, const int64_t* filt_off, const int32_t* filt_ids, int topn, void* scratch,
                     int32_t* top_ids, float* top_scores, hipStream_t st, const char* name) {
  if ((d != 64 && d != 100 && d != 128) || n_pref < 1 || n_pref > 32 || topn < 1 || topn > TOPN_MAX || n_items >= (1ll << 31)) return 1;
  if (d == 64) return launch_q_p<16>(n_pref, U, ldu, u_ids, nq, I, ldi, E, lde, item2ent, n_items, pref_ws, ppad, dp, filt_off, filt_ids, topn, scratch, top_ids, top_scores, st, name);
  if (d == 100) return launch_q_p<25>(n_pref, U, ldu, u_ids, nq, I, ldi, E, lde, item2ent, n_items, pref_ws, ppad, dp, filt_off, filt_ids, topn, scratch, top_ids, top_scores, st, name);
  return launch_q_p<32>(n_pref, U, ldu, u_ids, nq, I, ldi, E, lde, item2ent, n_items, pref_ws, ppad, dp, filt_off, filt_ids, topn, scratch, top_ids, top_scores, st, name);
}

}  // namespace ktup
