// K5 / K6 / K7 backward on the matrix cores, soft and straight-through Gumbel gate (transUP.py:69-115 / jTransUP.py:122-143,250-262 differentiated).
//
// Forward (per pair):  x = u + i (+ e),  q = u - i (- e),  L = x . Alog^T,  r = L . Ar,  n = L . Cn,  s = q . n,
//                      z = q + r - s n,  score = sum_k dist(z_k)      (Alog = A/2, Ar = beta A, Cn = beta C, ktup_pref_prepare)
// Backward for upstream g:   gz = g dist'(z),  av = gz . n,  gq = gz - av n,  gr = gz,  gn = -av q - s gz,
//                            gL_p = Ar_p . gr + Cn_p . gn,   gx = sum_p gL_p Alog_p,   gu = gq + gx,   gi = ge = gx - gq,
//                            gA_p += 1/2 gL_p x + beta L_p gr,   gC_p += beta L_p gn        (mixed tables A, C)
// The three tables are staged once, slot-ordered with an odd float4 pitch; both operand patterns read the same rows.
// A wave owns 16 pairs through all phases; every contraction runs on v_mfma_f32_16x16x4_f32 with the operand layouts of
// pref_fwd_mc (lane = (kq, pair); D registers of one phase are the B operands of the next):
//   A  forward recompute  L^T = Alog . x^T   then  n^T, r^T = Cn^T L^T, Ar^T L^T                (as pref_fwd_mc)
//   B  gL^T = ArSlot . gr^T + CnSlot . gn^T        B operands = the gz / gn registers of phase A's coordinate layout
//   C  gx^T = Alog2^T . gL^T                       B operands = the gL registers
//   D  gA += (gL/2)^T-by-pairs . X + (beta L) . GR,  gC += (beta L) . GN     K = the tile's 16 pairs; operands via small LDS
//      transposes; accumulators stay in registers across the wave's tiles and reach memory with one atomic per element.
// The first backward kernel (pref_bwd_kernel, lane = pair, 64-pair workgroup tiles, VALU) needs 118 us for the 1024 pairs
// of a B=512 step because only 16 workgroups exist and each walks its tile serially; here a tile is ~450 MFMAs.
// Occupancy is one wave per SIMD (four per CU: the per-wave LDS tiles), so the schedule hides its own waits: rows of the next
// tile gathered a tile ahead, LDS operands of step i + 1 read before the MFMAs of step i, phase C's stores issued under the next
// coordinate tile's MFMAs (tools/wave_model.py compares such schedules on the ISA).  With 17-20 preferences
// the four live rows of the second preference tile take v_mfma_f32_4x4x1 in phase D (BGeom::THIN).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ktup_common.h"
#include "ktup_lane_swap.h"
#include "ktup_pref_geom.h"

namespace ktup {
namespace {

template <int NCH_, int NP_, bool HASE_, bool HARD_>
struct BGeom {
  static constexpr int NCH = NCH_, NP = NP_, D = 4 * NCH;
  static constexpr bool HASE = HASE_, HARD = HARD_;   // HARD: straight-through Gumbel gate
  static constexpr int KG = (D + 15) / 16, CT = KG;
  static constexpr int PT = (NP + 3) / 4;                  // 16-slot preference tiles
  static constexpr int J = (16 * NCH + 63) / 64;
  static constexpr int TOTAL = 16 * NCH;
  static constexpr int PITCHA4 = 4 * KG + 1;               // slot-ordered tables (A operand of K = coordinate GEMMs)
  static constexpr int SLOT_F4 = PT * 16 * PITCHA4;
  static constexpr int TROW = 16 * PT;
  static constexpr int RP = PITCHA4 * 4;                   // the same rows serve the K = preference GEMMs: float pitch of a slot row
  static constexpr size_t TABLE_BYTES = (size_t)3 * SLOT_F4 * 16;
  static constexpr int TILE_F4 = 16 * NCH + 3;             // one (16 pairs x d) tile + 3 zero chunks
  static constexpr int LROW = 4 * NP;                      // rows of the transposed arrays that are kept (P <= 4 NP)
  // P in 17..20 (NP = 5; the paper's 20 preferences): the second 16-preference tile holds FOUR live rows.  Phase D (168 of a
  // tile's 441 matrix instructions) handles them with v_mfma_f32_4x4x1 (16 blocks of 4 x 4 outer products, 8 clocks) instead of a
  // 16 x 16 x 4 tile that is three quarters padding: 96 x 8 clocks instead of 84 x 32, and 16 accumulator registers instead of 56.
  static constexpr bool THIN = NP == 5;
  static constexpr int PTD = THIN ? PT - 1 : PT;           // full preference tiles of phase D
  static constexpr int NG = (D + 63) / 64;                 // THIN: 64-coordinate groups (one per 4x4x1 instruction)
  static constexpr int LT_F = LROW * 17;                   // transposed [preference][pair] arrays, pitch 17
  static constexpr int NOISE_F = HARD ? 16 * TROW : 0;     // HARD: Gumbel noise of the tile, [pair][preference]
  static constexpr size_t WAVE_BYTES = ((size_t)4 * TILE_F4 * 16 + (size_t)2 * LT_F * 4 + 2 * 3 * 16 * 4 + (size_t)NOISE_F * 4 + 15) & ~(size_t)15;
  static constexpr int NW = TABLE_BYTES + 4 * WAVE_BYTES <= 160 * 1024   ? 4      // one wave per SIMD: the accumulators alone are 112 registers
                            : TABLE_BYTES + 3 * WAVE_BYTES <= 160 * 1024 ? 3
                            : TABLE_BYTES + 2 * WAVE_BYTES <= 160 * 1024 ? 2
                                                                         : 1;
  static constexpr size_t LDS = TABLE_BYTES + NW * WAVE_BYTES;
};

struct BArgs {
  const v4 *U, *I, *E;
  uint32_t ldu4, ldi4, lde4;
  const int32_t* item2ent;
  const float *Alog, *Ar, *Cn;   // prepared tables, row pitch dp floats
  int dp, P, l1;
  float beta;
  const int64_t *u_ids, *i_ids;
  int64_t n, ent_pad;
  const float* gscore;
  float *gU, *gI, *gE, *gA, *gC;
  float *GU, *GV;                // ROWOUT kernels: per-pair row gradients (n x D, pitch D) instead of atomics into gU / gI / gE
  int gumbel;                    // KTUP_GUMBEL_* (HARD kernels)
  const float* uniform;
  uint64_t seed, offset;
};

// ROWOUT: the row gradients gu = gq + gx and gv = gx - gq leave as plain stores into GU / GV (one row per pair); the launcher
// then sums them per table row by sorted segments (ktup_segreduce.hip) -- for large batches / hot rows, where d float atomics
// per gathered row serialise on shared L2 lines.
template <typename G, bool ROWOUT>
__global__ __launch_bounds__(G::NW * 64) void pref_bwd_mc_kernel(BArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  constexpr int NCH = G::NCH, NP = G::NP, D = G::D, KG = G::KG, CT = G::CT, PT = G::PT, J = G::J, TOTAL = G::TOTAL;
  constexpr int PITCHA4 = G::PITCHA4, RP = G::RP, NW = G::NW;
  constexpr bool HASE = G::HASE, HARD = G::HARD;
  constexpr int TROW = G::TROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* AlogSlot = reinterpret_cast<v4*>(smem);                 // [PT*16 slots][PITCHA4]
  v4* ArSlot = AlogSlot + G::SLOT_F4;
  v4* CnSlot = ArSlot + G::SLOT_F4;
  // the K = preference GEMMs read the same rows by b32: preference 16 t + 4 reg + kq lives in slot row 16 t + 4 kq + reg
  const float* Alog2 = reinterpret_cast<const float*>(AlogSlot);
  const float* Ar2 = reinterpret_cast<const float*>(ArSlot);
  const float* Cn2 = reinterpret_cast<const float*>(CnSlot);
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* wbase = reinterpret_cast<char*>(CnSlot + G::SLOT_F4) + (size_t)w * G::WAVE_BYTES;
  v4* XT = reinterpret_cast<v4*>(wbase);                      // x   [16][NCH]
  v4* QT = XT + G::TILE_F4;                                   // q
  v4* GRT = QT + G::TILE_F4;                                  // gr = gz
  v4* GNT = GRT + G::TILE_F4;                                 // gn
  float* LT = reinterpret_cast<float*>(GNT + G::TILE_F4);     // [TROW][17]  beta * L   (transposed: preference major)
  float* GLT = LT + G::LT_F;                                  // [TROW][17]  gL / 2
  int32_t* sid2 = reinterpret_cast<int32_t*>(GLT + G::LT_F);  // [2][3][16]: ids of this tile and of the next one
  float* noise = reinterpret_cast<float*>(sid2 + 96);        // HARD: [16][TROW]
  // ---- stage the three tables in both layouts
  {
    const int P = a.P, dp = a.dp;                              // float4 granularity: rows are 16-byte aligned, pitch % 4 == 0
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < G::SLOT_F4; idx += NW * 64) {
      const int srow = idx / PITCHA4, c = idx - srow * PITCHA4;
      const int tt = srow >> 4, i = srow & 15;
      const int p = 16 * tt + 4 * (i & 3) + (i >> 2);          // slot -> preference (block transposed, as in pref_fwd_mc)
      const bool ok = p < P && c < NCH;
      AlogSlot[idx] = ok ? *reinterpret_cast<const v4*>(a.Alog + p * dp + 4 * c) : zero;
      ArSlot[idx] = ok ? *reinterpret_cast<const v4*>(a.Ar + p * dp + 4 * c) : zero;
      CnSlot[idx] = ok ? *reinterpret_cast<const v4*>(a.Cn + p * dp + 4 * c) : zero;
    }
    if (lane < 3) {
      XT[16 * NCH + lane] = (v4){0.f, 0.f, 0.f, 0.f}; QT[16 * NCH + lane] = XT[16 * NCH + lane];
      GRT[16 * NCH + lane] = XT[16 * NCH + lane]; GNT[16 * NCH + lane] = XT[16 * NCH + lane];
    }
  }
  __syncthreads();
  int grow[J], gc[J];
#pragma unroll
  for (int jj = 0; jj < J; ++jj) {
    const int e = lane + 64 * jj;
    const bool past = e >= TOTAL;
    grow[jj] = past ? 0 : e / NCH;
    gc[jj] = past ? 0 : e % NCH;
  }
  const bool last_ok = lane + 64 * (J - 1) < TOTAL;
  const bool l1 = a.l1 != 0;
  const float beta = a.beta;
  // table-gradient accumulators: phase D's D layout, lane (kq, n) <-> preference 16 pt + 4 kq + reg, coordinate 16 ct + n
  constexpr int PTD = G::PTD, NG = G::NG;
  constexpr bool THIN = G::THIN;
  v4 accA[PTD][CT], accC[PTD][CT];
#pragma unroll
  for (int pt = 0; pt < PTD; ++pt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { accA[pt][ct] = (v4){0.f, 0.f, 0.f, 0.f}; accC[pt][ct] = accA[pt][ct]; }
  // THIN: register r of lane l <-> preference 16 (PT - 1) + r, coordinate 64 g + l   (4x4x1: block = l / 4, D[i = r][j = l % 4])
  v4 thinA[NG], thinC[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) { thinA[g] = (v4){0.f, 0.f, 0.f, 0.f}; thinC[g] = thinA[g]; }
  const int64_t ntiles = (a.n + 15) / 16;
  const int64_t tstride = (int64_t)gridDim.x * NW;
  // The gathers run one tile ahead of the arithmetic (a wave is alone on its SIMD: nothing else hides three dependent
  // trips to memory per tile).  Ids are fetched three tiles ahead, the entity id two tiles ahead, the rows one tile ahead.
  auto load_ui = [&](int64_t tile, int32_t& u, int32_t& i) {
    const int64_t gr = tile * 16 + j;
    const bool ok = lane < 16 && gr < a.n;                    // tiles past the end: id 0 (row 0 is fetched and never used)
    u = ok ? (int32_t)a.u_ids[gr] : 0;
    i = ok ? (int32_t)a.i_ids[gr] : 0;
  };
  auto load_e = [&](int32_t i) -> int32_t { return (HASE && lane < 16) ? a.item2ent[i] : 0; };
  v4 uu[J], vv[J], ee[J];
  auto gather = [&](const int32_t* sid) {
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      asm volatile("" : "+v"(gc[jj]));
      const uint32_t idu = (uint32_t)sid[grow[jj]], idi = (uint32_t)sid[16 + grow[jj]];
      uu[jj] = a.U[(uint64_t)idu * a.ldu4 + (uint32_t)gc[jj]];
      vv[jj] = a.I[(uint64_t)idi * a.ldi4 + (uint32_t)gc[jj]];
      if (HASE) {
        const uint32_t ide = (uint32_t)sid[32 + grow[jj]];
        ee[jj] = a.E[(uint64_t)ide * a.lde4 + (uint32_t)gc[jj]];
      }
    }
  };
  auto put_ids = [&](int32_t* sid, int32_t u, int32_t i, int32_t e) {
    if (lane < 16) { sid[lane] = u; sid[16 + lane] = i; sid[32 + lane] = e; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  const int64_t tile0 = (int64_t)blockIdx.x * NW + w;
  int32_t nu1 = 0, ni1 = 0, ne1 = 0, nu2 = 0, ni2 = 0, ne2 = 0, nu3 = 0, ni3 = 0;
  int cb = 0;
  if (tile0 < ntiles) {
    int32_t nu0, ni0;
    load_ui(tile0, nu0, ni0);
    load_ui(tile0 + tstride, nu1, ni1);
    load_ui(tile0 + 2 * tstride, nu2, ni2);
    const int32_t ne0 = load_e(ni0);
    ne1 = load_e(ni1);
    put_ids(sid2, nu0, ni0, ne0);
    gather(sid2);
  }
  for (int64_t tile_id = tile0; tile_id < ntiles; tile_id += tstride) {
    const int64_t row0 = tile_id * 16;
    const int32_t* sid = sid2 + 48 * cb;
    // ---- x and q tiles from the rows fetched during the previous tile
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const v4 ve = HASE ? vv[jj] + ee[jj] : vv[jj];
      if (jj < J - 1 || last_ok) { XT[lane + 64 * jj] = uu[jj] + ve; QT[lane + 64 * jj] = uu[jj] + (-ve); }
    }
    // ---- next tile: ids to LDS, rows in flight; the id fetches of the tiles after it
    put_ids(sid2 + 48 * (cb ^ 1), nu1, ni1, ne1);                 // (its fence also publishes XT / QT)
    ne2 = load_e(ni2);
    load_ui(tile_id + 3 * tstride, nu3, ni3);
    gather(sid2 + 48 * (cb ^ 1));
    // ---- A1: L^T.  lg[tt][reg] of lane (kq, pair j) = logit of preference 16 tt + 4 reg + kq
    v4 lg[PT];
#pragma unroll
    for (int tt = 0; tt < PT; ++tt) lg[tt] = (v4){0.f, 0.f, 0.f, 0.f};
    // A wave is alone on its SIMD, so an LDS round trip in front of every group of MFMAs is dead time: in every phase the operands of
    // step i + 1 are read before the MFMAs of step i are issued (KTUP_PIN keeps the compiler from sinking the reads back down).
#define KTUP_PIN() __builtin_amdgcn_sched_barrier(0)
    {
      auto ld_b = [&](int g) -> v4 {
        v4 bv = XT[j * NCH + 4 * g + kq];
        if (4 * g + 3 >= NCH) {
          if (4 * g + kq >= NCH) bv = (v4){0.f, 0.f, 0.f, 0.f};
        }
        return bv;
      };
      v4 bn = ld_b(0), an[PT];
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) an[tt] = AlogSlot[(tt * 16 + j) * PITCHA4 + kq];
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const v4 bv = bn;
        v4 av[PT];
#pragma unroll
        for (int tt = 0; tt < PT; ++tt) av[tt] = an[tt];
        if (g + 1 < KG) {
          bn = ld_b(g + 1);
#pragma unroll
          for (int tt = 0; tt < PT; ++tt) an[tt] = AlogSlot[(tt * 16 + j) * PITCHA4 + 4 * (g + 1) + kq];
        }
        KTUP_PIN();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int tt = 0; tt < PT; ++tt) lg[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt][c], bv[c], lg[tt], 0, 0, 0);
      }
    }
    // ---- ST-Gumbel gate (transUP.py:118-170): forward weights w = one_hot(argmax(l + g)), backward through y = softmax(l + g).
    //      From here on `lg` holds the FORWARD weights (the raw logits for the soft gate); ysoft keeps y for the Jacobian.
    v4 ysoft[PT];
    if constexpr (HARD) {
      const int64_t grow = min(row0 + j, a.n - 1);
      const uint64_t base = (uint64_t)grow * (uint64_t)a.P;
      if (a.gumbel == KTUP_GUMBEL_PHILOX) {     // same stream and block sharing as pref_fwd_mc
        const uint64_t i0 = base + a.offset, fb = i0 >> 2, lb = (i0 + (uint64_t)a.P - 1) >> 2;
        const Philox ph(a.seed);
        for (uint64_t b = fb + kq; b <= lb; b += 4) {
          const uint4 r = ph(b, 0x4b545550ull /* "KTUP" stream tag */);
          const uint32_t wds[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
          for (int wd = 0; wd < 4; ++wd) {
            const int64_t pp = (int64_t)((b << 2) + wd) - (int64_t)i0;
            if (pp >= 0 && pp < a.P) noise[j * TROW + (int)pp] = gumbel_from_uniform(u01(wds[wd]));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      float best = -INFINITY;
      int bp = 0x7fffffff;
#pragma unroll
      for (int tt = 0; tt < PT; ++tt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {               // ascending p within the lane: strict > keeps the first maximum
          const int pp = 16 * tt + 4 * reg + kq;
          float v = -INFINITY;
          if (pp < a.P) {
            const float g = a.gumbel == KTUP_GUMBEL_PHILOX ? noise[j * TROW + pp] : gumbel_from_uniform(a.uniform[base + pp]);
            v = lg[tt][reg] + g;
            if (v > best || bp == 0x7fffffff) { best = v; bp = pp; }
          }
          lg[tt][reg] = v;                                // noisy logit (-inf for padding preferences)
        }
      {
        const u2 rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
        const u2 rp = __builtin_amdgcn_permlane32_swap((unsigned)bp, (unsigned)bp, false, false);
        const float v0 = __uint_as_float(rv[0]), v1 = __uint_as_float(rv[1]);
        const int p0 = (int)rp[0], p1 = (int)rp[1];
        const bool take1 = p0 == 0x7fffffff || (p1 != 0x7fffffff && (v1 > v0 || (v1 == v0 && p1 < p0)));
        best = take1 ? v1 : v0; bp = take1 ? p1 : p0;
      }
      {
        const u2 rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(best), __float_as_uint(best), false, false);
        const u2 rp = __builtin_amdgcn_permlane16_swap((unsigned)bp, (unsigned)bp, false, false);
        const float v0 = __uint_as_float(rv[0]), v1 = __uint_as_float(rv[1]);
        const int p0 = (int)rp[0], p1 = (int)rp[1];
        const bool take1 = p0 == 0x7fffffff || (p1 != 0x7fffffff && (v1 > v0 || (v1 == v0 && p1 < p0)));
        best = take1 ? v1 : v0; bp = take1 ? p1 : p0;
      }
      float den = 0.f;
#pragma unroll
      for (int tt = 0; tt < PT; ++tt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float e = expf(lg[tt][reg] - best);        // exp(-inf) = 0 for padding preferences
          ysoft[tt][reg] = e;
          den += e;
          lg[tt][reg] = (16 * tt + 4 * reg + kq == bp) ? 1.f : 0.f;
        }
      const float inv = 1.f / allsum_kq(den);
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) ysoft[tt] = ysoft[tt] * inv;
    }
    // ---- A2: n^T, r^T per coordinate tile; lane (kq, j) owns coordinates 16 ct + 4 kq + reg of pair j
    v4 nn[CT], zz[CT], qv[CT];
    v4 sacc = (v4){0.f, 0.f, 0.f, 0.f};
    {
      float cnn[NP], arn[NP];
      auto ld_t = [&](int ct) {
#pragma unroll
        for (int m = 0; m < NP; ++m) {
          const int prow = (16 * (m >> 2) + 4 * kq + (m & 3)) * RP + 16 * ct + j;
          cnn[m] = Cn2[prow];
          arn[m] = Ar2[prow];
        }
      };
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) qv[ct] = (4 * ct + kq < NCH) ? QT[j * NCH + 4 * ct + kq] : (v4){0.f, 0.f, 0.f, 0.f};
      ld_t(0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float cn[NP], ar[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) { cn[m] = cnn[m]; ar[m] = arn[m]; }
        if (ct + 1 < CT) ld_t(ct + 1);
        KTUP_PIN();
        nn[ct] = (v4){0.f, 0.f, 0.f, 0.f};
        zz[ct] = qv[ct];                                          // q + r accumulates on top of q
#pragma unroll
        for (int m = 0; m < NP; ++m) {
          nn[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cn[m], lg[m >> 2][m & 3], nn[ct], 0, 0, 0);
          zz[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[m], lg[m >> 2][m & 3], zz[ct], 0, 0, 0);
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) sacc += qv[ct] * nn[ct];
    }
    const float s = allsum_kq((sacc[0] + sacc[1]) + (sacc[2] + sacc[3]));
    const float g = row0 + j < a.n ? a.gscore[row0 + j] : 0.f;    // tail pairs contribute nothing
    const float g1 = l1 ? g : 0.f, g2 = l1 ? 0.f : 2.f * g;       // both norms without a branch per element (2 g z is exact either way)
    // gz (kept in zz), av
    v4 aacc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const v4 z = zz[ct] - s * nn[ct];
      v4 gz;
#pragma unroll
      for (int c = 0; c < 4; ++c) gz[c] = fmaf(g2, z[c], g1 * (z[c] > 0.f ? 1.f : (z[c] < 0.f ? -1.f : 0.f)));   // g * ddist1(z, l1)
      zz[ct] = gz;
      aacc += gz * nn[ct];
    }
    const float av = allsum_kq((aacc[0] + aacc[1]) + (aacc[2] + aacc[3]));
    // gq (into qv), gn (into nn); gr = gz.  Tiles GRT / GNT for phase D.
    v4 gq[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      gq[ct] = zz[ct] - av * nn[ct];
      nn[ct] = -av * qv[ct] - s * zz[ct];
      if (4 * ct + kq < NCH) { GRT[j * NCH + 4 * ct + kq] = zz[ct]; GNT[j * NCH + 4 * ct + kq] = nn[ct]; }
    }
    // ---- B: gL^T = ArSlot . gr^T + CnSlot . gn^T   (K = coordinates; B operands are zz / nn in registers)
    v4 gl[PT];
#pragma unroll
    for (int tt = 0; tt < PT; ++tt) gl[tt] = (v4){0.f, 0.f, 0.f, 0.f};
    {
      v4 arn[PT], cnn[PT];
      auto ld_s = [&](int g4) {
#pragma unroll
        for (int tt = 0; tt < PT; ++tt) {
          arn[tt] = ArSlot[(tt * 16 + j) * PITCHA4 + 4 * g4 + kq];
          cnn[tt] = CnSlot[(tt * 16 + j) * PITCHA4 + 4 * g4 + kq];
        }
      };
      ld_s(0);
#pragma unroll
      for (int g4 = 0; g4 < KG; ++g4) {
        v4 ar[PT], cn[PT];
#pragma unroll
        for (int tt = 0; tt < PT; ++tt) { ar[tt] = arn[tt]; cn[tt] = cnn[tt]; }
        if (g4 + 1 < KG) ld_s(g4 + 1);
        KTUP_PIN();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int tt = 0; tt < PT; ++tt) {
            gl[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tt][c], zz[g4][c], gl[tt], 0, 0, 0);
            gl[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cn[tt][c], nn[g4][c], gl[tt], 0, 0, 0);
          }
      }
    }
    if constexpr (HARD) {        // gl <- y * (gw - y . gw): the softmax Jacobian of y = softmax(l + g) applied to gw
      float dot = 0.f;
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) dot += (ysoft[tt][0] * gl[tt][0] + ysoft[tt][1] * gl[tt][1]) + (ysoft[tt][2] * gl[tt][2] + ysoft[tt][3] * gl[tt][3]);
      dot = allsum_kq(dot);
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) gl[tt] = ysoft[tt] * (gl[tt] - dot);
    }
    // transposed copies for phase D: LT[p][pair] = beta w (w = L for the soft gate), GLT[p][pair] = gl / 2
#pragma unroll
    for (int tt = 0; tt < PT; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int p = 16 * tt + 4 * reg + kq;
        if (16 * tt + 4 * reg + 3 < G::LROW || p < G::LROW) {
          LT[p * 17 + j] = beta * lg[tt][reg];
          GLT[p * 17 + j] = 0.5f * gl[tt][reg];
        }
      }
    // ---- C: gx^T = Alog2^T . gL^T, then the row gradients
    {
      const int64_t gr = row0 + j;
      const bool live = gr < a.n;
      const int32_t ur = sid[j], ir = sid[16 + j], er = sid[32 + j];
      float* pu = a.gU + (int64_t)ur * a.ldu4 * 4;
      float* pi = a.gI + (int64_t)ir * a.ldi4 * 4;
      float* pe = (HASE && er != a.ent_pad) ? a.gE + (int64_t)er * a.lde4 * 4 : nullptr;
      auto emit = [&](int ct, const v4& gx) {                    // the row gradients of coordinate tile ct
        const int c0 = 16 * ct + 4 * kq;
        if (live && 4 * ct + kq < NCH) {
          const v4 gu = gq[ct] + gx, gv = gx - gq[ct];
          if constexpr (ROWOUT) {
            *reinterpret_cast<v4*>(a.GU + gr * D + c0) = gu;
            *reinterpret_cast<v4*>(a.GV + gr * D + c0) = gv;
          } else {
            atomic_add4(pu + c0, make_float4(gu[0], gu[1], gu[2], gu[3]));
            atomic_add4(pi + c0, make_float4(gv[0], gv[1], gv[2], gv[3]));
            if (pe) atomic_add4(pe + c0, make_float4(gv[0], gv[1], gv[2], gv[3]));
          }
        }
      };
      float aln[NP];
      auto ld_a = [&](int ct) {
#pragma unroll
        for (int m = 0; m < NP; ++m) aln[m] = Alog2[(16 * (m >> 2) + 4 * kq + (m & 3)) * RP + 16 * ct + j];
      };
      ld_a(0);
      v4 gxp = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float al[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) al[m] = aln[m];
        if (ct + 1 < CT) ld_a(ct + 1);
        KTUP_PIN();
        v4 gx = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NP; ++m) gx = __builtin_amdgcn_mfma_f32_16x16x4f32(al[m], gl[m >> 2][m & 3], gx, 0, 0, 0);
        if (ct > 0) emit(ct - 1, gxp);                          // the previous tile's stores go out under this tile's MFMAs
        gxp = gx;
      }
      emit(CT - 1, gxp);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- D: table gradients, K = the tile's 16 pairs (4 k-steps).  A[i = preference][k = pair] from LT / GLT,
    //         B[k = pair][n = coordinate] from the XT / GRT / GNT tiles (b32 reads, 16 consecutive coordinates per kq group)
    {
      const float* xf = reinterpret_cast<const float*>(XT);
      const float* grf = reinterpret_cast<const float*>(GRT);
      const float* gnf = reinterpret_cast<const float*>(GNT);
      float aln[PTD], agln[PTD], bxn, bgrn, bgnn;
      auto ld_a = [&](int st) {                                 // rows past LROW: any finite value (their accumulators are never flushed)
#pragma unroll
        for (int pt = 0; pt < PTD; ++pt) {
          const int pr = min(16 * pt + j, G::LROW - 1);
          aln[pt] = LT[pr * 17 + 4 * st + kq];
          agln[pt] = GLT[pr * 17 + 4 * st + kq];
        }
      };
      auto ld_b = [&](int st, int ct) {
        const int col = 16 * ct + j;
        const int off = (4 * st + kq) * (NCH * 4) + col;
        const bool in = col < D;
        bxn = in ? xf[off] : 0.f; bgrn = in ? grf[off] : 0.f; bgnn = in ? gnf[off] : 0.f;
      };
      ld_a(0);
      ld_b(0, 0);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        float al[PTD], agl[PTD];
#pragma unroll
        for (int pt = 0; pt < PTD; ++pt) { al[pt] = aln[pt]; agl[pt] = agln[pt]; }
        if (st + 1 < 4) ld_a(st + 1);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float bx = bxn, bgr = bgrn, bgn = bgnn;
          if (ct + 1 < CT) ld_b(st, ct + 1);
          else if (st + 1 < 4) ld_b(st + 1, 0);
          KTUP_PIN();
#pragma unroll
          for (int pt = 0; pt < PTD; ++pt) {
            accA[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(agl[pt], bx, accA[pt][ct], 0, 0, 0);
            accA[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[pt], bgr, accA[pt][ct], 0, 0, 0);
            accC[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[pt], bgn, accC[pt][ct], 0, 0, 0);
          }
        }
      }
      if constexpr (THIN) {        // the four live rows of the last preference tile: one pair (K = 1) per instruction
        const int prow = (16 * PTD + (lane & 3)) * 17;
        float tal, tagl, tb[NG][3];
        auto ld_t = [&](int pr) {
          tal = LT[prow + pr]; tagl = GLT[prow + pr];
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int col = 64 * g + lane;
            const bool in = col < D;
            const int off = pr * (NCH * 4) + col;
            tb[g][0] = in ? xf[off] : 0.f; tb[g][1] = in ? grf[off] : 0.f; tb[g][2] = in ? gnf[off] : 0.f;
          }
        };
        ld_t(0);
#pragma unroll
        for (int pr = 0; pr < 16; ++pr) {
          const float al = tal, agl = tagl;
          float b[NG][3];
#pragma unroll
          for (int g = 0; g < NG; ++g) { b[g][0] = tb[g][0]; b[g][1] = tb[g][1]; b[g][2] = tb[g][2]; }
          if (pr + 1 < 16) ld_t(pr + 1);
          KTUP_PIN();
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            thinA[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(agl, b[g][0], thinA[g], 0, 0, 0);
            thinA[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(al, b[g][1], thinA[g], 0, 0, 0);
            thinC[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(al, b[g][2], thinC[g], 0, 0, 0);
          }
        }
      }
    }
#undef KTUP_PIN
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    nu1 = nu2; ni1 = ni2; ne1 = ne2; nu2 = nu3; ni2 = ni3;
    cb ^= 1;
  }
  // ---- flush the table gradients: lane (kq, n) holds preference 16 pt + 4 kq + reg, coordinate 16 ct + n
#pragma unroll
  for (int pt = 0; pt < PTD; ++pt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int p = 16 * pt + 4 * kq + reg, c = 16 * ct + j;
        if (p < a.P && c < D) {
          const float va = accA[pt][ct][reg], vc = accC[pt][ct][reg];
          if (va != 0.f) atomicAdd(a.gA + (int64_t)p * D + c, va);
          if (vc != 0.f) atomicAdd(a.gC + (int64_t)p * D + c, vc);
        }
      }
  if constexpr (THIN) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int p = 16 * PTD + reg, c = 64 * g + lane;
        if (p < a.P && c < D) {
          const float va = thinA[g][reg], vc = thinC[g][reg];
          if (va != 0.f) atomicAdd(a.gA + (int64_t)p * D + c, va);
          if (vc != 0.f) atomicAdd(a.gC + (int64_t)p * D + c, vc);
        }
      }
  }
}

template <typename G, bool ROWOUT>
int launch_r(const BArgs& a, hipStream_t st, const char* name) {
  static_assert(G::LDS <= 160 * 1024, "LDS budget");
  (void)hipFuncSetAttribute((const void*)pref_bwd_mc_kernel<G, ROWOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
  const int64_t ntiles = (a.n + 15) / 16;
  const int grid = grid_for((ntiles + G::NW - 1) / G::NW, 256);
  hipLaunchKernelGGL((pref_bwd_mc_kernel<G, ROWOUT>), dim3(grid), dim3(G::NW * 64), G::LDS, st, a);
  return check_launch(name);
}

template <typename G>
int launch(const BArgs& a, hipStream_t st, const char* name) {
  return a.GU ? launch_r<G, true>(a, st, name) : launch_r<G, false>(a, st, name);
}

template <int NCH, int NP>
int launch_e(const BArgs& a, hipStream_t st, const char* name) {
  if (a.gumbel != KTUP_GUMBEL_OFF) {
    if (a.E) return launch<BGeom<NCH, NP, true, true>>(a, st, name);
    return launch<BGeom<NCH, NP, false, true>>(a, st, name);
  }
  if (a.E) return launch<BGeom<NCH, NP, true, false>>(a, st, name);
  return launch<BGeom<NCH, NP, false, false>>(a, st, name);
}

template <int NCH>
int launch_np(const BArgs& a, int np, hipStream_t st, const char* name) {
  if (np <= 4) return launch_e<NCH, 4>(a, st, name);
  if (np <= 5) return launch_e<NCH, 5>(a, st, name);
  return launch_e<NCH, 8>(a, st, name);
}

}  // namespace

// Returns KTUP_OK / an error, or 1 when (d, P) is not covered (the caller runs pref_bwd_kernel).
int pref_bwd_mc(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                int64_t ent_pad, const float* Alog, const float* Ar, const float* Cn, int dp, float beta, int n_pref, int d,
                const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                uint64_t offset, const float* gscore, float* gU, float* gI, float* gE, float* gA, float* gC, hipStream_t st,
                const char* name, float* GU, float* GV) {
  if (n_pref > 32) return 1;
  // d = 256 always, and small batches (<= 256 tiles: at most one tile per CU) at the other widths: four waves per tile
  if (d == 256 || ((d == 64 || d == 100 || d == 128) && n <= opt_bwd_wide_max()))
    return pref_bwd_mc_wide(U, ldu, I, ldi, E, lde, item2ent, ent_pad, Alog, Ar, Cn, dp, beta, n_pref, d, u_ids, i_ids, n, l1, gumbel_mode,
                            uniform, seed, offset, gscore, gU, gI, gE, gA, gC, st, name, GU, GV);
  if (d != 64 && d != 100 && d != 128) return 1;
  if ((ldu | ldi | lde) & 3) return 1;
  if ((ldu >> 2) > 0xffffffffll || (ldi >> 2) > 0xffffffffll || (lde >> 2) > 0xffffffffll) return 1;
  BArgs a{};
  a.U = reinterpret_cast<const v4*>(U); a.I = reinterpret_cast<const v4*>(I); a.E = reinterpret_cast<const v4*>(E);
  a.ldu4 = (uint32_t)(ldu >> 2); a.ldi4 = (uint32_t)(ldi >> 2); a.lde4 = (uint32_t)(lde >> 2);
  a.item2ent = item2ent;
  a.Alog = Alog; a.Ar = Ar; a.Cn = Cn; a.dp = dp; a.P = n_pref; a.l1 = l1; a.beta = beta;
  a.u_ids = u_ids; a.i_ids = i_ids; a.n = n; a.ent_pad = ent_pad;
  a.gscore = gscore; a.gU = gU; a.gI = gI; a.gE = gE; a.gA = gA; a.gC = gC;
  a.GU = GU; a.GV = GV;
  a.gumbel = gumbel_mode; a.uniform = uniform; a.seed = seed; a.offset = offset;
  const int np = (n_pref + 3) / 4;
  if (d == 64) return launch_np<16>(a, np, st, name);
  if (d == 100) return launch_np<25>(a, np, st, name);
  return launch_np<32>(a, np, st, name);
}

}  // namespace ktup
