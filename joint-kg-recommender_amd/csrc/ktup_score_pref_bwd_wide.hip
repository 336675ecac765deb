// K5 / K6 / K7 backward on the matrix cores for d = 256 (BASELINE config 5): the coordinate-sliced variant of pref_bwd_mc_kernel.
//
// Same math, same MFMA operand layouts and the same phases A-D as ktup_score_pref_bwd_mc.hip (read its header first;
// transUP.py:69-115 / jTransUP.py:122-143,250-262 differentiated).  What changes is who owns what: at d = 256 one wave cannot
// hold a 16-pair tile (4 x 16 coordinate tiles of q / n / gz / gq plus 2 x 2 x 16 table-gradient accumulators per lane, and four
// 16 KB LDS tiles next to 100 KB of tables), so FOUR waves share a tile and each owns 64 of the 256 coordinates:
//   * every contraction over coordinates (logits L, s = q.n, av = gz.n, gL) is a per-wave partial over its 64 coordinates,
//     summed across the four waves through a small LDS array and a workgroup barrier -- always in wave order 0..3, so all four
//     waves hold bit-identical logits (the ST-Gumbel argmax must agree) ;
//   * every per-coordinate quantity (n, r, z, gz, gq, gn, gx, the row gradients, the table-gradient accumulators of phase D)
//     lives only in the owning wave: 4 coordinate tiles per lane instead of 16 ;
//   * the three tables sit in LDS ONCE per workgroup, one row per preference plus a shared zero row that stands in for the
//     empty slots of the last 16-slot tile (P = 20 fills 4 of its 16 slots), row pitch 65 float4 (odd: conflict-free b128 reads).
// LDS at P = 20: 65.5 KB tables + 8 KB reduction scratch + 4 x 19 KB wave-private tiles = 152 KB, one workgroup per CU.
// MFMA work per tile is unchanged (~1000 16x16x4 instructions) but split four ways: ~250 per wave.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ktup_common.h"
#include "ktup_lane_swap.h"
#include "ktup_pref_geom.h"

namespace ktup {
namespace {

template <int NP_, bool HASE_, bool HARD_>
struct WGeom {
  static constexpr int NP = NP_;
  static constexpr bool HASE = HASE_, HARD = HARD_;
  static constexpr int D = 256, NCH = 64, NWC = 4, NCW = 16, CTW = 4;   // 4 waves x 16 chunks = 4 coordinate tiles of 16 each
  static constexpr int PT = (NP + 3) / 4, TROW = 16 * PT;
  static constexpr int ROWS = 4 * NP + 1;                 // preferences 0 .. 4 NP - 1 (rows >= P zero) + the zero row
  static constexpr int RP4 = NCH + 1, RPF = 4 * RP4;      // table row pitch: 65 float4 = 260 floats
  static constexpr int TAB_F4 = ROWS * RP4;
  static constexpr int TP4 = NCW + 1, TPF = 4 * TP4;      // wave tile row pitch: 17 float4 = 68 floats
  static constexpr int TILE_F4 = 16 * TP4;
  static constexpr int LT_F = TROW * 17;
  static constexpr int NOISE_F = HARD ? 16 * TROW : 0;
  static constexpr int RED_F = NWC * 64 * PT * 4;         // cross-wave partials of lg / gl
  static constexpr size_t SHARED_BYTES = (size_t)3 * TAB_F4 * 16 + (size_t)RED_F * 4 + 2 * NWC * 16 * 4;
  static constexpr size_t WAVE_BYTES = ((size_t)3 * TILE_F4 * 16 + (size_t)2 * LT_F * 4 + 3 * 16 * 4 + (size_t)NOISE_F * 4 + 15) & ~(size_t)15;
  static constexpr size_t LDS = SHARED_BYTES + NWC * WAVE_BYTES;
};

struct WArgs {
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
  float *GU, *GV;                // ROWOUT: per-pair row gradients (n x 256)
  int gumbel;
  const float* uniform;
  uint64_t seed, offset;
};

template <typename G, bool ROWOUT>
__global__ __launch_bounds__(256) void pref_bwd_wide_kernel(WArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  constexpr int NP = G::NP, D = G::D, NCH = G::NCH, NCW = G::NCW, CTW = G::CTW, PT = G::PT, TROW = G::TROW;
  constexpr int RP4 = G::RP4, RPF = G::RPF, TP4 = G::TP4, TPF = G::TPF;
  constexpr bool HASE = G::HASE, HARD = G::HARD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* AlogT = reinterpret_cast<v4*>(smem);                    // [ROWS][RP4]
  v4* ArT = AlogT + G::TAB_F4;
  v4* CnT = ArT + G::TAB_F4;
  float* red = reinterpret_cast<float*>(CnT + G::TAB_F4);     // [4 waves][64 lanes][PT * 4]
  float* reds = red + G::RED_F;                               // [4][16]
  float* redav = reds + G::NWC * 16;                          // [4][16]
  const float* Alog2 = reinterpret_cast<const float*>(AlogT);
  const float* Ar2 = reinterpret_cast<const float*>(ArT);
  const float* Cn2 = reinterpret_cast<const float*>(CnT);
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // this wave's coordinate slice: [64 w, 64 w + 64)
  char* wbase = reinterpret_cast<char*>(redav + G::NWC * 16) + (size_t)w * G::WAVE_BYTES;
  v4* XT = reinterpret_cast<v4*>(wbase);                      // x    [16 pairs][TP4]   (this wave's 16 chunks)
  v4* QT = XT + G::TILE_F4;                                   // q, later gn (a lane overwrites exactly what it read)
  v4* GRT = QT + G::TILE_F4;                                  // gr = gz
  float* LT = reinterpret_cast<float*>(GRT + G::TILE_F4);     // [TROW][17]  beta * w   (preference major)
  float* GLT = LT + G::LT_F;                                  // [TROW][17]  gL / 2
  int32_t* sid = reinterpret_cast<int32_t*>(GLT + G::LT_F);   // [3][16]
  float* noise = reinterpret_cast<float*>(sid + 48);          // HARD: [16][TROW]
  const int P = a.P;
  // ---- stage the three tables: row p = preference p (zero beyond P), row pitch 65 float4
  {
    const int dp = a.dp;
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < G::TAB_F4; idx += 256) {
      const int row = idx / RP4, c = idx - row * RP4;
      const bool ok = row < P && c < NCH;
      AlogT[idx] = ok ? *reinterpret_cast<const v4*>(a.Alog + row * dp + 4 * c) : zero;
      ArT[idx] = ok ? *reinterpret_cast<const v4*>(a.Ar + row * dp + 4 * c) : zero;
      CnT[idx] = ok ? *reinterpret_cast<const v4*>(a.Cn + row * dp + 4 * c) : zero;
    }
  }
  __syncthreads();
  // ---- loop-invariant lane geometry
  // (a) slot rows, A operand of the K = coordinate GEMMs (phases A1, B): slot j of tile tt <-> preference 16 tt + 4 (j & 3) + (j >> 2)
  int ra[PT];
#pragma unroll
  for (int tt = 0; tt < PT; ++tt) {
    const int p = 16 * tt + 4 * (j & 3) + (j >> 2);
    ra[tt] = (p < P ? p : P) * RP4 + NCW * w + kq;            // float4 index; + 4 g per k group
  }
  // (b) preference rows, A operand of the K = preference GEMMs (phases A2, C): k step m, k index kq <-> preference 4 m + kq
  int rb[NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    const int p = 4 * m + kq;
    rb[m] = (p < P ? p : P) * RPF + 4 * NCW * w + j;          // float index; + 16 ct per coordinate tile
  }
  const int grow_l = lane >> 4, gch = lane & 15;              // gather: row grow_l + 4 jj, chunk gch of the wave's slice
  const bool l1 = a.l1 != 0;
  const float beta = a.beta;
  v4 accA[PT][CTW], accC[PT][CTW];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) { accA[pt][ct] = (v4){0.f, 0.f, 0.f, 0.f}; accC[pt][ct] = accA[pt][ct]; }
  const int64_t ntiles = (a.n + 15) / 16;
  for (int64_t tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
    const int64_t row0 = tile_id * 16;
    if (lane < 16) {
      const int64_t gr = row0 + lane;
      const bool ok = gr < a.n;
      const int64_t uid = ok ? a.u_ids[gr] : 0, iid = ok ? a.i_ids[gr] : 0;
      sid[lane] = (int32_t)uid;
      sid[16 + lane] = (int32_t)iid;
      sid[32 + lane] = HASE ? a.item2ent[iid] : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- gather this wave's 64 coordinates of the 16 pairs: x and q tiles
    {
      v4 uu[4], vv[4], ee[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int r = grow_l + 4 * jj;
        const uint32_t idu = (uint32_t)sid[r], idi = (uint32_t)sid[16 + r];
        uu[jj] = a.U[(uint64_t)idu * a.ldu4 + (uint32_t)(NCW * w + gch)];
        vv[jj] = a.I[(uint64_t)idi * a.ldi4 + (uint32_t)(NCW * w + gch)];
        if (HASE) {
          const uint32_t ide = (uint32_t)sid[32 + r];
          ee[jj] = a.E[(uint64_t)ide * a.lde4 + (uint32_t)(NCW * w + gch)];
        }
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int r = grow_l + 4 * jj;
        const v4 ve = HASE ? vv[jj] + ee[jj] : vv[jj];
        XT[r * TP4 + gch] = uu[jj] + ve;
        QT[r * TP4 + gch] = uu[jj] + (-ve);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- A1: partial L^T over this wave's coordinates, then the cross-wave sum
    v4 lg[PT];
#pragma unroll
    for (int tt = 0; tt < PT; ++tt) lg[tt] = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < CTW; ++g) {
      const v4 bv = XT[j * TP4 + 4 * g + kq];
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) {
        const v4 av = AlogT[ra[tt] + 4 * g];
#pragma unroll
        for (int c = 0; c < 4; ++c) lg[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[c], lg[tt], 0, 0, 0);
      }
    }
    {
      v4* myred = reinterpret_cast<v4*>(red) + (w * 64 + lane) * PT;
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) myred[tt] = lg[tt];
      __syncthreads();
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) {
        v4 sum = reinterpret_cast<const v4*>(red)[(0 * 64 + lane) * PT + tt];
#pragma unroll
        for (int ww = 1; ww < G::NWC; ++ww) sum += reinterpret_cast<const v4*>(red)[(ww * 64 + lane) * PT + tt];
        lg[tt] = sum;
      }
    }
    // ---- ST-Gumbel gate (as pref_bwd_mc_kernel; every wave does it on the same full logits and the same draws)
    v4 ysoft[PT];
    if constexpr (HARD) {
      const int64_t grow = min(row0 + j, a.n - 1);
      const uint64_t base = (uint64_t)grow * (uint64_t)P;
      if (a.gumbel == KTUP_GUMBEL_PHILOX) {
        const uint64_t i0 = base + a.offset, fb = i0 >> 2, lb = (i0 + (uint64_t)P - 1) >> 2;
        const Philox ph(a.seed);
        for (uint64_t b = fb + kq; b <= lb; b += 4) {
          const uint4 r = ph(b, 0x4b545550ull /* "KTUP" stream tag */);
          const uint32_t wds[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
          for (int wd = 0; wd < 4; ++wd) {
            const int64_t pp = (int64_t)((b << 2) + wd) - (int64_t)i0;
            if (pp >= 0 && pp < P) noise[j * TROW + (int)pp] = gumbel_from_uniform(u01(wds[wd]));
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
        for (int reg = 0; reg < 4; ++reg) {
          const int pp = 16 * tt + 4 * reg + kq;
          float v = -INFINITY;
          if (pp < P) {
            const float g = a.gumbel == KTUP_GUMBEL_PHILOX ? noise[j * TROW + pp] : gumbel_from_uniform(a.uniform[base + pp]);
            v = lg[tt][reg] + g;
            if (v > best || bp == 0x7fffffff) { best = v; bp = pp; }
          }
          lg[tt][reg] = v;
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
          const float e = expf(lg[tt][reg] - best);
          ysoft[tt][reg] = e;
          den += e;
          lg[tt][reg] = (16 * tt + 4 * reg + kq == bp) ? 1.f : 0.f;
        }
      const float inv = 1.f / allsum_kq(den);
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) ysoft[tt] = ysoft[tt] * inv;
    }
    // ---- A2: n^T, (q + r)^T of this wave's 4 coordinate tiles; lane (kq, j) owns coordinates 64 w + 16 ct + 4 kq + reg of pair j
    v4 nn[CTW], zz[CTW], qv[CTW];
    v4 sacc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) {
      qv[ct] = QT[j * TP4 + 4 * ct + kq];
      nn[ct] = (v4){0.f, 0.f, 0.f, 0.f};
      zz[ct] = qv[ct];
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        nn[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(Cn2[rb[m] + 16 * ct], lg[m >> 2][m & 3], nn[ct], 0, 0, 0);
        zz[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar2[rb[m] + 16 * ct], lg[m >> 2][m & 3], zz[ct], 0, 0, 0);
      }
      sacc += qv[ct] * nn[ct];
    }
    float s = allsum_kq((sacc[0] + sacc[1]) + (sacc[2] + sacc[3]));
    if (kq == 0) reds[w * 16 + j] = s;
    __syncthreads();
    s = (reds[j] + reds[16 + j]) + (reds[32 + j] + reds[48 + j]);
    const float g = row0 + j < a.n ? a.gscore[row0 + j] : 0.f;
    v4 aacc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) {
      const v4 z = zz[ct] - s * nn[ct];
      v4 gz;
#pragma unroll
      for (int c = 0; c < 4; ++c) gz[c] = g * ddist1(z[c], l1);
      zz[ct] = gz;
      aacc += gz * nn[ct];
    }
    float av = allsum_kq((aacc[0] + aacc[1]) + (aacc[2] + aacc[3]));
    if (kq == 0) redav[w * 16 + j] = av;
    __syncthreads();
    av = (redav[j] + redav[16 + j]) + (redav[32 + j] + redav[48 + j]);
    v4 gq[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) {
      gq[ct] = zz[ct] - av * nn[ct];
      nn[ct] = -av * qv[ct] - s * zz[ct];
      GRT[j * TP4 + 4 * ct + kq] = zz[ct];
      QT[j * TP4 + 4 * ct + kq] = nn[ct];                       // gn takes q's place
    }
    // ---- B: partial gL^T = ArSlot . gr^T + CnSlot . gn^T over this wave's coordinates, then the cross-wave sum
    v4 gl[PT];
#pragma unroll
    for (int tt = 0; tt < PT; ++tt) gl[tt] = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g4 = 0; g4 < CTW; ++g4) {
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) {
        const v4 ar = ArT[ra[tt] + 4 * g4];
        const v4 cn = CnT[ra[tt] + 4 * g4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          gl[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[c], zz[g4][c], gl[tt], 0, 0, 0);
          gl[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cn[c], nn[g4][c], gl[tt], 0, 0, 0);
        }
      }
    }
    {
      v4* myred = reinterpret_cast<v4*>(red) + (w * 64 + lane) * PT;      // every wave read its logits before the `reds` barrier
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) myred[tt] = gl[tt];
      __syncthreads();
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) {
        v4 sum = reinterpret_cast<const v4*>(red)[(0 * 64 + lane) * PT + tt];
#pragma unroll
        for (int ww = 1; ww < G::NWC; ++ww) sum += reinterpret_cast<const v4*>(red)[(ww * 64 + lane) * PT + tt];
        gl[tt] = sum;
      }
    }
    if constexpr (HARD) {        // softmax Jacobian of y = softmax(l + g) applied to gw
      float dot = 0.f;
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) dot += (ysoft[tt][0] * gl[tt][0] + ysoft[tt][1] * gl[tt][1]) + (ysoft[tt][2] * gl[tt][2] + ysoft[tt][3] * gl[tt][3]);
      dot = allsum_kq(dot);
#pragma unroll
      for (int tt = 0; tt < PT; ++tt) gl[tt] = ysoft[tt] * (gl[tt] - dot);
    }
#pragma unroll
    for (int tt = 0; tt < PT; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int p = 16 * tt + 4 * reg + kq;
        LT[p * 17 + j] = beta * lg[tt][reg];
        GLT[p * 17 + j] = 0.5f * gl[tt][reg];
      }
    // ---- C: gx^T = Alog2^T . gL^T of this wave's coordinates, then the row gradients
    {
      const int64_t gr = row0 + j;
      const bool live = gr < a.n;
      const int32_t ur = sid[j], ir = sid[16 + j], er = sid[32 + j];
      float* pu = a.gU + (int64_t)ur * a.ldu4 * 4;
      float* pi = a.gI + (int64_t)ir * a.ldi4 * 4;
      float* pe = (HASE && er != a.ent_pad) ? a.gE + (int64_t)er * a.lde4 * 4 : nullptr;
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) {
        v4 gx = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NP; ++m)
          gx = __builtin_amdgcn_mfma_f32_16x16x4f32(Alog2[rb[m] + 16 * ct], gl[m >> 2][m & 3], gx, 0, 0, 0);
        const int c0 = 4 * NCW * w + 16 * ct + 4 * kq;
        if (live) {
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
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- D: table gradients of this wave's coordinates, K = the tile's 16 pairs
    {
      const float* xf = reinterpret_cast<const float*>(XT);
      const float* grf = reinterpret_cast<const float*>(GRT);
      const float* gnf = reinterpret_cast<const float*>(QT);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        float al[PT], agl[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) { al[pt] = LT[(16 * pt + j) * 17 + 4 * st + kq]; agl[pt] = GLT[(16 * pt + j) * 17 + 4 * st + kq]; }
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
          const int off = (4 * st + kq) * TPF + 16 * ct + j;
          const float bx = xf[off], bgr = grf[off], bgn = gnf[off];
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            accA[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(agl[pt], bx, accA[pt][ct], 0, 0, 0);
            accA[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[pt], bgr, accA[pt][ct], 0, 0, 0);
            accC[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[pt], bgn, accC[pt][ct], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();     // the next tile rewrites `red` and the wave tiles
  }
  // ---- flush the table gradients of this wave's coordinates
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int p = 16 * pt + 4 * kq + reg, c = 4 * NCW * w + 16 * ct + j;
        if (p < P) {
          const float va = accA[pt][ct][reg], vc = accC[pt][ct][reg];
          if (va != 0.f) atomicAdd(a.gA + (int64_t)p * D + c, va);
          if (vc != 0.f) atomicAdd(a.gC + (int64_t)p * D + c, vc);
        }
      }
}

template <typename G, bool ROWOUT>
int launch_r(const WArgs& a, hipStream_t st, const char* name) {
  static_assert(G::LDS <= 160 * 1024, "LDS budget");
  (void)hipFuncSetAttribute((const void*)pref_bwd_wide_kernel<G, ROWOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
  const int64_t ntiles = (a.n + 15) / 16;
  const int grid = grid_for(ntiles, 256);                  // one workgroup (4 waves) per CU: the LDS footprint allows no more
  hipLaunchKernelGGL((pref_bwd_wide_kernel<G, ROWOUT>), dim3(grid), dim3(256), G::LDS, st, a);
  return check_launch(name);
}

template <typename G>
int launch(const WArgs& a, hipStream_t st, const char* name) {
  return a.GU ? launch_r<G, true>(a, st, name) : launch_r<G, false>(a, st, name);
}

template <int NP>
int launch_e(const WArgs& a, hipStream_t st, const char* name) {
  if (a.gumbel != KTUP_GUMBEL_OFF) {
    if (a.E) return launch<WGeom<NP, true, true>>(a, st, name);
    return launch<WGeom<NP, false, true>>(a, st, name);
  }
  if (a.E) return launch<WGeom<NP, true, false>>(a, st, name);
  return launch<WGeom<NP, false, false>>(a, st, name);
}

}  // namespace

// Returns KTUP_OK / an error, or 1 when the shape is not covered (the caller runs pref_bwd_kernel).
int pref_bwd_mc_wide(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                     int64_t ent_pad, const float* Alog, const float* Ar, const float* Cn, int dp, float beta, int n_pref, int d,
                     const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                     uint64_t offset, const float* gscore, float* gU, float* gI, float* gE, float* gA, float* gC, hipStream_t st,
                     const char* name, float* GU, float* GV) {
  if (d != 256 || n_pref > 20) return 1;                   // LDS: (P + 1) table rows x 3 next to the four waves' tiles
  if ((ldu | ldi | lde) & 3) return 1;
  if ((ldu >> 2) > 0xffffffffll || (ldi >> 2) > 0xffffffffll || (lde >> 2) > 0xffffffffll) return 1;
  WArgs a;
  a.U = reinterpret_cast<const v4*>(U); a.I = reinterpret_cast<const v4*>(I); a.E = reinterpret_cast<const v4*>(E);
  a.ldu4 = (uint32_t)(ldu >> 2); a.ldi4 = (uint32_t)(ldi >> 2); a.lde4 = (uint32_t)(lde >> 2);
  a.item2ent = item2ent;
  a.Alog = Alog; a.Ar = Ar; a.Cn = Cn; a.dp = dp; a.P = n_pref; a.l1 = l1; a.beta = beta;
  a.u_ids = u_ids; a.i_ids = i_ids; a.n = n; a.ent_pad = ent_pad;
  a.gscore = gscore; a.gU = gU; a.gI = gI; a.gE = gE; a.gA = gA; a.gC = gC;
  a.GU = GU; a.GV = GV;
  a.gumbel = gumbel_mode; a.uniform = uniform; a.seed = seed; a.offset = offset;
  const int np = (n_pref + 3) / 4;
  if (np <= 4) return launch_e<4>(a, st, name);
  return launch_e<5>(a, st, name);
}

}  // namespace ktup
