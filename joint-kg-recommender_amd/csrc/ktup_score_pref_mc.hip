// K5-K7 forward, soft gate, compile-time geometry: the matrix-core kernel for the common embedding sizes.
//
// Reference path: jTransUP/models/transUP.py:69-82,105-115 (TUP) and jTransUP/models/jTransUP.py:122-143,250-262 (KTUP):
//   x = u + i (+ e),  logits = x . A^T,  r = logits . Ar,  n = logits . Cn,  q = u - i (- e),
//   score = sum_k dist( q + r - (q . n) n )_k         (A, Ar, Cn: the mixed / pre-scaled tables of ktup_pref_prepare)
//
// Formulation (same as pref_fwd4 in ktup_score_pref.hip): a wave owns 16 pairs; logits^T = A . x^T and then
// r^T = Ar^T . logits^T, n^T = Cn^T . logits^T on v_mfma_f32_16x16x4_f32, pairs along the MFMA columns, so that the
// stage-1 accumulators ARE the stage-2 B operands.
//
// Why a second kernel: measurements on MI355X (tools/coissue_bench.hip, tools/valu_rate_bench.hip, profiles/) show that
// fp32 MFMA and VALU instructions do NOT overlap on a gfx950 SIMD -- not inside a wave, not across waves -- so this
// kernel's time is (MFMA passes + VALU instructions + exposed memory).  pref_fwd4 spends ~4000 clk per tile in MFMA
// and ~3500 clk in address arithmetic, bounds checks and scalar float4 math, because its geometry is run-time data.
// Here everything about the geometry is a template constant:
//   * gather: (row, chunk) of each of a lane's J loads is loop invariant; one v_mad_u64_u32 + one v_lshl_add_u64 per
//     row load (ids read from LDS with immediate offsets);
//   * all LDS operand addresses are one base register + immediates; no bounds checks in the matrix phases;
//   * x / q / distance math on <4 x float> values -> v_pk_add_f32 / v_pk_fma_f32;
//   * when the last 16-slot preference tile holds <= 4 preferences (P = 20), its logits come from
//     v_mfma_f32_4x4x1_16B_f32 (16 blocks = 4 pair groups x 4 k-quarters, 8 clk each) instead of a 3/4-empty 16x16x4.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "ktup_common.h"
#include "ktup_pref_geom.h"
#include "ktup_lane_swap.h"

namespace ktup {
namespace {

template <int NCH_, int NP_, bool HASE_, bool HARD_>
struct McGeom {
  static constexpr int NCH = NCH_, NP = NP_;
  static constexpr bool HASE = HASE_, HARD = HARD_;   // HARD: straight-through Gumbel gate (forward value = one-hot)
  static constexpr int D = 4 * NCH;
  static constexpr int KG = (D + 15) / 16;               // stage-1 k groups of 16 coordinates
  static constexpr int CT = KG;                          // stage-2 coordinate tiles of 16
  static constexpr int REM = NP * 4 - ((NP * 4 - 1) / 16) * 16;   // slots used in the last 16-slot tile (multiple of 4)
  static constexpr bool REM4 = (NP > 4) && REM == 4;     // last tile = one group of <= 4 preferences -> 4x4x1 path
  static constexpr int PT = (NP + 3) / 4;                // 16-slot preference tiles (including a REM4 tile)
  static constexpr int PTF = REM4 ? PT - 1 : PT;         // tiles computed with 16x16x4
  static constexpr bool TAIL1 = NCH - 4 * (KG - 1) == 1;  // the last k group / coordinate tile holds ONE float4 chunk (d = 100):
  static constexpr int KGF = TAIL1 ? KG - 1 : KG;        //   stage 1 finishes with one b32-operand MFMA instead of four,
  static constexpr int CTF = TAIL1 ? CT - 1 : CT;        //   stage 2 does those 4 coordinates with 4x4x1 MFMAs
  static constexpr int J = (16 * NCH + 63) / 64;         // float4 loads per lane, table and tile
  static constexpr int TOTAL = 16 * NCH;
  static constexpr int PITCHA4 = 4 * KG + 1;             // odd float4 pitch of the slot-ordered logit table
  static constexpr int KQ = (NCH + 3) / 4;               // REM4: float4 chunks per k-quarter
  static constexpr int TROW = 4 * NP;
  static constexpr int TPITCH = 16 * CT + ((16 * CT) % 32 == 0 ? 16 : 0);   // == 16 (mod 32)
  static constexpr int A_F4 = PTF * 16 * PITCHA4;        // float4s of the 16x16x4 logit table
  static constexpr int A4_F4 = REM4 ? 4 * 4 * KQ : 0;    // REM4 table: [4 prefs][4 quarters][KQ] float4
  static constexpr int HP = NCH | 1;                     // HARD: odd float4 pitch of the row-major tables (one row is looked up per pair)
  static constexpr int T_F = HARD ? TROW * HP * 4 : TROW * TPITCH;   // floats per stage-2 table
  static constexpr size_t TABLE_BYTES = (size_t)(A_F4 + A4_F4) * 16 + (size_t)2 * T_F * 4;
  static constexpr int XT_F4 = 16 * NCH + 3;             // x / q tile + 3 zero chunks (stage-1 reads run past the last row)
  static constexpr int NOISE_F = HARD ? 16 * TROW : 0;   // HARD: Gumbel noise of the tile, [pair][preference]
  static constexpr size_t WAVE_BYTES = ((size_t)XT_F4 * 16 + 3 * 16 * 4 + (size_t)NOISE_F * 4 + 15) & ~(size_t)15;
  static constexpr int NW_MAX = (int)((160 * 1024 - TABLE_BYTES) / WAVE_BYTES);
  static constexpr int NW = NW_MAX >= 16 ? 16 : NW_MAX >= 4 ? (NW_MAX & ~3) : NW_MAX;   // < 2: the geometry does not fit (d = 256 with P > 20)
};

struct McArgs {
  const v4 *U, *I, *E;
  uint32_t ldu4, ldi4, lde4;
  const int32_t* item2ent;
  const float *Alog, *Ar, *Cn;   // prepared tables, row pitch dp floats
  int dp, P, l1;
  int gumbel;                    // KTUP_GUMBEL_* (HARD kernels): INPUT reads `uniform` (n x P), PHILOX draws (seed, offset)
  const float* uniform;
  uint64_t seed, offset;
  const int64_t *u_ids, *i_ids;
  int64_t n;
  float* score;
  int nt;                        // row gathers with the nontemporal hint (option nt_gather: tables far beyond the Infinity Cache)
};

// L1: the distance is compile-time too -- with a run-time flag the compiler evaluates |z| AND z^2 for every coordinate and
// selects (v_and + v_cndmask per element: ~70 of a tile's ~330 VALU instructions).
template <typename G, bool L1>
__global__ __launch_bounds__(G::NW * 64) void pref_fwd_mc_kernel(McArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  constexpr int NCH = G::NCH, NP = G::NP, KG = G::KG, CT = G::CT, PTF = G::PTF, J = G::J, TOTAL = G::TOTAL;
  constexpr int PITCHA4 = G::PITCHA4, TPITCH = G::TPITCH, KQ = G::KQ;
  constexpr bool HASE = G::HASE, REM4 = G::REM4, TAIL1 = G::TAIL1, HARD = G::HARD;
  constexpr int HP = G::HP;
  constexpr int KGF = G::KGF, CTF = G::CTF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* AlogS = reinterpret_cast<v4*>(smem);                       // [PTF * 16 slots][PITCHA4]
  v4* A4S = AlogS + G::A_F4;                                     // REM4: [4 prefs][4 quarters][KQ]
  float* CnS = reinterpret_cast<float*>(A4S + G::A4_F4);         // [TROW][TPITCH]
  float* ArS = CnS + G::T_F;
  const int t = threadIdx.x, lane = t & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nthr = blockDim.x, nwr = nthr >> 6;                  // waves actually launched (<= G::NW; fewer for small batches)
  char* wbase = reinterpret_cast<char*>(ArS + G::T_F) + (size_t)w * G::WAVE_BYTES;
  v4* xt = reinterpret_cast<v4*>(wbase);                         // [16 * NCH] + 3 zero chunks
  int32_t* sid = reinterpret_cast<int32_t*>(xt + G::XT_F4);      // [3][16]
  float* noise = reinterpret_cast<float*>(sid + 48);             // HARD: [16][TROW]
  // ---- stage the tables once per workgroup
  {
    // float4 granularity (table rows are 16-byte aligned with a pitch that is a multiple of 4 floats): for a B=512 step this
    // staging is a visible part of the launch
    const int P = a.P, dp = a.dp;
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
    for (int idx = t; idx < G::A_F4; idx += nthr) {
      const int srow = idx / PITCHA4, c = idx - srow * PITCHA4;
      const int tt = srow >> 4, i = srow & 15;
      const int p = 16 * tt + 4 * (i & 3) + (i >> 2);            // slot -> preference (block transposed)
      AlogS[idx] = (p < P && c < NCH) ? *reinterpret_cast<const v4*>(a.Alog + p * dp + 4 * c) : zero;
    }
    if (REM4) {
      for (int idx = t; idx < G::A4_F4; idx += nthr) {     // [pref i][quarter kp][KQ chunks]
        const int i = idx / (4 * KQ), rem = idx - i * (4 * KQ);
        const int kp = rem / KQ, kk = rem - kp * KQ;
        const int p = 16 * PTF + i, c = KQ * kp + kk;
        A4S[idx] = (p < P && c < NCH) ? *reinterpret_cast<const v4*>(a.Alog + p * dp + 4 * c) : zero;
      }
    }
    {
      constexpr int pitch4 = HARD ? HP : TPITCH / 4;
      v4* Cn4 = reinterpret_cast<v4*>(CnS);
      v4* Ar4 = reinterpret_cast<v4*>(ArS);
      for (int idx = t; idx < G::T_F / 4; idx += nthr) {
        const int p = idx / pitch4, c = idx - p * pitch4;
        const bool ok = p < P && c < NCH;
        Cn4[idx] = ok ? *reinterpret_cast<const v4*>(a.Cn + p * dp + 4 * c) : zero;
        Ar4[idx] = ok ? *reinterpret_cast<const v4*>(a.Ar + p * dp + 4 * c) : zero;
      }
    }
    if (lane < 3) xt[16 * NCH + lane] = (v4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  // ---- loop-invariant lane geometry
  int grow[J], gc[J];                                            // (row, chunk) of this lane's jj-th load
#pragma unroll
  for (int jj = 0; jj < J; ++jj) {
    const int e = lane + 64 * jj;
    const bool past = e >= TOTAL;                                // lanes past the tile re-read chunk 0 of row 0 (never stored)
    grow[jj] = past ? 0 : e / NCH;
    gc[jj] = past ? 0 : e % NCH;
  }
  const bool last_ok = lane + 64 * (J - 1) < TOTAL;
  const v4* xb = xt + j * NCH + kq;                              // stage-1 B operand / q rows of this lane
  const v4* tab = AlogS + j * PITCHA4 + kq;                      // stage-1 A operand
  const float* tn0 = CnS + kq * TPITCH + j;
  const float* tr0 = ArS + kq * TPITCH + j;
  constexpr bool l1 = L1;
  const int64_t ntiles = (a.n + 15) / 16;
  const int64_t wstride = (int64_t)gridDim.x * nwr;
  bool first = true;
  for (int64_t tile_id = (int64_t)blockIdx.x * nwr + w; tile_id < ntiles; tile_id += wstride) {
    // wave-uniform 64-bit tile origin (scalar registers) + 32-bit lane offsets: kept as 64-bit per-lane row indices, these were
    // six registers the kernel had to spill at 128 -- with scratch reloads in front of the id loads and inside the matrix phases
    const int64_t row0 = tile_id * 16;
    const int rem = (int)(a.n - row0 < 16 ? a.n - row0 : 16);
    if (first && lane < 16) {
      const bool ok = lane < rem;
      const int64_t uid = ok ? (a.u_ids + row0)[lane] : 0, iid = ok ? (a.i_ids + row0)[lane] : 0;
      sid[lane] = (int32_t)uid;
      sid[16 + lane] = (int32_t)iid;
      sid[32 + lane] = HASE ? a.item2ent[iid] : 0;
    }
    first = false;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- gather: x -> LDS tile, q in registers.  Two batches of row loads: 16 waves x 3 x JA KB in flight per CU is
    // already several times the latency-bandwidth product, and 3 x J live float4 would push the kernel past 128 VGPRs.
    v4 q[J];
    constexpr int JA = G::NW >= 16 ? (J + 1) / 2 : J;
#pragma unroll
    for (int j0 = 0; j0 < J; j0 += JA) {
      v4 uu[JA], vv[JA], ee[JA];
#pragma unroll
      for (int jb = 0; jb < JA; ++jb) {
        const int jj = j0 + jb;
        if (jj < J) {
          asm volatile("" : "+v"(gc[jj]));   // opaque per tile: LICM would hoist 3 x J 64-bit (table + chunk) bases and spill them
          const uint32_t idu = (uint32_t)sid[grow[jj]], idi = (uint32_t)sid[16 + grow[jj]];
          const v4* pu = a.U + ((uint64_t)idu * a.ldu4 + (uint32_t)gc[jj]);
          const v4* pv = a.I + ((uint64_t)idi * a.ldi4 + (uint32_t)gc[jj]);
          // scattered rows of tables much larger than the caches: the default policy allocates every line it will never
          // re-read (tools/gather_bench footprint: 4.68 vs 4.86 TB/s at 9.7 GB; the other way round below ~2 GB)
          uu[jb] = a.nt ? __builtin_nontemporal_load(pu) : *pu;
          vv[jb] = a.nt ? __builtin_nontemporal_load(pv) : *pv;
          if (HASE) {
            const uint32_t ide = (uint32_t)sid[32 + grow[jj]];
            const v4* pe = a.E + ((uint64_t)ide * a.lde4 + (uint32_t)gc[jj]);
            ee[jb] = a.nt ? __builtin_nontemporal_load(pe) : *pe;
          }
        }
      }
#pragma unroll
      for (int jb = 0; jb < JA; ++jb) {
        const int jj = j0 + jb;
        if (jj < J) {
          const v4 ve = HASE ? vv[jb] + ee[jb] : vv[jb];
          if (jj < J - 1 || last_ok) xt[lane + 64 * jj] = uu[jb] + ve;
          q[jj] = uu[jb] + (-ve);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- next tile's ids travel under the matrix phases
    int32_t nx_u = 0, nx_i = 0, nx_e = 0;
    const bool pre = lane < 16 && tile_id + wstride < ntiles;
    if (pre) {
      const int64_t row1 = (tile_id + wstride) * 16;
      const bool ok = lane < a.n - row1;
      const int64_t uid = ok ? (a.u_ids + row1)[lane] : 0, iid = ok ? (a.i_ids + row1)[lane] : 0;
      nx_u = (int32_t)uid; nx_i = (int32_t)iid;
      nx_e = HASE ? a.item2ent[iid] : 0;
    }
    // ---- stage 1: logits^T.  lg[tt][reg] of lane (kq, pair j) = logit of preference 16 tt + 4 reg + kq
    v4 lg[G::PT];
#pragma unroll
    for (int tt = 0; tt < G::PT; ++tt) lg[tt] = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gk = 0; gk < KGF; ++gk) {
      v4 bv = xb[4 * gk];
      if (4 * gk + 3 >= NCH) {                                   // chunks past the row (table is zero there): see REM4 note
        if (4 * gk + kq >= NCH) bv = (v4){0.f, 0.f, 0.f, 0.f};
      }
      v4 av[PTF];
#pragma unroll
      for (int tt = 0; tt < PTF; ++tt) av[tt] = tab[tt * 16 * PITCHA4 + 4 * gk];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int tt = 0; tt < PTF; ++tt) lg[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt][c], bv[c], lg[tt], 0, 0, 0);
      }
    }
    if (TAIL1) {                                                 // coordinates 16 KGF + kq: one MFMA, scalar operands
      const float bs = reinterpret_cast<const float*>(xt + j * NCH + 4 * KGF)[kq];
#pragma unroll
      for (int tt = 0; tt < PTF; ++tt) {
        const float as = reinterpret_cast<const float*>(AlogS + (tt * 16 + j) * PITCHA4 + 4 * KGF)[kq];
        lg[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, bs, lg[tt], 0, 0, 0);
      }
    }
    if (REM4) {
      // 16 blocks of 4x4x1: block b = lane >> 2 = (k-quarter kp = b >> 2 == kq, pair group pg = b & 3), A row i = lane & 3
      // = preference 16 PTF + i, B column jb = lane & 3 = pair 4 pg + jb.  D[reg i] of lane (b, jb) sums over the quarter.
      const int pg = (lane >> 2) & 3, jb = lane & 3;
      const v4* a4 = A4S + (jb * 4 + kq) * KQ;                   // row (pref jb, quarter kq)
      const v4* b4 = xt + (4 * pg + jb) * NCH + kq * KQ;         // pair 4 pg + jb, quarter kq
      v4 acc = (v4){0.f, 0.f, 0.f, 0.f}, acc1 = acc;              // two chains: a 4x4x1 result is not ready for the next issue slot
#pragma unroll
      for (int kk = 0; kk < KQ; ++kk) {
        const v4 av = a4[kk];
        v4 bv = b4[kk];
        if (3 * KQ + kk >= NCH) {                                // the last quarter runs past the row: A is zero there,
          if (kq * KQ + kk >= NCH) bv = (v4){0.f, 0.f, 0.f, 0.f};   // but keep a neighbour's inf / nan out of 0 * x
        }
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c], bv[c], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c + 1], bv[c + 1], acc1, 0, 0, 0);
        }
      }
      acc += acc1;
      // lane (kq, pair j16 = lane & 15) needs preference 16 PTF + kq of pair j16 = 4 pg + jb, summed over the 4 k-quarters
      // held by lanes l, l^16, l^32, l^48 (same block position): a reduce-scatter over kq
      const float mine = scatter_kq(acc);
      lg[PTF] = (v4){mine, 0.f, 0.f, 0.f};
    }
    // ---- q overwrites x
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      if (jj < J - 1 || last_ok) xt[lane + 64 * jj] = q[jj];
    }
    v4 dacc = (v4){0.f, 0.f, 0.f, 0.f};
    if constexpr (HARD) {
      // ---- ST-Gumbel gate (transUP.py:118-170): y = one_hot(argmax_p(logit_p + g_p)), g = -log(-log(u + eps) + eps); the
      // forward value of the straight-through estimator is exactly the one-hot, so r = Ar[p*], n = Cn[p*].
      const int64_t grow = min(row0 + j, a.n - 1);
      const uint64_t base = (uint64_t)grow * (uint64_t)a.P;
      float best = -INFINITY;
      int bp = 0x7fffffff;
      if (a.gumbel == KTUP_GUMBEL_PHILOX) {
        // same stream as the other kernels (ktup_score_pref.hip draw_uniform): word (idx & 3) of Philox block idx >> 2,
        // idx = grow * P + p + offset.  The 4 kq lanes of a pair share its <= P/4 + 1 blocks and trade through LDS.
        const uint64_t i0 = base + a.offset, fb = i0 >> 2, lb = (i0 + (uint64_t)a.P - 1) >> 2;
        const Philox ph(a.seed);
        for (uint64_t b = fb + kq; b <= lb; b += 4) {
          const uint4 r = ph(b, 0x4b545550ull /* "KTUP" stream tag */);
          const uint32_t wds[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
          for (int wd = 0; wd < 4; ++wd) {
            const int64_t pp = (int64_t)((b << 2) + wd) - (int64_t)i0;
            if (pp >= 0 && pp < a.P) noise[j * G::TROW + (int)pp] = gumbel_from_uniform(u01(wds[wd]));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int sidx = 0; sidx < NP; ++sidx) {                    // ascending p: strict > keeps the first maximum (torch.max)
        const int pp = 4 * sidx + kq;
        if (pp < a.P) {
          const float g = a.gumbel == KTUP_GUMBEL_PHILOX ? noise[j * G::TROW + pp] : gumbel_from_uniform(a.uniform[base + pp]);
          const float v = lg[sidx >> 2][sidx & 3] + g;
          if (v > best || bp == 0x7fffffff) { best = v; bp = pp; }
        }
      }
      // first-max over the 4 kq lanes: (value, preference) pairs, lower preference wins ties
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
        bp = take1 ? p1 : p0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // q tile written above
      __builtin_amdgcn_wave_barrier();
      const v4* nrow = reinterpret_cast<const v4*>(CnS) + bp * HP + kq;
      const v4* rrow = reinterpret_cast<const v4*>(ArS) + bp * HP + kq;
      v4 sacc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {                          // lane (kq, j) owns q chunks 4 ct + kq
        if (4 * ct + 3 < NCH) {
          sacc += xb[4 * ct] * nrow[4 * ct];
        } else if (4 * ct < NCH) {
          if (4 * ct + kq < NCH) sacc += xb[4 * ct] * nrow[4 * ct];
        }
      }
      const float sfull = allsum_kq((sacc[0] + sacc[1]) + (sacc[2] + sacc[3]));
      const v4 ms = (v4){-sfull, -sfull, -sfull, -sfull};
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        bool on = 4 * ct + 3 < NCH;
        if (!on && 4 * ct < NCH) on = 4 * ct + kq < NCH;
        if (4 * ct < NCH) {
          if (on) {
            const v4 tv = __builtin_elementwise_fma(ms, nrow[4 * ct], xb[4 * ct] + rrow[4 * ct]);
            if (l1) dacc += __builtin_elementwise_abs(tv);
            else dacc = __builtin_elementwise_fma(tv, tv, dacc);
          }
        }
      }
    } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- stage 2a: n^T coordinate tiles
    v4 accN[CTF];
#pragma unroll
    for (int ct = 0; ct < CTF; ++ct) {
      accN[ct] = (v4){0.f, 0.f, 0.f, 0.f};
      float ta[NP];
#pragma unroll
      for (int m = 0; m < NP; ++m) ta[m] = tn0[(16 * (m >> 2) + 4 * (m & 3)) * TPITCH + 16 * ct];
#pragma unroll
      for (int m = 0; m < NP; ++m) accN[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[m], lg[m >> 2][m & 3], accN[ct], 0, 0, 0);
    }
    // TAIL1: the last 4 coordinates c0 + i (c0 = 16 CTF) on 4x4x1 blocks.  Block (kq, pair group): A row i = T[kq + 4 s][c0 + i],
    // B column = logit(kq + 4 s, pair) = lg[s >> 2][s & 3] of THIS lane, k step s.  n4 / r4 of lane (kq, pair) are partial
    // over the NP preferences {kq + 4 s}; the sum over kq comes with the s reduction (linear) or a reduce-scatter (distance).
    v4 n4 = (v4){0.f, 0.f, 0.f, 0.f}, r4 = n4, q4 = n4;
    if (TAIL1) {
      const float* tn4 = CnS + kq * TPITCH + 16 * CTF + (lane & 3);
      const float* tr4 = ArS + kq * TPITCH + 16 * CTF + (lane & 3);
      float tan[NP], tar[NP];
#pragma unroll
      for (int sidx = 0; sidx < NP; ++sidx) { tan[sidx] = tn4[4 * sidx * TPITCH]; tar[sidx] = tr4[4 * sidx * TPITCH]; }
#pragma unroll
      for (int sidx = 0; sidx < NP; ++sidx) {
        n4 = __builtin_amdgcn_mfma_f32_4x4x1f32(tan[sidx], lg[sidx >> 2][sidx & 3], n4, 0, 0, 0);
        r4 = __builtin_amdgcn_mfma_f32_4x4x1f32(tar[sidx], lg[sidx >> 2][sidx & 3], r4, 0, 0, 0);
      }
      q4 = xt[j * NCH + 4 * CTF];                                // q chunk of coordinates c0 .. c0 + 3 (same for the 4 kq lanes)
    }
    // ---- s = q . n   (lane (kq, j) owns coordinates 16 ct + 4 kq .. + 3 = q chunk 4 ct + kq of pair j)
    v4 sacc = TAIL1 ? q4 * n4 : (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < CTF; ++ct) {
      if (4 * ct + 3 < NCH) {
        sacc += xb[4 * ct] * accN[ct];
      } else if (4 * ct < NCH) {
        if (4 * ct + kq < NCH) sacc += xb[4 * ct] * accN[ct];
      }
    }
    const float sfull = allsum_kq((sacc[0] + sacc[1]) + (sacc[2] + sacc[3]));
    const v4 ms = (v4){-sfull, -sfull, -sfull, -sfull};
    // ---- stage 2b: r^T tiles and the distance
#pragma unroll
    for (int ct = 0; ct < CTF; ++ct) {
      float ta[NP];
#pragma unroll
      for (int m = 0; m < NP; ++m) ta[m] = tr0[(16 * (m >> 2) + 4 * (m & 3)) * TPITCH + 16 * ct];
      v4 qv = (v4){0.f, 0.f, 0.f, 0.f};
      if (4 * ct + 3 < NCH) {
        qv = xb[4 * ct];
      } else if (4 * ct < NCH) {
        if (4 * ct + kq < NCH) qv = xb[4 * ct];
      }
      v4 accR = qv;                                              // q + r: start the accumulator at q
#pragma unroll
      for (int m = 0; m < NP; ++m) accR = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[m], lg[m >> 2][m & 3], accR, 0, 0, 0);
      const v4 tv = __builtin_elementwise_fma(ms, accN[ct], accR);
      if (l1) dacc += __builtin_elementwise_abs(tv);
      else dacc = __builtin_elementwise_fma(tv, tv, dacc);
    }
    if (TAIL1) {                                                 // lane kq takes coordinate c0 + kq
      const float nf = scatter_kq(n4), rf = scatter_kq(r4);
      const float qe = kq == 0 ? q4[0] : kq == 1 ? q4[1] : kq == 2 ? q4[2] : q4[3];
      const float tv = fmaf(-sfull, nf, qe + rf);
      dacc[0] += l1 ? fabsf(tv) : tv * tv;
    }
    }
    const float score = allsum_kq((dacc[0] + dacc[1]) + (dacc[2] + dacc[3]));
    if (kq == 0 && j < rem) (a.score + row0)[j] = score;
    if (pre) { sid[lane] = nx_u; sid[16 + lane] = nx_i; sid[32 + lane] = nx_e; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- d = 256 (config 5's width), soft gate, large batches: FOUR waves share a 16-pair tile, 64 coordinates each ----------------
// The kernel above keeps a whole 16 x 256 tile per wave: 48 row loads of 16 bytes in flight per lane and the q tile in registers are
// 256 VGPRs + 42 AGPRs, its x tiles fill the LDS at four waves per CU -- ONE wave per SIMD, which gathers, then computes, then
// gathers: on HBM-resident tables 0.59 ms per 716,800 pairs = 0.47 of the HBM peak where the pure gather of the same 1 KB rows
// reaches 0.73-0.79 (profiles/r04_gather_footprint.txt).  Here a wave owns 64 coordinates of the tile (the split of
// ktup_score_pref_bwd_wide.hip): 12 row loads per lane, all of them for the NEXT tile and in flight under the current tile's matrix
// phases; 12 waves per CU (three tiles in flight, three waves per SIMD).  The price: the three contractions over d -- logits, s = q.n,
// the distance -- are per-wave partials summed across the four waves through LDS (three workgroup barriers per tile, partials
// always added in wave order).
template <int NP_, bool HASE_>
struct FwGeom {
  static constexpr int NP = NP_;                           // groups of four preferences: P <= 4 NP, NP in {4, 5}
  static constexpr bool HASE = HASE_;
  static constexpr int NCH = 64, NWC = 4, NCW = 16;        // float4 chunks per row; waves per tile; chunks per wave
  static constexpr int PITCHA4 = 4 * 16 + 1;               // the slot-ordered logit table of McGeom (odd float4 pitch)
  static constexpr int XP = NCW + 1;                       // odd float4 pitch of a wave's x / q tile
  static constexpr int TROW = 4 * NP, TPITCH = 16 * 16 + 16;
  static constexpr int A0_F4 = 16 * PITCHA4;               // preferences 0..15 (slot i = preference 4 (i & 3) + (i >> 2))
  static constexpr int A1_F4 = NP > 4 ? 4 * PITCHA4 : 0;   // preferences 16..19, row-major: their logits come from 4x4x1 blocks (a 16x16x4 tile
                                                           // would be fifteen sixteenths padding: 512 of a wave's 2,300 matrix clocks per tile)
  static constexpr int T_F = TROW * TPITCH;
  static constexpr size_t TABLE_BYTES = (size_t)(A0_F4 + A1_F4) * 16 + (size_t)2 * T_F * 4;
  static constexpr int NLG = NP > 4 ? 5 : 4;               // live logit registers per lane (the second tile's only live one is reg 0)
  static constexpr size_t WAVE_BYTES = (size_t)16 * XP * 16 + 48 * 4;
  static constexpr size_t GROUP_BYTES = NWC * WAVE_BYTES + (size_t)NWC * NLG * 64 * 4 + (size_t)2 * NWC * 16 * 4;
  static constexpr int TG_FIT = (int)((160 * 1024 - TABLE_BYTES) / GROUP_BYTES);
  static constexpr int TG = TG_FIT >= 3 ? 3 : TG_FIT;      // tile groups (of four waves) per workgroup: three -- with four (128 VGPRs per lane) the KTUP variants spill 13-24 registers
};

template <typename G, bool L1>
__global__ __launch_bounds__(768) void pref_fwd_wide_kernel(McArgs a) {
  constexpr int NP = G::NP, NCH = G::NCH, NWC = G::NWC, XP = G::XP, PITCHA4 = G::PITCHA4, TPITCH = G::TPITCH, NLG = G::NLG;
  constexpr bool HASE = G::HASE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* A0S = reinterpret_cast<v4*>(smem);
  v4* A1S = A0S + G::A0_F4;                                      // [4][PITCHA4]
  float* CnS = reinterpret_cast<float*>(A1S + G::A1_F4);         // [TROW][TPITCH]
  float* ArS = CnS + G::T_F;
  const int t = threadIdx.x, lane = t & 63, kq = lane >> 4, j = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wv >> 2, w = wv & 3;                           // tile group; coordinate slice [64 w, 64 w + 64)
  const int nthr = blockDim.x, ngrp = nthr >> 8;
  char* gbase = reinterpret_cast<char*>(ArS + G::T_F) + (size_t)grp * G::GROUP_BYTES;
  v4* xt = reinterpret_cast<v4*>(gbase + (size_t)w * G::WAVE_BYTES);   // [16 pairs][XP]: this wave's 16 chunks of x, later of q
  int32_t* sid = reinterpret_cast<int32_t*>(xt + 16 * XP);       // [3][16]: ids of the tile whose rows are fetched next
  float* red = reinterpret_cast<float*>(gbase + NWC * G::WAVE_BYTES);  // [NWC][NLG][64] logit partials
  float* sred = red + NWC * NLG * 64;                            // [NWC][16]  q . n partials
  float* dred = sred + NWC * 16;                                 // [NWC][16]  distance partials
  {
    const int P = a.P, dp = a.dp;
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
    for (int idx = t; idx < G::A0_F4; idx += nthr) {
      const int i = idx / PITCHA4, c = idx - i * PITCHA4;
      const int p = 4 * (i & 3) + (i >> 2);
      A0S[idx] = (p < P && c < NCH) ? *reinterpret_cast<const v4*>(a.Alog + p * dp + 4 * c) : zero;
    }
    for (int idx = t; idx < G::A1_F4; idx += nthr) {
      const int r = idx / PITCHA4, c = idx - r * PITCHA4;
      A1S[idx] = (16 + r < P && c < NCH) ? *reinterpret_cast<const v4*>(a.Alog + (16 + r) * dp + 4 * c) : zero;
    }
    v4* Cn4 = reinterpret_cast<v4*>(CnS);
    v4* Ar4 = reinterpret_cast<v4*>(ArS);
    for (int idx = t; idx < G::T_F / 4; idx += nthr) {
      const int p = idx / (TPITCH / 4), c = idx - p * (TPITCH / 4);
      const bool ok = p < P && c < NCH;
      Cn4[idx] = ok ? *reinterpret_cast<const v4*>(a.Cn + p * dp + 4 * c) : zero;
      Ar4[idx] = ok ? *reinterpret_cast<const v4*>(a.Ar + p * dp + 4 * c) : zero;
    }
  }
  __syncthreads();
  const uint32_t gc = 16u * (uint32_t)w + (uint32_t)j;           // this lane's chunk of every row it fetches (rows kq, kq + 4, ..)
  const v4* xb = xt + j * XP + kq;                               // MFMA operand view: chunk 4 g + kq of pair j
  const v4* tab0 = A0S + j * PITCHA4 + kq + 16 * w;
  // preferences 16..19 on v_mfma_f32_4x4x1 (16 blocks): block (quarter kq of this wave's coordinates, pair group) -- lane (kq, pair j)
  // supplies A = preference 16 + (j & 3) and B = pair j at chunks 4 kq .. 4 kq + 3 of the slice
  const v4* tab1 = A1S + (j & 3) * PITCHA4 + 4 * kq + 16 * w;
  const v4* xb4 = xt + j * XP + 4 * kq;
  const float* tn0 = CnS + kq * TPITCH + j + 64 * w;
  const float* tr0 = ArS + kq * TPITCH + j + 64 * w;
  const int64_t ntiles = (a.n + 15) / 16;
  const int64_t tstride = (int64_t)gridDim.x * ngrp;
  const int64_t tfirst = (int64_t)blockIdx.x * ngrp;             // < ntiles (grid sizing)
  const int niter = (int)((ntiles - tfirst + tstride - 1) / tstride);   // the same for every wave of the workgroup (barriers)
  int32_t nx_u = 0, nx_i = 0, nx_e = 0;
  auto fetch_ids = [&](int64_t tile) {
    nx_u = 0; nx_i = 0; nx_e = 0;
    if (lane < 16 && tile < ntiles) {
      const int64_t row = tile * 16 + lane;
      const bool ok = row < a.n;
      const int64_t uid = ok ? a.u_ids[row] : 0, iid = ok ? a.i_ids[row] : 0;
      nx_u = (int32_t)uid; nx_i = (int32_t)iid;
      nx_e = HASE ? a.item2ent[iid] : 0;
    }
  };
  v4 uu[4], vv[4], ee[4];
  auto issue_rows = [&]() {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int row = kq + 4 * jj;
      const uint32_t idu = (uint32_t)sid[row], idi = (uint32_t)sid[16 + row];
      const v4* pu = a.U + ((uint64_t)idu * a.ldu4 + gc);
      const v4* pv = a.I + ((uint64_t)idi * a.ldi4 + gc);
      uu[jj] = a.nt ? __builtin_nontemporal_load(pu) : *pu;
      vv[jj] = a.nt ? __builtin_nontemporal_load(pv) : *pv;
      if (HASE) {
        const uint32_t ide = (uint32_t)sid[32 + row];
        const v4* pe = a.E + ((uint64_t)ide * a.lde4 + gc);
        ee[jj] = a.nt ? __builtin_nontemporal_load(pe) : *pe;
      }
    }
  };
  // ---- prologue: ids and rows of the first tile, ids of the second
  fetch_ids(tfirst + grp);
  if (lane < 16) { sid[lane] = nx_u; sid[16 + lane] = nx_i; sid[32 + lane] = nx_e; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (tfirst + grp < ntiles) issue_rows();
  fetch_ids(tfirst + grp + tstride);
  for (int it = 0; it < niter; ++it) {
    const int64_t tile_id = tfirst + grp + (int64_t)it * tstride;
    const bool live = tile_id < ntiles;                          // wave-uniform
    v4 q[4];
    v4 lg0 = (v4){0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const v4 ve = HASE ? vv[jj] + ee[jj] : vv[jj];
        xt[(kq + 4 * jj) * XP + j] = uu[jj] + ve;
        q[jj] = uu[jj] + (-ve);
      }
    }
    if (lane < 16) { sid[lane] = nx_u; sid[16 + lane] = nx_i; sid[32 + lane] = nx_e; }   // (the rows of THIS tile were addressed before)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- the next tile's rows travel under this tile's phases, the ids of the tile after it too
    if (tile_id + tstride < ntiles) issue_rows();
    fetch_ids(tile_id + 2 * tstride);
    if (live) {
      // ---- stage 1: this wave's 64 coordinates of logits^T (layout of the kernel above: lg[reg] of lane (kq, pair j) = preference 4 reg + kq)
#pragma unroll
      for (int gk = 0; gk < 4; ++gk) {
        const v4 bv = xb[4 * gk];
        const v4 av0 = tab0[4 * gk];
#pragma unroll
        for (int c = 0; c < 4; ++c) lg0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[c], bv[c], lg0, 0, 0, 0);
      }
      float* rw = red + w * NLG * 64 + lane;
      rw[0] = lg0[0]; rw[64] = lg0[1]; rw[128] = lg0[2]; rw[192] = lg0[3];
      if (NP > 4) {
        v4 acc = (v4){0.f, 0.f, 0.f, 0.f}, acc1 = acc;               // two chains: a 4x4x1 result is not ready for the next issue slot
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const v4 av = tab1[kk], bv = xb4[kk];
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[0], bv[0], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[1], bv[1], acc1, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[2], bv[2], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[3], bv[3], acc1, 0, 0, 0);
        }
        acc += acc1;
        rw[256] = scatter_kq(acc);                                  // lane (kq, pair j): preference 16 + kq, summed over the four quarters
      }
      // ---- q overwrites x (this wave's own slice: its stage-1 reads are behind it)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) xt[(kq + 4 * jj) * XP + j] = q[jj];
    }
    __syncthreads();
    // the PREVIOUS tile's score: its distance partials were written before this barrier and are rewritten after the next one (a third
    // barrier per tile only for this store cost 2-3 % of the kernel)
    if (it > 0 && w == 0 && kq == 0) {
      const int64_t row0 = (tile_id - tstride) * 16;
      if (row0 + j < a.n) (a.score + row0)[j] = (dred[j] + dred[16 + j]) + (dred[32 + j] + dred[48 + j]);
    }
    v4 accN[4];
    float lgs[NLG];
    if (live) {
#pragma unroll
      for (int k = 0; k < NLG; ++k) {
        const float* r0 = red + k * 64 + lane;
        lgs[k] = (r0[0] + r0[NLG * 64]) + (r0[2 * NLG * 64] + r0[3 * NLG * 64]);
      }
      // ---- stage 2a: n^T for this wave's coordinates, s partial
      v4 sacc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        accN[ct] = (v4){0.f, 0.f, 0.f, 0.f};
        float ta[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) ta[m] = tn0[(16 * (m >> 2) + 4 * (m & 3)) * TPITCH + 16 * ct];
#pragma unroll
        for (int m = 0; m < NP; ++m) accN[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[m], lgs[m], accN[ct], 0, 0, 0);
        sacc += xb[4 * ct] * accN[ct];
      }
      const float sp = allsum_kq((sacc[0] + sacc[1]) + (sacc[2] + sacc[3]));
      if (kq == 0) sred[w * 16 + j] = sp;
    }
    __syncthreads();
    if (live) {
      const float sfull = (sred[j] + sred[16 + j]) + (sred[32 + j] + sred[48 + j]);
      const v4 ms = (v4){-sfull, -sfull, -sfull, -sfull};
      v4 dacc = (v4){0.f, 0.f, 0.f, 0.f};
      // ---- stage 2b: r^T tiles and the distance
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        float ta[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) ta[m] = tr0[(16 * (m >> 2) + 4 * (m & 3)) * TPITCH + 16 * ct];
        v4 accR = xb[4 * ct];                                     // q + r: start the accumulator at q
#pragma unroll
        for (int m = 0; m < NP; ++m) accR = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[m], lgs[m], accR, 0, 0, 0);
        const v4 tv = __builtin_elementwise_fma(ms, accN[ct], accR);
        if (L1) dacc += __builtin_elementwise_abs(tv);
        else dacc = __builtin_elementwise_fma(tv, tv, dacc);
      }
      const float dp_ = allsum_kq((dacc[0] + dacc[1]) + (dacc[2] + dacc[3]));
      if (kq == 0) dred[w * 16 + j] = dp_;
    }
  }
  __syncthreads();
  if (w == 0 && kq == 0) {                                       // the last tile's score
    const int64_t row0 = (tfirst + grp + (int64_t)(niter - 1) * tstride) * 16;
    if (row0 + j < a.n) (a.score + row0)[j] = (dred[j] + dred[16 + j]) + (dred[32 + j] + dred[48 + j]);
  }
}

template <typename G, bool L1>
int launch_fwd_wide_l(const McArgs& a, hipStream_t st, const char* name) {
  static_assert(G::TG >= 1, "LDS budget");
  const int64_t ntiles = (a.n + 15) / 16;
  int tg = ntiles >= 256 * 3 ? 3 : ntiles >= 256 * 2 ? 2 : 1;
  if (tg > G::TG) tg = G::TG;
  if (opt_fwd_wide() >= 2 && opt_fwd_wide() < tg) tg = opt_fwd_wide();      // (A/B: option fwd_wide = 2: two tile groups per workgroup)
  const size_t lds = G::TABLE_BYTES + (size_t)tg * G::GROUP_BYTES;
  (void)hipFuncSetAttribute((const void*)pref_fwd_wide_kernel<G, L1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(G::TABLE_BYTES + (size_t)G::TG * G::GROUP_BYTES));
  const int grid = grid_for((ntiles + tg - 1) / tg, 256);
  hipLaunchKernelGGL((pref_fwd_wide_kernel<G, L1>), dim3(grid), dim3(tg * 256), lds, st, a);
  return check_launch(name);
}

int launch_fwd_wide(const McArgs& a, int np, hipStream_t st, const char* name) {
#define KTUP_FW(NPV)                                                                                                  \
  {                                                                                                                   \
    if (a.E) return a.l1 ? launch_fwd_wide_l<FwGeom<NPV, true>, true>(a, st, name) : launch_fwd_wide_l<FwGeom<NPV, true>, false>(a, st, name);    \
    return a.l1 ? launch_fwd_wide_l<FwGeom<NPV, false>, true>(a, st, name) : launch_fwd_wide_l<FwGeom<NPV, false>, false>(a, st, name);           \
  }
  if (np <= 4) KTUP_FW(4)
  KTUP_FW(5)
#undef KTUP_FW
}

template <typename G, bool L1>
int launch_mc_l(const McArgs& a, hipStream_t st, const char* name) {
  static_assert(G::NW >= 2, "LDS budget");
  constexpr size_t lds = G::TABLE_BYTES + (size_t)G::NW * G::WAVE_BYTES;
  (void)hipFuncSetAttribute((const void*)pref_fwd_mc_kernel<G, L1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int64_t ntiles = (a.n + 15) / 16;
  // small batches (a B = 512 training step is 64 tiles): fewer waves per workgroup, so that the tiles spread over many CUs
  // (one tile per SIMD) instead of queueing four deep on the matrix pipe of four CUs
  int nw = (int)((ntiles + 255) / 256);
  nw = nw < 4 ? (G::NW < 4 ? G::NW : 4) : (nw > G::NW ? G::NW : nw);
  const int grid = grid_for((ntiles + nw - 1) / nw, 256);
  hipLaunchKernelGGL((pref_fwd_mc_kernel<G, L1>), dim3(grid), dim3(nw * 64), lds, st, a);
  return check_launch(name);
}

template <typename G>
int launch_mc(const McArgs& a, hipStream_t st, const char* name) {
  if constexpr (G::NW < 2) {
    return 1;                                   // tables + one wave's tiles exceed the LDS: the caller's generic kernel runs
  } else {
    return a.l1 ? launch_mc_l<G, true>(a, st, name) : launch_mc_l<G, false>(a, st, name);
  }
}

template <int NCH, int NP>
int launch_mc_e(const McArgs& a, hipStream_t st, const char* name) {
  if (a.gumbel != KTUP_GUMBEL_OFF) {
    if (a.E) return launch_mc<McGeom<NCH, NP, true, true>>(a, st, name);
    return launch_mc<McGeom<NCH, NP, false, true>>(a, st, name);
  }
  if (a.E) return launch_mc<McGeom<NCH, NP, true, false>>(a, st, name);
  return launch_mc<McGeom<NCH, NP, false, false>>(a, st, name);
}

template <int NCH>
int launch_mc_np(const McArgs& a, int np, hipStream_t st, const char* name) {
  if (np <= 2) return launch_mc_e<NCH, 2>(a, st, name);
  if (np <= 4) return launch_mc_e<NCH, 4>(a, st, name);
  if (np <= 5) return launch_mc_e<NCH, 5>(a, st, name);
  return launch_mc_e<NCH, 8>(a, st, name);
}

}  // namespace

// Returns KTUP_OK / an error, or 1 when (d, P) is not one of the instantiated geometries (the caller then runs the
// run-time-geometry kernel).
int pref_fwd_mc(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                const float* Alog, const float* Ar, const float* Cn, int dp, int n_pref, int d, const int64_t* u_ids,
                const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform, uint64_t seed, uint64_t offset,
                float* score, hipStream_t st, const char* name) {
  if (n_pref > 32 || (d != 64 && d != 100 && d != 128 && d != 256)) return 1;
  if ((ldu | ldi | lde) & 3) return 1;
  if ((ldu >> 2) > 0xffffffffll || (ldi >> 2) > 0xffffffffll || (lde >> 2) > 0xffffffffll) return 1;
  McArgs a;
  a.U = reinterpret_cast<const v4*>(U); a.I = reinterpret_cast<const v4*>(I); a.E = reinterpret_cast<const v4*>(E);
  a.ldu4 = (uint32_t)(ldu >> 2); a.ldi4 = (uint32_t)(ldi >> 2); a.lde4 = (uint32_t)(lde >> 2);
  a.item2ent = item2ent;
  a.Alog = Alog; a.Ar = Ar; a.Cn = Cn;
  a.dp = dp; a.P = n_pref; a.l1 = l1;
  a.gumbel = gumbel_mode; a.uniform = uniform; a.seed = seed; a.offset = offset;
  a.u_ids = u_ids; a.i_ids = i_ids; a.n = n; a.score = score; a.nt = opt_nt_gather();
  const int np = (n_pref + 3) / 4;
  if (d == 64) return launch_mc_np<16>(a, np, st, name);
  if (d == 100) return launch_mc_np<25>(a, np, st, name);
  if (d == 256) {
    // soft gate, P <= 20, more than a chip-load of tiles: the coordinate-split kernel (option fwd_wide); else 4 waves per CU, a tile each
    if (gumbel_mode == KTUP_GUMBEL_OFF && np <= 5 && n >= 16 * 256 && opt_fwd_wide()) return launch_fwd_wide(a, np, st, name);
    return launch_mc_np<64>(a, np, st, name);
  }
  return launch_mc_np<32>(a, np, st, name);
}

}  // namespace ktup
