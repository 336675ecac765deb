// K5 / K6 / K7 backward on the matrix cores, coordinate-sliced: four waves share a 16-pair tile.  Used for d = 256 (BASELINE
// config 5, where one wave cannot hold a tile), for SMALL batches at d = 64 / 100 / 128 (a B = 512 step is 64 tiles: with one
// wave per tile its whole dependent chain -- 450 MFMAs, five LDS round trips, 300 atomic instructions -- runs on one SIMD while
// 97 % of the chip idles; sliced four ways the chain is ~4x shorter), and as the fused training-step kernel (STEP, below).
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
// LDS at d = 256, P = 20: 65.5 KB tables + 8 KB reduction scratch + 4 x 19 KB wave-private tiles = 152 KB, one workgroup per CU
// (d = 100: 87 KB).  MFMA work per tile is unchanged but split four ways.  A wave owns CTW coordinate tiles of 16 = NCW = 4 CTW
// chunks; for d = 100 (25 chunks, CTW = 2) the last wave holds one real chunk: tables and tiles are zero beyond d, so the padded
// coordinates contribute exact zeros and are simply never written back.
//
// STEP (ktup_train_rec_step): the rec half of a training step in this one launch -- see ktup_train_step.hip.  A tile holds 8
// (u, pos) pairs in slots 0-7 and the 8 (u, neg) pairs of the same examples in slots 8-15; the score is the forward value the
// backward recomputes anyway, the BPR term of example k = f(score[j] - score[j ^ 8]) is formed in registers, and its gradient
// feeds phases B-D.  Raw preference tables are mixed while staging (ktup_pref_prepare), gA / gC go straight to the four raw
// tables' gradients, one extra workgroup computes orthogonalLoss(pref, pref_norm).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ktup_common.h"
#include "ktup_lane_swap.h"
#include "ktup_pref_geom.h"

namespace ktup {
namespace {

template <int NCH_, int CTW_, int NP_, bool HASE_, bool HARD_, int NWC_ = 4>
struct WGeom {
  static constexpr int NP = NP_;
  static constexpr bool HASE = HASE_, HARD = HARD_;
  static constexpr int NCH = NCH_, D = 4 * NCH, NWC = NWC_, CTW = CTW_, NCW = 4 * CTW;   // NWC waves x CTW coordinate tiles of 16 = NCW chunks each
  static constexpr int NT = 64 * NWC;                     // threads of a workgroup
  static constexpr int NCHP = NWC * NCW;                  // padded chunk count (>= NCH; tables / tiles are zero beyond NCH)
  static_assert(NCHP >= NCH && NCHP - NCH < NCW, "the last wave must own at least one real chunk");
  static constexpr int PT = (NP + 3) / 4, TROW = 16 * PT;
  static constexpr int ROWS = 4 * NP + 1;                 // preferences 0 .. 4 NP - 1 (rows >= P zero) + the zero row
  static constexpr int RP4 = NCHP + 1, RPF = 4 * RP4;     // table row pitch (odd float4 count: conflict-free b128 reads)
  static constexpr int TAB_F4 = ROWS * RP4;
  static constexpr int TP4 = NCW + 1, TPF = 4 * TP4;      // wave tile row pitch
  static constexpr int TILE_F4 = 16 * TP4;
  static constexpr int GJ = CTW, GR = 64 / NCW;           // gather: GJ passes of GR rows x NCW chunks
  static constexpr int LT_F = TROW * 17;
  static constexpr int NOISE_F = HARD ? 16 * TROW : 0;
  static constexpr int RED_F = NWC * 64 * PT * 4;         // cross-wave partials of lg / gl
  // LT / GLT (beta w and gL / 2, preference major) hold the SAME numbers in every wave -- all waves carry the full logits after the
  // cross-wave sum -- so the workgroup keeps one copy: every wave writes all of it (equal values) before it reads it
  // (`red` twice: tiles alternate between the copies, which is what lets a tile end without a workgroup barrier; 4 x NWC x 16 floats of
  //  per-pair partials: reds | redav | redsc of the three-barrier path = the float4 per (wave, pair) of the merged one)
  static constexpr int REDC = NWC_ <= 4 ? 2 : 1;          // (the opt-in eight-wave form at d = 256 has no LDS for the second copy: it keeps its end-of-tile barrier)
  static constexpr size_t SHARED_BYTES = (size_t)3 * TAB_F4 * 16 + (size_t)REDC * RED_F * 4 + 4 * NWC * 16 * 4 + (size_t)2 * LT_F * 4;
  static constexpr size_t WAVE_BYTES = ((size_t)3 * TILE_F4 * 16 + 2 * 3 * 16 * 4 + (size_t)NOISE_F * 4 + 15) & ~(size_t)15;
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
  float *GU, *GV;                // ROWOUT: per-pair row gradients (n x D)
  // STEP (fused training step; ktup_train_rec_step): see BArgs in ktup_score_pref_bwd_mc.hip for the same fields
  const float *pref, *pnorm, *rel, *norm;   // RAW preference-side tables (rel / norm null for TUP), pitch ldp
  int64_t ldp, B;                // example k: pos = (u_ids[k], i_ids[k]), neg = (u_ids[k + B], i_ids[k + B])
  float target, gscale;
  float* loss;                   // loss[0] += mean_k -logsigmoid(target (pos_k - neg_k)); loss[1] += orthogonalLoss(pref, pnorm)
  float *gP, *gPn, *gR, *gRn;    // gradients of the raw tables, pitch D
  int orth;
  double* sumsq; int sumsq_slots; // STEP + ROWOUT (may be null): += sum over the stored rows of |row|^2 x (entries that read the row): GU rows
                                 // once, GV rows once for the item and once more for its entity -- the squared norm of the step's row
                                 // gradients if no two entries shared a table row (ktup_shard_reduce_norm's dup_only walk corrects the rest)
  const int64_t* neg_ids;        // STEP + ROWOUT (may be null): u_ids / i_ids / neg_ids are the id COLUMNS (n_batches x B each) and the kernel
  const int64_t* cursor; int64_t n_batches;   // reads batch (*cursor mod n_batches) itself -- no entry list has to exist before it starts
  double* gnorm;                 // STEP, gradients by atomics (may be null): the gradient-norm workspace of ktup_common.h -- every add of this
                                 // launch tracks the squared norm of the buffers it builds
  int u_once;                    // STEP + ROWOUT: u_ids holds B ids (example k's user, shared by its two pairs) and GU B rows (the sum)
  int noflush;                   // measurement knob (option dbg_noflush)
  // STEP + ROWOUT (may be null): the small tables' gradients go to n_rep REPLICAS -- grep[replica][A | C][P][D], workgroup b adds to replica
  // b mod n_rep -- instead of straight to gP / gPn (/ gR / gRn), and the step's next launch folds the replicas into those
  // (ktup_shard_reduce_norm_fold / _store_fold).  Every tile workgroup adding its 2 P D sums to ONE copy is 256 float atomics per address,
  // all issued as the workgroups end together: 10 us of a 117 us config-5 step (dbg_noflush 1), twice that when pref and rel have
  // gradients of their own; into 8, 16, 32 or 256 copies the same adds cost 4 (measured, all four alike).
  float* grep; int n_rep;
  int gumbel;
  const float* uniform;
  uint64_t seed, offset;
};

KTUP_DEV float wstep_neg_logsigmoid(float x) { return fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))); }   // as ktup_loss.hip / torch
KTUP_DEV float wstep_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

template <typename G, bool ROWOUT, bool STEP>
__global__ __launch_bounds__(G::NT) void pref_bwd_wide_kernel(WArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  constexpr int NP = G::NP, D = G::D, NCH = G::NCH, NCW = G::NCW, CTW = G::CTW, PT = G::PT, TROW = G::TROW;
  constexpr int RP4 = G::RP4, RPF = G::RPF, TP4 = G::TP4, TPF = G::TPF;
  constexpr bool HASE = G::HASE, HARD = G::HARD;
  constexpr bool RAGGED = G::NCHP != NCH;                     // d = 100: the last wave's slice runs past the row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* AlogT = reinterpret_cast<v4*>(smem);                    // [ROWS][RP4]
  v4* ArT = AlogT + G::TAB_F4;
  v4* CnT = ArT + G::TAB_F4;
  float* red0 = reinterpret_cast<float*>(CnT + G::TAB_F4);    // [2 copies][4 waves][64 lanes][PT * 4]
  float* reds = red0 + G::REDC * G::RED_F;                    // [4][16]
  float* redav = reds + G::NWC * 16;                          // [4][16]
  float* redsc = redav + G::NWC * 16;                         // [4][16]  STEP: partial scores
  const float* Alog2 = reinterpret_cast<const float*>(AlogT);
  const float* Ar2 = reinterpret_cast<const float*>(ArT);
  const float* Cn2 = reinterpret_cast<const float*>(CnT);
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // this wave's coordinate slice: [64 w, 64 w + 64)
  float* LT = reds + 4 * G::NWC * 16;                         // [TROW][17]  beta * w   (preference major; one copy per workgroup)
  float* GLT = LT + G::LT_F;                                  // [TROW][17]  gL / 2
  char* wbase = reinterpret_cast<char*>(GLT + G::LT_F) + (size_t)w * G::WAVE_BYTES;
  v4* XT = reinterpret_cast<v4*>(wbase);                      // x    [16 pairs][TP4]   (this wave's 16 chunks)
  v4* QT = XT + G::TILE_F4;                                   // q, later gn (a lane overwrites exactly what it read)
  v4* GRT = QT + G::TILE_F4;                                  // gr = gz
  int32_t* sid0 = reinterpret_cast<int32_t*>(GRT + G::TILE_F4);   // [2][3][16]: the ids of this tile and of the next one
  float* noise = reinterpret_cast<float*>(sid0 + 96);         // HARD: [16][TROW]
  const int P = a.P;
  auto wsum16 = [](const float* r, int jj) {                  // the waves' partials of slot jj, always summed in wave order
    float t = (r[jj] + r[16 + jj]) + (r[32 + jj] + r[48 + jj]);
#pragma unroll
    for (int ww = 4; ww < G::NWC; ww += 4) t += (r[16 * ww + jj] + r[16 * ww + 16 + jj]) + (r[16 * ww + 32 + jj] + r[16 * ww + 48 + jj]);
    return t;
  };
  int nblk = gridDim.x;                                       // workgroups that walk tiles
  if constexpr (STEP) {
    if (a.orth) {
      nblk = nblk - 1;
      if ((int)blockIdx.x == nblk) {
        // the extra workgroup: orthogonalLoss(pref, pref_norm) = sum_p (pn_p . p_p)^2 / |p_p|^2 (utils/loss.py:18-19), value and
        // gradient, rows dealt to the four waves; it touches no tile and leaves before any barrier
        float lo = 0.f, sq = 0.f;
        const bool track_o = !ROWOUT && a.gnorm != nullptr;
        const int set_o = track_o ? gnorm_set(a.gnorm) : 0;
        for (int p = w; p < P; p += G::NWC) {
          float dot = 0.f, nr = 0.f;
          for (int c = lane; c < NCH; c += 64) {
            const v4 r4 = *reinterpret_cast<const v4*>(a.pref + (int64_t)p * a.ldp + 4 * c);
            const v4 w4 = *reinterpret_cast<const v4*>(a.pnorm + (int64_t)p * a.ldp + 4 * c);
            const v4 dv = r4 * w4, nv = r4 * r4;
            dot += (dv[0] + dv[1]) + (dv[2] + dv[3]); nr += (nv[0] + nv[1]) + (nv[2] + nv[3]);
          }
          dot = group_sum<64>(dot); nr = group_sum<64>(nr);
          const float c1 = a.gscale * 2.f * dot / nr, c2 = a.gscale * 2.f * dot * dot / (nr * nr);
          for (int c = lane; c < NCH; c += 64) {
            const v4 r4 = *reinterpret_cast<const v4*>(a.pref + (int64_t)p * a.ldp + 4 * c);
            const v4 w4 = *reinterpret_cast<const v4*>(a.pnorm + (int64_t)p * a.ldp + 4 * c);
            const v4 gr4 = c1 * w4 - c2 * r4, gw4 = c1 * r4;
            if (track_o) {
              const float4 g0 = make_float4(gr4[0], gr4[1], gr4[2], gr4[3]), g1 = make_float4(gw4[0], gw4[1], gw4[2], gw4[3]);
              const float4 o0 = atomic_add4_old(a.gP + (int64_t)p * D + 4 * c, g0), o1 = atomic_add4_old(a.gPn + (int64_t)p * D + 4 * c, g1);
              sq += sq_gain4(o0, g0) + sq_gain4(o1, g1);
            } else {
              atomic_add4(a.gP + (int64_t)p * D + 4 * c, make_float4(gr4[0], gr4[1], gr4[2], gr4[3]));
              atomic_add4(a.gPn + (int64_t)p * D + 4 * c, make_float4(gw4[0], gw4[1], gw4[2], gw4[3]));
            }
          }
          lo += dot * dot / nr;
        }
        // the waves' shares of the value summed in wave order by one thread (one atomic per wave landed in any order: the value's last bit
        // changed from run to run); no barrier was passed yet: every wave is still here
        sq = track_o ? group_sum<64>(sq) : 0.f;
        if (lane == 0) { reds[w] = sq; reds[G::NWC + w] = lo; }
        __syncthreads();
        if (tid == 0) {
          double t = 0.0;
          float lsum = 0.f;
          for (int ww = 0; ww < G::NWC; ++ww) { t += (double)reds[ww]; lsum += reds[G::NWC + ww]; }
          if (lsum != 0.f) atomicAdd(a.loss + 1, lsum);
          if (track_o) gnorm_add(a.gnorm, set_o, t);
        }
        return;
      }
    }
  }
  const int grow_l = lane / NCW, gch = lane % NCW;            // gather: row grow_l + GR jj, chunk gch of the wave's slice
  const bool gok = !RAGGED || NCW * w + gch < NCH;            // chunks past the row stay zero in the tiles
  const int64_t ntiles = STEP ? (a.B + 7) / 8 : (a.n + 15) / 16;
  // ---- the gather runs ONE TILE AHEAD: while tile t goes through its phases (a chain of eight workgroup barriers on a wave that is
  // alone on its SIMD), the rows of the workgroup's next tile are in flight into registers, its ids already in the other id buffer
  constexpr int GJ = G::GJ, GR = G::GR;
  v4 uu[GJ], vv[GJ], ee[GJ];
  int64_t b0 = 0;                                               // id columns: the batch's first row
  if constexpr (STEP) { if (a.neg_ids) b0 = a.cursor ? ((*a.cursor) % a.n_batches) * a.B : 0; }
  int32_t nuid = 0, niid = 0, neid = 0;                         // lanes 0-15: the ids that are staged next (fetched a whole tile earlier)
  auto fetch_ids = [&](int64_t tile) {
    if (lane < 16) {
      const int64_t gr = STEP ? tile * 8 + (lane & 7) + (lane >> 3) * a.B : tile * 16 + lane;
      const bool ok = tile < ntiles && (STEP ? tile * 8 + (lane & 7) < a.B : gr < a.n);
      int64_t uid = 0, iid = 0;
      if (ok) {
        if (STEP && a.neg_ids) {
          const int64_t k = b0 + tile * 8 + (lane & 7);
          uid = a.u_ids[k]; iid = (lane >> 3) ? a.neg_ids[k] : a.i_ids[k];
        } else {
          uid = a.u_ids[(STEP && a.u_once) ? tile * 8 + (lane & 7) : gr]; iid = a.i_ids[gr];
        }
      }
      nuid = (int32_t)uid; niid = (int32_t)iid;
      neid = HASE ? a.item2ent[iid] : 0;
    }
  };
  auto stage_ids = [&](int32_t* dst) {
    if (lane < 16) { dst[lane] = nuid; dst[16 + lane] = niid; dst[32 + lane] = neid; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto load_rows = [&](const int32_t* src) {
#pragma unroll
    for (int jj = 0; jj < GJ; ++jj) {
      const int r = grow_l + GR * jj;
      const uint32_t idu = (uint32_t)src[r], idi = (uint32_t)src[16 + r];
      const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
      uu[jj] = gok ? a.U[(uint64_t)idu * a.ldu4 + (uint32_t)(NCW * w + gch)] : zero;
      vv[jj] = gok ? a.I[(uint64_t)idi * a.ldi4 + (uint32_t)(NCW * w + gch)] : zero;
      if (HASE) {
        const uint32_t ide = (uint32_t)src[32 + r];
        ee[jj] = gok ? a.E[(uint64_t)ide * a.lde4 + (uint32_t)(NCW * w + gch)] : zero;
      }
    }
  };
  // the first tile's ids (two dependent loads: the item id, then its entity) fly under the table staging
  if ((int64_t)blockIdx.x < ntiles) fetch_ids(blockIdx.x);
  // ---- stage the three tables: row p = preference p (zero beyond P), odd float4 row pitch.  All of a thread's loads are issued
  // before the first store: a B = 512 step is ONE tile per workgroup, so this prologue is on the step's critical path (with a
  // load -> store loop it cost ~8 dependent L2 round trips)
  {
    constexpr int NIT = (G::TAB_F4 + G::NT - 1) / G::NT;
    const int dp = a.dp;
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
    v4 t0[NIT], t1[NIT], t2[NIT], t3[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + G::NT * it;
      const int row = idx / RP4, c = idx - row * RP4;
      const bool ok = idx < G::TAB_F4 && row < P && c < NCH;
      t0[it] = t1[it] = t2[it] = t3[it] = zero;
      if constexpr (STEP) {      // the raw tables (mixed below like ktup_pref_prepare: Alog = (pref + rel) / 2, Ar = beta A, Cn = beta C)
        if (ok) {
          t0[it] = *reinterpret_cast<const v4*>(a.pref + (int64_t)row * a.ldp + 4 * c);
          t1[it] = *reinterpret_cast<const v4*>(a.pnorm + (int64_t)row * a.ldp + 4 * c);
          if (a.rel) {
            t2[it] = *reinterpret_cast<const v4*>(a.rel + (int64_t)row * a.ldp + 4 * c);
            t3[it] = *reinterpret_cast<const v4*>(a.norm + (int64_t)row * a.ldp + 4 * c);
          }
        }
      } else if (ok) {
        t0[it] = *reinterpret_cast<const v4*>(a.Alog + row * dp + 4 * c);
        t1[it] = *reinterpret_cast<const v4*>(a.Ar + row * dp + 4 * c);
        t2[it] = *reinterpret_cast<const v4*>(a.Cn + row * dp + 4 * c);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + G::NT * it;
      if (idx < G::TAB_F4) {
        if constexpr (STEP) {
          const v4 A = t0[it] + t2[it], C = t1[it] + t3[it];
          AlogT[idx] = 0.5f * A;
          ArT[idx] = a.beta * A;
          CnT[idx] = a.beta * C;
        } else {
          AlogT[idx] = t0[it]; ArT[idx] = t1[it]; CnT[idx] = t2[it];
        }
      }
    }
  }
  if ((int64_t)blockIdx.x < ntiles) {
    stage_ids(sid0);
    load_rows(sid0);
    fetch_ids((int64_t)blockIdx.x + nblk);
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
  float lpart = 0.f;                                          // STEP: this lane's share of the BPR loss value
  float ssq = 0.f;                                            // STEP + ROWOUT: this lane's share of the stored rows' squared norms
  constexpr bool TRK = STEP && !ROWOUT && NCH <= 32;          // (d = 256 has no registers left for the returned values: pref_step_mc refuses)
  const bool track = TRK && a.gnorm != nullptr;               // STEP, atomics: ... of the squared norm of the buffers the adds build
  const int gset = track ? gnorm_set(a.gnorm) : 0;
  const bool l1 = a.l1 != 0;
  const float beta = a.beta;
  v4 accA[PT][CTW], accC[PT][CTW];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) { accA[pt][ct] = (v4){0.f, 0.f, 0.f, 0.f}; accC[pt][ct] = accA[pt][ct]; }
  int cur = 0;
  for (int64_t tile_id = blockIdx.x; tile_id < ntiles; tile_id += nblk, cur ^= 1) {
    const int64_t row0 = tile_id * 16;
    // STEP: slot j holds example k = 8 tile + (j & 7); slots 8-15 are the negatives (rows k + B of the id arrays)
    const int64_t kpair = STEP ? tile_id * 8 + (j & 7) : row0 + j;
    const bool live_j = STEP ? kpair < a.B : kpair < a.n;
    const int64_t row_j = STEP ? kpair + (j >> 3) * a.B : kpair;          // row in the [pos ; neg] id / draw order
    const int32_t* sid = sid0 + 48 * cur;
    float* red = red0 + (G::REDC > 1 ? cur : 0) * G::RED_F;               // this tile's copy of the cross-wave scratch
    // ---- this wave's coordinate slice of the 16 pairs (gathered during the previous tile): x and q tiles
#pragma unroll
    for (int jj = 0; jj < GJ; ++jj) {
      const int r = grow_l + GR * jj;
      const v4 ve = HASE ? vv[jj] + ee[jj] : vv[jj];
      XT[r * TP4 + gch] = uu[jj] + ve;
      QT[r * TP4 + gch] = uu[jj] + (-ve);
    }
    if (tile_id + nblk < ntiles) {                                        // uniform over the workgroup
      stage_ids(sid0 + 48 * (cur ^ 1));
      load_rows(sid0 + 48 * (cur ^ 1));
      fetch_ids(tile_id + 2 * (int64_t)nblk);
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
      const int64_t grow = STEP ? (live_j ? row_j : 0) : min(row0 + j, a.n - 1);
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
    // MERGED (squared L2, the fused step): s = q.n, the score |z|^2 of z = (q + r) - s n and av = gz.n = 2 g z.n are all functions of FOUR
    // sums over the coordinates -- q.n, |q + r|^2, (q + r).n, |n|^2:  |z|^2 = |q + r|^2 - 2 s (q + r).n + s^2 |n|^2,  z.n = (q + r).n - s |n|^2 --
    // so the three cross-wave reductions of the chain (s, then the score, then av: a store, a workgroup barrier and a read each) are one.
    // The L1 distance needs z itself before it can be summed and keeps the three-step chain.
    constexpr bool MERGED_OK = STEP;
    const bool merged = MERGED_OK && !l1;
    v4 p_q2 = (v4){0.f, 0.f, 0.f, 0.f}, p_qn = p_q2, p_n2 = p_q2;
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
      if (MERGED_OK && merged) {
        p_q2 = __builtin_elementwise_fma(zz[ct], zz[ct], p_q2);
        p_qn = __builtin_elementwise_fma(zz[ct], nn[ct], p_qn);
        p_n2 = __builtin_elementwise_fma(nn[ct], nn[ct], p_n2);
      }
    }
    float s, g, av;
    if (MERGED_OK && merged) {
      // lane (kq, j) ends up with component kq of pair j's four sums over this wave's coordinates; summed across the waves in wave order
      const v4 part = (v4){(sacc[0] + sacc[1]) + (sacc[2] + sacc[3]), (p_q2[0] + p_q2[1]) + (p_q2[2] + p_q2[3]),
                           (p_qn[0] + p_qn[1]) + (p_qn[2] + p_qn[3]), (p_n2[0] + p_n2[1]) + (p_n2[2] + p_n2[3])};
      reds[(w * 16 + j) * 4 + kq] = scatter_kq(part);
      __syncthreads();
      const v4* r4 = reinterpret_cast<const v4*>(reds);
      auto wsum4 = [&](int jj) {
        v4 t = (r4[jj] + r4[16 + jj]) + (r4[32 + jj] + r4[48 + jj]);
#pragma unroll
        for (int ww = 4; ww < G::NWC; ww += 4) t += (r4[16 * ww + jj] + r4[16 * ww + 16 + jj]) + (r4[16 * ww + 32 + jj] + r4[16 * ww + 48 + jj]);
        return t;
      };
      const v4 tj = wsum4(j), to = wsum4(j ^ 8);
      s = tj[0];
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) zz[ct] = zz[ct] - s * nn[ct];    // z
      const float score = fmaf(s, fmaf(s, tj[3], -2.f * tj[2]), tj[1]);
      const float other = fmaf(to[0], fmaf(to[0], to[3], -2.f * to[2]), to[1]);
      const bool negh = (j >> 3) != 0;
      const float diff = negh ? other - score : score - other;   // pos - neg
      const float g0 = a.gscale * (1.f / (float)a.B);
      const float gd = -g0 * a.target * wstep_sigmoid(-a.target * diff);    // d/dpos of mean_k -logsigmoid(target diff_k)
      g = live_j ? (negh ? -gd : gd) : 0.f;
      if (w == 0 && live_j && !negh && kq == 0) lpart += wstep_neg_logsigmoid(a.target * diff);
      av = 2.f * g * fmaf(-s, tj[3], tj[2]);                     // gz . n with gz = 2 g z
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) zz[ct] = (2.f * g) * zz[ct];
    } else {
    s = allsum_kq((sacc[0] + sacc[1]) + (sacc[2] + sacc[3]));
    if (kq == 0) reds[w * 16 + j] = s;
    __syncthreads();
    s = wsum16(reds, j);
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) zz[ct] = zz[ct] - s * nn[ct];    // z
    if constexpr (STEP) {
      // z -> score (this IS the forward): partial over this wave's coordinates, summed across the waves, then the BPR term of
      // example k from the two halves of the tile (padded coordinates hold exact zeros)
      v4 dacc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) {
        if (l1) dacc += __builtin_elementwise_abs(zz[ct]);
        else dacc = __builtin_elementwise_fma(zz[ct], zz[ct], dacc);
      }
      float score = allsum_kq((dacc[0] + dacc[1]) + (dacc[2] + dacc[3]));
      if (kq == 0) redsc[w * 16 + j] = score;
      __syncthreads();
      score = wsum16(redsc, j);
      const float other = wsum16(redsc, j ^ 8);
      const bool negh = (j >> 3) != 0;
      const float diff = negh ? other - score : score - other;   // pos - neg
      const float g0 = a.gscale * (1.f / (float)a.B);
      const float gd = -g0 * a.target * wstep_sigmoid(-a.target * diff);    // d/dpos of mean_k -logsigmoid(target diff_k)
      g = live_j ? (negh ? -gd : gd) : 0.f;
      if (w == 0 && live_j && !negh && kq == 0) lpart += wstep_neg_logsigmoid(a.target * diff);
    } else {
      g = live_j ? a.gscore[row0 + j] : 0.f;
    }
    v4 aacc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) {
      const v4 z = zz[ct];
      v4 gz;
#pragma unroll
      for (int c = 0; c < 4; ++c) gz[c] = g * ddist1(z[c], l1);
      zz[ct] = gz;
      aacc += gz * nn[ct];
    }
    av = allsum_kq((aacc[0] + aacc[1]) + (aacc[2] + aacc[3]));
    if (kq == 0) redav[w * 16 + j] = av;
    __syncthreads();
    av = wsum16(redav, j);
    }
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
    float4 tu[TRK ? CTW : 1], tv[TRK ? CTW : 1], ou[TRK ? CTW : 1], oi[TRK ? CTW : 1], oe[TRK ? CTW : 1];   // tracked norm: values added / found
    bool has_e = false;
    // ---- C: gx^T = Alog2^T . gL^T of this wave's coordinates, then the row gradients
    {
      const int64_t gr = STEP ? row_j : row0 + j;                // ROWOUT row: the pair's place in the [pos ; neg] order
      const bool live = live_j;
      const int32_t ur = sid[j], ir = sid[16 + j], er = sid[32 + j];
      float* pu = a.gU + (int64_t)ur * a.ldu4 * 4;
      float* pi = a.gI + (int64_t)ir * a.ldi4 * 4;
      float* pe = (HASE && er != a.ent_pad) ? a.gE + (int64_t)er * a.lde4 * 4 : nullptr;
#pragma unroll
      for (int ct = 0; ct < (TRK ? CTW : 0); ++ct) { tu[ct] = f4zero(); tv[ct] = f4zero(); ou[ct] = f4zero(); oi[ct] = f4zero(); oe[ct] = f4zero(); }
      has_e = pe != nullptr;
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) {
        v4 gx = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NP; ++m)
          gx = __builtin_amdgcn_mfma_f32_16x16x4f32(Alog2[rb[m] + 16 * ct], gl[m >> 2][m & 3], gx, 0, 0, 0);
        const int c0 = 4 * NCW * w + 16 * ct + 4 * kq;
        v4 gu = gq[ct] + gx;
        const v4 gv = gx - gq[ct];
        bool u_mine = true;
        if constexpr (STEP) {
          // slots j and j ^ 8 hold a positive and its negative -- the same user in a BPR batch: their user-row gradients are summed
          // through one DPP rotate (row_ror:8, every lane of the wave active here) and slot j < 8 issues ONE atomic for both instead
          // of two on the same address (a third of the step's row atomics)
          const int32_t uo = __builtin_amdgcn_update_dpp(ur, ur, 0x128, 0xf, 0xf, false);
          v4 go;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int x = __float_as_int(gu[c]);
            go[c] = __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false));
          }
          if (ROWOUT ? a.u_once != 0 : uo == ur) {
            u_mine = j < 8;
            gu = gu + go;
          }
        }
        if (live && (!RAGGED || c0 < D)) {
          if constexpr (ROWOUT) {
            if (u_mine) *reinterpret_cast<v4*>(a.GU + ((STEP && a.u_once) ? kpair : gr) * D + c0) = gu;
            *reinterpret_cast<v4*>(a.GV + gr * D + c0) = gv;
            if constexpr (STEP) {
              const v4 g2 = gu * gu, v2 = gv * gv;
              const float su = (g2[0] + g2[1]) + (g2[2] + g2[3]), sv = (v2[0] + v2[1]) + (v2[2] + v2[3]);
              ssq += (u_mine ? su : 0.f) + ((HASE && er != a.ent_pad) ? 2.f * sv : sv);
            }
          } else {
            if (TRK && track) {          // the returned values wait in registers until the tile's table-gradient MFMAs are issued
              const float4 fu = make_float4(gu[0], gu[1], gu[2], gu[3]), fv = make_float4(gv[0], gv[1], gv[2], gv[3]);
              if (u_mine) { tu[ct] = fu; ou[ct] = atomic_add4_old(pu + c0, fu); }
              tv[ct] = fv;
              oi[ct] = atomic_add4_old(pi + c0, fv);
              if (pe) oe[ct] = atomic_add4_old(pe + c0, fv);
            } else {
              if (u_mine) atomic_add4(pu + c0, make_float4(gu[0], gu[1], gu[2], gu[3]));
              atomic_add4(pi + c0, make_float4(gv[0], gv[1], gv[2], gv[3]));
              if (pe) atomic_add4(pe + c0, make_float4(gv[0], gv[1], gv[2], gv[3]));
            }
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
    if (TRK && track) {
#pragma unroll
      for (int ct = 0; ct < (TRK ? CTW : 0); ++ct) ssq += sq_gain4(ou[ct], tu[ct]) + sq_gain4(oi[ct], tv[ct]) + (has_e ? sq_gain4(oe[ct], tv[ct]) : 0.f);
    }
    // (with two copies of `red` no workgroup barrier is needed here: the next tile's cross-wave scratch is the OTHER copy, the wave tiles are
    //  private, and every other shared array is written after the next tile's first barrier and read before this tile's last one)
    if constexpr (G::REDC == 1) __syncthreads();
  }
  // ---- flush the table gradients of this wave's coordinates
  if (TRK && track && !a.noflush) {          // tracked norm: all adds issued, then the returned values folded in (ktup_common.h)
    float fo[PT][CTW][4][4];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int p = 16 * pt + 4 * kq + reg, c = 4 * NCW * w + 16 * ct + j;
          const bool in = p < P && (!RAGGED || c < D);
          const float va = accA[pt][ct][reg], vc = accC[pt][ct][reg];
          fo[pt][ct][reg][0] = (in && va != 0.f) ? atomicAdd(a.gP + (int64_t)p * D + c, va) : 0.f;
          fo[pt][ct][reg][1] = (in && va != 0.f && a.gR) ? atomicAdd(a.gR + (int64_t)p * D + c, va) : -0.5f * va;     // (gain 0)
          fo[pt][ct][reg][2] = (in && vc != 0.f) ? atomicAdd(a.gPn + (int64_t)p * D + c, vc) : 0.f;
          fo[pt][ct][reg][3] = (in && vc != 0.f && a.gRn) ? atomicAdd(a.gRn + (int64_t)p * D + c, vc) : -0.5f * vc;
        }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int p = 16 * pt + 4 * kq + reg, c = 4 * NCW * w + 16 * ct + j;
          if (p < P && (!RAGGED || c < D)) {
            const float va = accA[pt][ct][reg], vc = accC[pt][ct][reg];
            ssq += (sq_gain(fo[pt][ct][reg][0], va) + sq_gain(fo[pt][ct][reg][1], va)) + (sq_gain(fo[pt][ct][reg][2], vc) + sq_gain(fo[pt][ct][reg][3], vc));
          }
        }
  } else if (STEP && ROWOUT && a.grep && !a.noflush) {
    float* mine = a.grep + (int64_t)((int)blockIdx.x % a.n_rep) * 2 * P * D;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int p = 16 * pt + 4 * kq + reg, c = 4 * NCW * w + 16 * ct + j;
          if (p < P && (!RAGGED || c < D)) {
            const float va = accA[pt][ct][reg], vc = accC[pt][ct][reg];
            if (va != 0.f) atomicAdd(mine + p * D + c, va);
            if (vc != 0.f) atomicAdd(mine + (P + p) * D + c, vc);
          }
        }
  } else if (!a.noflush)
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int p = 16 * pt + 4 * kq + reg, c = 4 * NCW * w + 16 * ct + j;
        if (p < P && (!RAGGED || c < D)) {
          const float va = accA[pt][ct][reg], vc = accC[pt][ct][reg];
          if constexpr (STEP) {       // A = pref + rel and C = pref_norm + norm: the mixed-table gradient goes to both summands
            if (va != 0.f) { atomicAdd(a.gP + (int64_t)p * D + c, va); if (a.gR) atomicAdd(a.gR + (int64_t)p * D + c, va); }
            if (vc != 0.f) { atomicAdd(a.gPn + (int64_t)p * D + c, vc); if (a.gRn) atomicAdd(a.gRn + (int64_t)p * D + c, vc); }
          } else {
            if (va != 0.f) atomicAdd(a.gA + (int64_t)p * D + c, va);
            if (vc != 0.f) atomicAdd(a.gC + (int64_t)p * D + c, vc);
          }
        }
      }
  if constexpr (STEP) {
    lpart = group_sum<64>(lpart);
    if (lane == 0 && lpart != 0.f) atomicAdd(a.loss, lpart * (1.f / (float)a.B));
    if constexpr (!ROWOUT) {
      if (track) {
        ssq = group_sum<64>(ssq);
        __syncthreads();
        if (lane == 0) reds[w] = ssq;
        __syncthreads();
        if (tid == 0) {
          double t = 0.0;
          for (int ww = 0; ww < G::NWC; ++ww) t += (double)reds[ww];
          gnorm_add(a.gnorm, gset, t);
        }
      }
    }
    if constexpr (ROWOUT) {
      if (a.sumsq) {            // ONE double atomic per workgroup (one per wave cost 11 us: ~2000 of them queue on 16 addresses)
        ssq = group_sum<64>(ssq);
        __syncthreads();
        if (lane == 0) reds[w] = ssq;
        __syncthreads();
        if (tid == 0) {
          double t = 0.0;
          for (int ww = 0; ww < G::NWC; ++ww) t += (double)reds[ww];
          if (t != 0.0) atomicAdd(a.sumsq + (a.sumsq_slots > 1 ? blockIdx.x % a.sumsq_slots : 0), t);
        }
      }
    }
  }
}

template <typename G, bool ROWOUT, bool STEP>
int launch_r(const WArgs& a, hipStream_t st, const char* name) {
  static_assert(G::LDS <= 160 * 1024, "LDS budget");
  (void)hipFuncSetAttribute((const void*)pref_bwd_wide_kernel<G, ROWOUT, STEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
  const int64_t ntiles = STEP ? (a.B + 7) / 8 : (a.n + 15) / 16;
  // d = 256: one workgroup (4 waves) per CU is all the LDS allows; narrower tables: two or three fit
  const int per_cu = (int)((160 * 1024) / G::LDS) < 1 ? 1 : (int)((160 * 1024) / G::LDS);
  // option `deterministic` (the fused step with gradients by atomics): ONE workgroup walks every tile -- a wave owns its coordinates of every
  // row, so each gradient cell receives its adds from one wave in program order, and the table gradients are flushed once
  const int tile_wgs = (STEP && !ROWOUT && opt_deterministic()) ? 1 : grid_for(ntiles, 256 * per_cu);
  const int grid = tile_wgs + ((STEP && a.orth) ? 1 : 0);
  WArgs b = a;
  if (!(STEP && ROWOUT)) b.grep = nullptr;
  hipLaunchKernelGGL((pref_bwd_wide_kernel<G, ROWOUT, STEP>), dim3(grid), dim3(G::NT), G::LDS, st, b);
  return check_launch(name);
}

template <typename G>
int launch(const WArgs& a, hipStream_t st, const char* name) {
  if (a.loss) return a.GU ? launch_r<G, true, true>(a, st, name) : launch_r<G, false, true>(a, st, name);
  return a.GU ? launch_r<G, true, false>(a, st, name) : launch_r<G, false, false>(a, st, name);
}

template <int NCH, int CTW, int NP, int NWC = 4>
int launch_e(const WArgs& a, hipStream_t st, const char* name) {
  if (a.gumbel != KTUP_GUMBEL_OFF) {
    if (a.E) return launch<WGeom<NCH, CTW, NP, true, true, NWC>>(a, st, name);
    return launch<WGeom<NCH, CTW, NP, false, true, NWC>>(a, st, name);
  }
  if (a.E) return launch<WGeom<NCH, CTW, NP, true, false, NWC>>(a, st, name);
  return launch<WGeom<NCH, CTW, NP, false, false, NWC>>(a, st, name);
}

// P <= 20 everywhere (NP in {4, 5}); the narrower widths also take P <= 32 (NP = 8), which d = 256 has no LDS for
int launch_d(const WArgs& a, int d, int np, hipStream_t st, const char* name) {
  if (d == 256) {                    // eight waves x 32 coordinates: two waves per SIMD (four x 64 left one wave alone with its latencies)
    if (opt_wide_waves() == 4) {
      if (np <= 4) return launch_e<64, 4, 4>(a, st, name);
      if (np <= 5) return launch_e<64, 4, 5>(a, st, name);
      return 1;
    }
    if (np <= 4) return launch_e<64, 2, 4, 8>(a, st, name);
    if (np <= 5) return launch_e<64, 2, 5, 8>(a, st, name);
    return 1;
  }
  if (d == 64) {
    if (np <= 4) return launch_e<16, 1, 4>(a, st, name);
    if (np <= 5) return launch_e<16, 1, 5>(a, st, name);
    return launch_e<16, 1, 8>(a, st, name);
  }
  if (d == 100) {
    if (np <= 4) return launch_e<25, 2, 4>(a, st, name);
    if (np <= 5) return launch_e<25, 2, 5>(a, st, name);
    return launch_e<25, 2, 8>(a, st, name);
  }
  if (d == 128) {
    if (np <= 4) return launch_e<32, 2, 4>(a, st, name);
    if (np <= 5) return launch_e<32, 2, 5>(a, st, name);
    return launch_e<32, 2, 8>(a, st, name);
  }
  return 1;
}

}  // namespace

// Returns KTUP_OK / an error, or 1 when the shape is not covered (the caller runs another kernel).
int pref_bwd_mc_wide(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                     int64_t ent_pad, const float* Alog, const float* Ar, const float* Cn, int dp, float beta, int n_pref, int d,
                     const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                     uint64_t offset, const float* gscore, float* gU, float* gI, float* gE, float* gA, float* gC, hipStream_t st,
                     const char* name, float* GU, float* GV) {
  if (n_pref > 32) return 1;
  if ((ldu | ldi | lde) & 3) return 1;
  if ((ldu >> 2) > 0xffffffffll || (ldi >> 2) > 0xffffffffll || (lde >> 2) > 0xffffffffll) return 1;
  WArgs a{};
  a.U = reinterpret_cast<const v4*>(U); a.I = reinterpret_cast<const v4*>(I); a.E = reinterpret_cast<const v4*>(E);
  a.ldu4 = (uint32_t)(ldu >> 2); a.ldi4 = (uint32_t)(ldi >> 2); a.lde4 = (uint32_t)(lde >> 2);
  a.item2ent = item2ent;
  a.Alog = Alog; a.Ar = Ar; a.Cn = Cn; a.dp = dp; a.P = n_pref; a.l1 = l1; a.beta = beta;
  a.u_ids = u_ids; a.i_ids = i_ids; a.n = n; a.ent_pad = ent_pad;
  a.gscore = gscore; a.gU = gU; a.gI = gI; a.gE = gE; a.gA = gA; a.gC = gC;
  a.GU = GU; a.GV = GV;
  a.gumbel = gumbel_mode; a.uniform = uniform; a.seed = seed; a.offset = offset;
  return launch_d(a, d, (n_pref + 3) / 4, st, name);
}

// The rec half of a B = 512 training step in one launch (STEP kernels above; see ktup_train_step.hip).  Returns 1 for shapes
// without a fused kernel (the caller keeps its multi-launch route).
int pref_step_mc(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde, const int32_t* item2ent,
                 int64_t ent_pad, const float* pref, const float* pnorm, const float* rel, const float* norm, int64_t ldp, int n_pref,
                 int d, const int64_t* u_ids, const int64_t* i_ids, int64_t B, int l1, int gumbel_mode, const float* uniform,
                 uint64_t seed, uint64_t offset, float target, float gscale, int orth, float* loss, float* gU, float* gI, float* gE,
                 float* gP, float* gPn, float* gR, float* gRn, hipStream_t st,
                 const char* name, float* GU, float* GV, double* sumsq, int sumsq_slots, const int64_t* neg_ids, const int64_t* cursor,
                 int64_t n_batches, double* gnorm, void* small_ws, size_t small_ws_bytes) {
  if (n_pref > 32 || (d == 256 && n_pref > 20)) return 1;
  if ((ldu | ldi | lde | ldp) & 3) return 1;
  if ((ldu >> 2) > 0xffffffffll || (ldi >> 2) > 0xffffffffll || (lde >> 2) > 0xffffffffll) return 1;
  WArgs a{};
  a.U = reinterpret_cast<const v4*>(U); a.I = reinterpret_cast<const v4*>(I); a.E = reinterpret_cast<const v4*>(E);
  a.ldu4 = (uint32_t)(ldu >> 2); a.ldi4 = (uint32_t)(ldi >> 2); a.lde4 = (uint32_t)(lde >> 2);
  a.item2ent = item2ent;
  a.P = n_pref; a.l1 = l1; a.beta = rel ? 0.5f : 1.0f;
  a.u_ids = u_ids; a.i_ids = i_ids; a.n = 2 * B; a.ent_pad = ent_pad;
  a.gU = gU; a.gI = gI; a.gE = gE;
  a.gumbel = gumbel_mode; a.uniform = uniform; a.seed = seed; a.offset = offset;
  a.pref = pref; a.pnorm = pnorm; a.rel = rel; a.norm = norm; a.ldp = ldp; a.B = B;
  a.target = target; a.gscale = gscale; a.loss = loss; a.gP = gP; a.gPn = gPn; a.gR = gR; a.gRn = gRn; a.orth = orth;
  a.noflush = opt_dbg_noflush();
  a.GU = GU; a.GV = GV;          // both set: the row gradients leave as rows of GU (example k: both pairs) / GV (pair k) instead of atomics
  a.u_once = GU != nullptr;
  if (gnorm && (GU || d > 128)) return set_error(KTUP_ERR_UNSUPPORTED, "%s: the tracked gradient norm exists for d <= 128, gradients by atomics", name);
  a.gnorm = gnorm;
  a.sumsq = sumsq; a.sumsq_slots = sumsq_slots;
  a.neg_ids = neg_ids; a.cursor = cursor; a.n_batches = n_batches > 0 ? n_batches : 1;
  if (small_ws && GU && small_ws_bytes >= pref_step_small_ws_bytes(B, n_pref, d) && (reinterpret_cast<uintptr_t>(small_ws) & 15u) == 0) {
    a.grep = reinterpret_cast<float*>(small_ws); a.n_rep = KTUP_TRAIN_SMALL_REPLICAS;
  }
  return launch_d(a, d, (n_pref + 3) / 4, st, name);
}

// bytes of the small-gradient replicas of a rows step (0: the shape has no fused kernel)
size_t pref_step_small_ws_bytes(int64_t B, int n_pref, int d) {
  if (B <= 0 || n_pref <= 0 || !(d == 64 || d == 100 || d == 128 || d == 256)) return 0;
  return (size_t)KTUP_TRAIN_SMALL_REPLICAS * 2 * n_pref * d * sizeof(float);
}

}  // namespace ktup
