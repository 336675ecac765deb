// K5/K6/K7: TUP / KTUP preference-gated translation score, forward and backward.
//   reference: jTransUP/models/transUP.py:69-82,105-170 ; jTransUP/models/jTransUP.py:122-143,250-315
//
// Per scored pair (u, i):   x = u + v,  q = u - v   with v = I[i]  (TUP)  or  I[i] + E[item2ent[i]]  (KTUP)
//   l  = (A x) / 2                        A = pref (+ rel)            (P logits)
//   w  = l                                 soft  (raw logits are the mixture weights)
//      = onehot(argmax(l + gumbel))        hard  (ST-Gumbel; backward through softmax(l + g))
//   r  = beta A^T w ,  n = beta C^T w      C = pref_norm (+ norm);  beta = 1 (TUP), 1/2 (KTUP)
//   s  = q . n ,  z = q + r - s n ,  score = sum |z|  or  sum z^2
// (the reference projects u and v on n separately and subtracts; same value up to fp32 rounding.)
//
// MI355X mapping.  A workgroup of NW waves owns a tile of 64 pairs.
//   gather : all NW*64 lanes stream the tile's U / I / E rows as one linear run of 16-B chunks (coalesced,
//            CH independent 16-B loads per table per lane in flight), x = u + v goes to LDS;
//   stage 1: lane = pair, wave = group of 5 preferences: logits as FMAs of the pair's x (ds_read_b128)
//            against table values that are WAVE-UNIFORM -> they come through the scalar cache (s_load) and
//            occupy SGPRs, not VGPRs or LDS bandwidth;
//   stage 2: lane = pair, wave = slice of the d coordinates (chunk c = wave + NW*j): r and n as FMAs of the
//            pair's logits against SGPR table values; q re-written to the same LDS tile;
//   s and the final sum are combined across the NW waves through two tiny LDS arrays.
// No cross-lane shuffles at all in the P x d contractions; LDS holds one 64 x d tile (25.6 KB at d=100).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ktup_pref_geom.h"

using namespace ktup;

namespace {

KTUP_DEV float mixed(const float* __restrict__ a, const float* __restrict__ b, int64_t ld, int row, int k, float scale) {
  return scale * (a[row * ld + k] + (b ? b[row * ld + k] : 0.f));
}

__global__ void pref_prepare_kernel(const float* __restrict__ pref, const float* __restrict__ pnorm,
                                    const float* __restrict__ rel, const float* __restrict__ norm, int64_t ld, int P,
                                    int d, int dp, int ppad, PrefGeom2 g2, int64_t off2, float* __restrict__ ws) {
  const float beta = rel ? 0.5f : 1.0f;
  const int total1 = (ppad + 2 * P) * dp;
  const int nA2 = g2.ppad2 * g2.dpa, nAC2 = P * g2.NW * g2.ev * 16;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total1 + nA2 + nAC2; idx += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (idx < total1) {
      int row = idx / dp;
      const int k = idx - row * dp;
      if (row < ppad) {  // logit table: (pref + rel) / 2  (transUP.py:108, jTransUP.py:253); the /2 is exact
        if (row < P && k < d) v = mixed(pref, rel, ld, row, k, 0.5f);
      } else if (row < ppad + P) {
        if (k < d) v = mixed(pref, rel, ld, row - ppad, k, beta);
      } else {
        if (k < d) v = mixed(pnorm, norm, ld, row - ppad - P, k, beta);
      }
      ws[idx] = v;
    } else if (idx < total1 + nA2) {
      const int i2 = idx - total1, row = i2 / g2.dpa, k = i2 - row * g2.dpa;
      if (row < P && k < d) v = mixed(pref, rel, ld, row, k, 0.5f);
      ws[off2 + i2] = v;
    } else {
      const int i3 = idx - total1 - nA2;
      const int entry = i3 / (g2.ev * 16), within = i3 - entry * (g2.ev * 16);
      const int p = entry / g2.NW, w = entry - p * g2.NW;
      const int q4 = within / 4, e = within & 3;            // float4 index inside the entry, element
      if (q4 < 2 * g2.CH) {
        const int j = q4 < g2.CH ? q4 : q4 - g2.CH;
        const int k = 4 * (w + g2.NW * j) + e;
        if (k < d) v = q4 < g2.CH ? mixed(pref, rel, ld, p, k, beta) : mixed(pnorm, norm, ld, p, k, beta);
      }
      ws[off2 + nA2 + i3] = v;
    }
  }
}

struct PrefArgs {
  const float4 *U, *I, *E;      // tables as float4 (E may be null)
  int64_t ldu4, ldi4, lde4;     // pitches in float4
  const int32_t* item2ent;      // null for TUP
  const float4 *Alog, *Ar, *Cn; // prepared tables, pitch dp4
  const float *Alog2, *AC2;     // 64-byte-vector layouts for pref_fwd2 (ktup_pref_geom.h)
  int dpa16, ppad2;             // Alog2 row pitch in 64-byte vectors, rows
  int P, ppad, lp, nch, dp4;  // lp = LDS pitch of the per-pair logit rows (odd: conflict-free b32)
  const int64_t *u_ids, *i_ids;
  int64_t n;
  int l1, gumbel;
  const float* uniform;
  uint64_t seed, offset;
  float* score;
  // backward only
  const float* gscore;
  float *gU, *gI, *gE, *gA, *gC;
  int64_t ent_pad;
  float alpha_beta;  // beta (alpha is always 1/2)
};

KTUP_DEV float draw_uniform(const PrefArgs& a, int64_t grow, int p) {
  if (a.gumbel == KTUP_GUMBEL_INPUT) return a.uniform[grow * a.P + p];
  const uint64_t idx = (uint64_t)grow * (uint64_t)a.P + (uint64_t)p + a.offset;
  const uint4 r = Philox(a.seed)(idx >> 2, 0x4b545550ull /* "KTUP" stream tag */);
  const uint32_t w = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
  return u01(w);
}

// ---------------------------------------------------------------------------------------------
// Shared front end: ids -> gather -> X tile in LDS -> logits in LDS.  Leaves uu/vv (this lane's CH
// chunks of u and v in the LINEAR gather mapping) in registers for the later q = u - v rewrite.
template <int CH, int NW>
struct Tile {
  static constexpr int NT = NW * 64;
  float4* tile;     // [TR * nch]
  float* logit;     // [TR * ppad]
  float* red_s;     // [NW * TR]
  float* red_z;     // [NW * TR]
  int64_t* s_u;     // [TR] row ids
  int64_t* s_i;
  int64_t* s_e;
  KTUP_DEV void carve(char* smem, int nch, int lp) {
    tile = reinterpret_cast<float4*>(smem);
    logit = reinterpret_cast<float*>(tile + TR * nch);
    red_s = logit + TR * lp;
    red_z = red_s + NW * TR;
    s_u = reinterpret_cast<int64_t*>(red_z + NW * TR);
    s_i = s_u + TR;
    s_e = s_i + TR;
  }
  __host__ __device__ static size_t bytes(int nch, int lp) {
    return (size_t)TR * nch * 16 + (size_t)TR * lp * 4 + 2 * (size_t)NW * TR * 4 + 3 * (size_t)TR * 8;
  }
};

template <int CH, int NW>
KTUP_DEV void tile_front(const PrefArgs& a, Tile<CH, NW>& T, int64_t row0, int t, int lane, int w, float4 (&uu)[CH],
                         float4 (&vv)[CH]) {
  constexpr int NT = NW * 64;
  const int nch = a.nch;
  // ---- phase 0: ids (tail rows alias row 0 of the batch's tables: valid memory, result discarded)
  if (t < TR) {
    const int64_t gr = row0 + t;
    const bool ok = gr < a.n;
    const int64_t uid = ok ? a.u_ids[gr] : 0, iid = ok ? a.i_ids[gr] : 0;
    T.s_u[t] = uid;
    T.s_i[t] = iid;
    T.s_e[t] = a.E ? (int64_t)a.item2ent[iid] : 0;
  }
  __syncthreads();
  // ---- phase 1: coalesced gather in the linear mapping v = t + NT*j  ->  (row = v / nch, chunk = v % nch)
  const int total = TR * nch;
  const int qstep = NT / nch, rstep = NT - qstep * nch;
  {
    int v = t, row = t / nch, c = t - (t / nch) * nch;
    float4 ee[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (v < total) {
        uu[j] = a.U[T.s_u[row] * a.ldu4 + c];
        vv[j] = a.I[T.s_i[row] * a.ldi4 + c];
        ee[j] = a.E ? a.E[T.s_e[row] * a.lde4 + c] : f4zero();
      } else {
        uu[j] = f4zero(); vv[j] = f4zero(); ee[j] = f4zero();
      }
      v += NT; row += qstep; c += rstep;
      if (c >= nch) { c -= nch; ++row; }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      vv[j] = vv[j] + ee[j];                       // ie = i + e   (jTransUP.py:132-135)
      const int vj = t + NT * j;
      if (vj < total) T.tile[vj] = uu[j] + vv[j];  // x = u + ie, LDS image is linear in v
    }
  }
  __syncthreads();
  // ---- stage 1: logits.  lane = pair, wave w = preferences [pbase, pbase + PB)
  const float4* xrow = T.tile + lane * nch;
  const sptr4 Alog = as_scalar(a.Alog);
  for (int pbase = w * PB; pbase < a.ppad; pbase += NW * PB) {
    float acc[PB];
#pragma unroll
    for (int pp = 0; pp < PB; ++pp) acc[pp] = 0.f;
    for (int c = 0; c < nch; ++c) {
      const float4 x = xrow[c];
#pragma unroll
      for (int pp = 0; pp < PB; ++pp) {
        const float4 av = sld(Alog, (pbase + pp) * a.dp4 + c);
        acc[pp] = fmaf(x.x, av.x, fmaf(x.y, av.y, fmaf(x.z, av.z, fmaf(x.w, av.w, acc[pp]))));
      }
    }
#pragma unroll
    for (int pp = 0; pp < PB; ++pp) T.logit[lane * a.lp + pbase + pp] = acc[pp];
    if (a.gumbel != KTUP_GUMBEL_OFF) {  // hard gate: store l + g; padding preferences never win the argmax
      const int64_t grow = min(row0 + lane, a.n - 1);
#pragma unroll 1
      for (int pp = 0; pp < PB; ++pp) {  // (same lane re-reads its own LDS words: no barrier needed)
        const int p = pbase + pp;
        float* slot = T.logit + lane * a.lp + p;
        *slot = p < a.P ? *slot + gumbel_from_uniform(draw_uniform(a, grow, p)) : -INFINITY;
      }
    }
  }
  __syncthreads();  // logits complete, every X read done -> the tile may be overwritten with q
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int vj = t + NT * j;
    if (vj < total) T.tile[vj] = uu[j] - vv[j];  // q = u - ie
  }
}

// first-max argmax over the row's P noisy logits (ties -> lowest index, like torch.max)
KTUP_DEV int row_argmax(const float* lrow, int P) {
  int best = 0;
  float bv = lrow[0];
  for (int p = 1; p < P; ++p) {
    const float v = lrow[p];
    if (v > bv) { bv = v; best = p; }
  }
  return best;
}

template <int CH, int NW>
__global__ __launch_bounds__(NW * 64) void pref_fwd_kernel(PrefArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Tile<CH, NW> T;
  T.carve(smem, a.nch, a.lp);
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nch = a.nch;
  const bool l1 = a.l1 != 0;
  const int64_t ntiles = (a.n + TR - 1) / TR;
  const sptr4 Ar = as_scalar(a.Ar), Cn = as_scalar(a.Cn);
  for (int64_t tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
    const int64_t row0 = tile_id * TR;
    float4 uu[CH], vv[CH];
    tile_front<CH, NW>(a, T, row0, t, lane, w, uu, vv);
    __syncthreads();  // q tile + logits visible
    // ---- stage 2: lane = pair, wave = coordinate slice (chunks w, w+NW, ...)
    float4 r[CH], nn[CH];
    const float* lrow = T.logit + lane * a.lp;
    if (a.gumbel == KTUP_GUMBEL_OFF) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { r[j] = f4zero(); nn[j] = f4zero(); }
      for (int p = 0; p < a.P; ++p) {
        const float wgt = lrow[p];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          r[j] = fma4(wgt, sld(Ar, p * a.dp4 + w + NW * j), r[j]);
          nn[j] = fma4(wgt, sld(Cn, p * a.dp4 + w + NW * j), nn[j]);
        }
      }
    } else {
      const int ps = row_argmax(lrow, a.P);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        r[j] = a.Ar[ps * a.dp4 + w + NW * j];
        nn[j] = a.Cn[ps * a.dp4 + w + NW * j];
      }
    }
    float4 q[CH];
    float sp = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = w + NW * j;
      q[j] = c < nch ? T.tile[lane * nch + c] : f4zero();
      sp += dot4(q[j], nn[j]);
    }
    T.red_s[w * TR + lane] = sp;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) s += T.red_s[k * TR + lane];
    float zp = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) zp += dist4(fma4(-s, nn[j], q[j] + r[j]), l1);
    T.red_z[w * TR + lane] = zp;
    __syncthreads();
    if (w == 0 && row0 + lane < a.n) {
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) tot += T.red_z[k * TR + lane];
      a.score[row0 + lane] = tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward.  With gz = gscore * f'(z), a = gz . n :
//   gq = gz - a n ;  gr = gz ;  gn = -s gz - a q
//   gw_p = beta (A_p . gr + C_p . gn)                     (per pair, per preference)
//   gl   = gw (soft)   or   y * (gw - sum_p y_p gw_p),  y = softmax(l + g)   (hard, ST estimator)
//   gx   = (1/2) A^T gl ;  gu = gq + gx ;  gv = -gq + gx  -> scattered to U / I (/ E, except the pad row)
//   gA_p += (1/2) gl_p x + beta w_p gr ;  gC_p += beta w_p gn       (mixed-table grads)
// Extra LDS vs forward: gw[TR][ppad], and two more 64 x d tiles (gr and gn) for the table-gradient pass,
// which maps lane -> coordinate k and loops the tile's 64 pairs; its accumulators live in registers across
// all tiles of the workgroup and are flushed with one atomic per (p, k) at the end.
// KS = column slices of the table-gradient pass: its x / gr / gn tiles hold nch / KS chunks per pair (d > 128 does not
// fit three whole extra tiles in 160 KB of LDS); slice sl = chunks [sl * SW, (sl + 1) * SW) = this wave's j in [sl * CH / KS, ...).
template <int CH, int NW, int KS>
__global__ __launch_bounds__(NW * 64) void pref_bwd_kernel(PrefArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64;
  Tile<CH, NW> T;
  T.carve(smem, a.nch, a.lp);
  const int nch = a.nch, d = nch * 4;
  char* extra = smem + Tile<CH, NW>::bytes(nch, a.lp);
  float* gw = reinterpret_cast<float*>(extra);                 // [TR * lp]  gw then gl
  float* wt = gw + TR * a.lp;                                  // [TR * lp]  mixture weights w
  float4* tgr = reinterpret_cast<float4*>(wt + TR * a.lp + (TR * a.lp & 3 ? 4 - (TR * a.lp & 3) : 0));  // 16-B aligned, [TR * SW] gr
  static_assert(CH % KS == 0, "slices are whole j ranges");
  const int SW = KS == 1 ? nch : NW * (CH / KS);               // chunks per slice: c = w + NW j, j in a slice's range
  float4* tgn = tgr + TR * SW;                                 // [TR * SW]   gn
  float4* tx = tgn + TR * SW;                                  // [TR * SW]   x (again)
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool l1 = a.l1 != 0;
  const float beta = a.alpha_beta;
  const int64_t ntiles = (a.n + TR - 1) / TR;
  const sptr4 Ar = as_scalar(a.Ar), Cn = as_scalar(a.Cn), Alog = as_scalar(a.Alog);

  // table-gradient accumulators: thread -> coordinate k = t % d_round, preference group t / d  (PG prefs each)
  constexpr int PGMAX = 16;
  const int kcol = t % d;                 // NT >= d is guaranteed by the geometry (NT=256 for d<=128, 512 for d<=256)
  const int ngrp = NT / d;                // preference groups that fit
  const int grp = t / d;
  const int pg = (a.P + ngrp - 1) / ngrp; // preferences per group (<= PGMAX checked on host)
  const bool tg_active = grp < ngrp;
  float accA[PGMAX], accC[PGMAX];
#pragma unroll
  for (int i = 0; i < PGMAX; ++i) { accA[i] = 0.f; accC[i] = 0.f; }

  for (int64_t tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
    const int64_t row0 = tile_id * TR;
    float4 uu[CH], vv[CH];
    tile_front<CH, NW>(a, T, row0, t, lane, w, uu, vv);
    // x again (tile_front overwrote the X tile with q), and zero gw
    {
      if (KS == 1) {
        const int total = TR * nch;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const int vj = t + NT * j;
          if (vj < total) tx[vj] = uu[j] + vv[j];
        }
      }
      for (int i = t; i < TR * a.lp; i += NT) gw[i] = 0.f;
    }
    __syncthreads();
    // ---- mixture weights
    const float* lrow = T.logit + lane * a.lp;
    float* wrow = wt + lane * a.lp;
    int ps = 0;
    if (a.gumbel != KTUP_GUMBEL_OFF) ps = row_argmax(lrow, a.P);
    if (w == 0) {
      if (a.gumbel == KTUP_GUMBEL_OFF) {
        for (int p = 0; p < a.P; ++p) wrow[p] = lrow[p];
      } else {
        for (int p = 0; p < a.P; ++p) wrow[p] = p == ps ? 1.f : 0.f;
      }
    }
    // ---- stage 2 (as forward)
    float4 r[CH], nn[CH];
    if (a.gumbel == KTUP_GUMBEL_OFF) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { r[j] = f4zero(); nn[j] = f4zero(); }
      for (int p = 0; p < a.P; ++p) {
        const float wgt = lrow[p];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          r[j] = fma4(wgt, sld(Ar, p * a.dp4 + w + NW * j), r[j]);
          nn[j] = fma4(wgt, sld(Cn, p * a.dp4 + w + NW * j), nn[j]);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        r[j] = a.Ar[ps * a.dp4 + w + NW * j];
        nn[j] = a.Cn[ps * a.dp4 + w + NW * j];
      }
    }
    float4 q[CH];
    float sp = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = w + NW * j;
      q[j] = c < nch ? T.tile[lane * nch + c] : f4zero();
      sp += dot4(q[j], nn[j]);
    }
    T.red_s[w * TR + lane] = sp;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) s += T.red_s[k * TR + lane];
    const float g = row0 + lane < a.n ? a.gscore[row0 + lane] : 0.f;  // tail pairs contribute nothing
    float4 gz[CH];
    float ap = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      gz[j] = g * ddist4(fma4(-s, nn[j], q[j] + r[j]), l1);
      ap += dot4(gz[j], nn[j]);
    }
    T.red_z[w * TR + lane] = ap;
    __syncthreads();
    float av = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) av += T.red_z[k * TR + lane];
    float4 gn[CH], gq[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      gq[j] = fma4(-av, nn[j], gz[j]);
      gn[j] = fma4(-av, q[j], (-s) * gz[j]);
      const int c = w + NW * j;
      if (KS == 1 && c < nch) { tgr[lane * nch + c] = gz[j]; tgn[lane * nch + c] = gn[j]; }
    }
    // ---- gw_p = Ar_p . gr + Cn_p . gn  (Ar, Cn already carry beta), partial over this wave's chunks
    for (int p = 0; p < a.P; ++p) {
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < CH; ++j)
        part += dot4(sld(Ar, p * a.dp4 + w + NW * j), gz[j]) + dot4(sld(Cn, p * a.dp4 + w + NW * j), gn[j]);
      atomicAdd(&gw[lane * a.lp + p], part);  // ds_add_f32 across the NW waves
    }
    __syncthreads();
    // ---- gl (in place in gw) for the hard gate: softmax Jacobian of y = softmax(l + g)
    if (a.gumbel != KTUP_GUMBEL_OFF && w == 0) {
      float* grow = gw + lane * a.lp;
      float m = -INFINITY;
      for (int p = 0; p < a.P; ++p) m = fmaxf(m, lrow[p]);
      float den = 0.f, dotyg = 0.f;
      for (int p = 0; p < a.P; ++p) { const float e = expf(lrow[p] - m); den += e; dotyg += e * grow[p]; }
      const float inv = 1.f / den;
      dotyg *= inv;
      for (int p = 0; p < a.P; ++p) { const float y = expf(lrow[p] - m) * inv; grow[p] = y * (grow[p] - dotyg); }
    }
    if (a.gumbel != KTUP_GUMBEL_OFF) __syncthreads();
    // ---- gx = Alog^T gl (Alog carries the 1/2), then scatter gu = gq + gx, gv = -gq + gx
    {
      const float* glrow = gw + lane * a.lp;
      float4 gx[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) gx[j] = f4zero();
      for (int p = 0; p < a.P; ++p) {
        const float glp = glrow[p];
#pragma unroll
        for (int j = 0; j < CH; ++j) gx[j] = fma4(glp, sld(Alog, p * a.dp4 + w + NW * j), gx[j]);
      }
      if (row0 + lane < a.n) {
        const int64_t ur = T.s_u[lane], ir = T.s_i[lane], er = T.s_e[lane];
        float* pu = a.gU + ur * a.ldu4 * 4;
        float* pi = a.gI + ir * a.ldi4 * 4;
        float* pe = (a.E && er != a.ent_pad) ? a.gE + er * a.lde4 * 4 : nullptr;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const int c = w + NW * j;
          if (c < nch) {
            const float4 gv = gx[j] - gq[j];
            atomic_add4(pu + 4 * c, gq[j] + gx[j]);
            atomic_add4(pi + 4 * c, gv);
            if (pe) atomic_add4(pe + 4 * c, gv);
          }
        }
      }
    }
    // ---- table gradients: lane -> coordinate, loop the tile's pairs (gl in gw, w in wt, x / gr / gn tiles), slice by slice
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) {
      if (KS > 1) {
        if (sl > 0) __syncthreads();                     // the previous slice has been consumed
        const int total = TR * nch;
#pragma unroll
        for (int j = 0; j < CH; ++j) {                   // x: linear gather mapping
          const int vj = t + NT * j;
          const int row = vj / nch, c = vj - row * nch;
          if (vj < total && c / SW == sl) tx[row * SW + c - sl * SW] = uu[j] + vv[j];
        }
#pragma unroll
        for (int j = sl * (CH / KS); j < (sl + 1) * (CH / KS); ++j) {   // gr / gn: this wave's chunks of the slice
          const int c = w + NW * j;
          if (c < nch) { tgr[lane * SW + c - sl * SW] = gz[j]; tgn[lane * SW + c - sl * SW] = gn[j]; }
        }
        __syncthreads();
      }
      const int k0 = sl * SW * 4;                        // first coordinate of the slice
      if (tg_active && kcol >= k0 && kcol < k0 + SW * 4) {
        const float* fx = reinterpret_cast<const float*>(tx);
        const float* fgr = reinterpret_cast<const float*>(tgr);
        const float* fgn = reinterpret_cast<const float*>(tgn);
        const int p0 = grp * pg, ds = SW * 4, kk = kcol - k0;
        const int64_t rows_here = min((int64_t)TR, a.n - row0);
        for (int rr = 0; rr < rows_here; ++rr) {
          const float xk = fx[rr * ds + kk], grk = fgr[rr * ds + kk], gnk = fgn[rr * ds + kk];
          const float* glr = gw + rr * a.lp + p0;
          const float* wr = wt + rr * a.lp + p0;
#pragma unroll
          for (int i = 0; i < PGMAX; ++i) {
            if (i < pg && p0 + i < a.P) {
              const float wv = wr[i];
              accA[i] = fmaf(0.5f * glr[i], xk, fmaf(beta * wv, grk, accA[i]));
              accC[i] = fmaf(beta * wv, gnk, accC[i]);
            }
          }
        }
      }
    }
    __syncthreads();  // tiles / gw / wt are rewritten by the next iteration
  }
  if (tg_active) {
    const int p0 = grp * pg;
#pragma unroll
    for (int i = 0; i < PGMAX; ++i) {
      if (i < pg && p0 + i < a.P) {
        atomicAdd(a.gA + (int64_t)(p0 + i) * d + kcol, accA[i]);
        atomicAdd(a.gC + (int64_t)(p0 + i) * d + kcol, accC[i]);
      }
    }
  }
}


// =============================================================================================================
// pref_fwd2: the tuned forward.  Same math and mapping as pref_fwd_kernel; what changes is how the wave-uniform table
// values reach the SGPRs and how much bookkeeping surrounds each FMA (rocprof on v1: 56 M SALU + 18 M SMEM
// wave-instructions next to 79 M VALU, waves parked in s_waitcnt 48 % of their cycles):
//   * tables are pre-laid-out by ktup_pref_prepare so that what a wave needs for one step is contiguous and aligned:
//     stage 1 uses one s_load_dwordx8 per preference per 2 chunks, stage 2 three s_load_dwordx16 per preference
//     (v1: one s_load_dwordx4 plus 4 SALU address ops per 4 FMAs);
//   * geometry fits d exactly (d=100: 5 waves x 5 chunks = 25 chunks, 20 preferences = 5 waves x 4): no padded FMAs;
//   * the gate mode is a template parameter (the soft kernel carries no Gumbel / Philox code or registers);
//   * the next tile's ids (and the dependent item->entity lookup) are prefetched under the current tile's compute;
//   * this file is built with -fno-slp-vectorize: SLP turns the scalar-operand FMAs into v_pk_fma_f32, which needs its
//     SGPR operands in aligned pairs and costs one s_mov per FMA to build them (v_pk_fma is not faster than 2 v_fmac).
typedef float v8f __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) v8f* sptr8;

template <int NW, int CH, bool HARD>
__global__ __launch_bounds__(NW * 64) void pref_fwd2_kernel(PrefArgs a) {
  KTUP_RESOLVE_GUMBEL(a);
  constexpr int NT = NW * 64, EV = (2 * CH + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nch = a.nch, lp = a.ppad2 | 1;
  float4* tile = reinterpret_cast<float4*>(smem);                       // [TR * nch] + 4 zero chunks of slack
  float* logit = reinterpret_cast<float*>(tile + TR * nch + 4);         // [TR * lp]
  float* red_s = logit + TR * lp;                                       // [NW * TR]
  float* red_z = red_s + NW * TR;                                       // [NW * TR]
  int32_t* sid = reinterpret_cast<int32_t*>(red_z + NW * TR);           // [2][3][TR] double-buffered row ids
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool l1 = a.l1 != 0;
  const int64_t ntiles = (a.n + TR - 1) / TR;
  const int total = TR * nch;
  const int qstep = NT / nch, rstep = NT - qstep * nch;
  const int nv2 = (nch + 1) / 2;  // stage-1 steps of 2 chunks (the table is zero padded past nch)

  if (t < 4) tile[total + t] = f4zero();
  if (t < TR) {  // ids of the first tile
    const int64_t gr = (int64_t)blockIdx.x * TR + t;
    const bool ok = gr < a.n;
    const int64_t uid = ok ? a.u_ids[gr] : 0, iid = ok ? a.i_ids[gr] : 0;
    sid[t] = (int32_t)uid;
    sid[TR + t] = (int32_t)iid;
    sid[2 * TR + t] = a.E ? a.item2ent[iid] : 0;
  }
  __syncthreads();

  int it = 0;
  for (int64_t tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x, ++it) {
    const int64_t row0 = tile_id * TR;
    const int32_t* cur = sid + (it & 1) * 3 * TR;
    int32_t* nxt = sid + ((it & 1) ^ 1) * 3 * TR;
    // ---- gather (linear mapping) -- row loads first, the id prefetch queues behind them
    float4 uu[CH], vv[CH];
    int32_t pre_u = 0, pre_i = 0, pre_e = 0;
    const bool pre = t < TR && tile_id + gridDim.x < ntiles;
    {
      float4 ee[CH];
      int v = t, row = t / nch, c = t - (t / nch) * nch;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (v < total) {
          uu[j] = a.U[(int64_t)cur[row] * a.ldu4 + c];
          vv[j] = a.I[(int64_t)cur[TR + row] * a.ldi4 + c];
          ee[j] = a.E ? a.E[(int64_t)cur[2 * TR + row] * a.lde4 + c] : f4zero();
        } else {
          uu[j] = f4zero(); vv[j] = f4zero(); ee[j] = f4zero();
        }
        v += NT; row += qstep; c += rstep;
        if (c >= nch) { c -= nch; ++row; }
      }
      const int64_t ngr = (tile_id + gridDim.x) * TR + t;
      int64_t nuid = 0, niid = 0;
      if (pre && ngr < a.n) { nuid = a.u_ids[ngr]; niid = a.i_ids[ngr]; }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        vv[j] = vv[j] + ee[j];
        const int vj = t + NT * j;
        if (vj < total) tile[vj] = uu[j] + vv[j];
      }
      pre_u = (int32_t)nuid; pre_i = (int32_t)niid;
      if (pre && a.E) pre_e = a.item2ent[niid];   // dependent lookup: consumed only after stage 1
    }
    __syncthreads();
    // ---- stage 1: lane = pair, wave = 4 preferences; per step 2 chunks of x against one 32-byte scalar load per preference
    {
      const float4* xrow = tile + lane * nch;
      for (int pbase = w * PB2; pbase < a.ppad2; pbase += NW * PB2) {
        const sptr8 A0 = (sptr8)(uintptr_t)(a.Alog2 + (size_t)pbase * a.dpa16 * 16);
        const int pitch8 = a.dpa16 * 2;
        float acc[PB2];
#pragma unroll
        for (int pp = 0; pp < PB2; ++pp) acc[pp] = 0.f;
        for (int v = 0; v < nv2; ++v) {
          v8f A[PB2];
#pragma unroll
          for (int pp = 0; pp < PB2; ++pp) A[pp] = A0[pp * pitch8 + v];
          const float4 x0 = xrow[2 * v], x1 = xrow[2 * v + 1];  // a chunk past the row end meets zero table entries
#pragma unroll
          for (int pp = 0; pp < PB2; ++pp) {
            acc[pp] = fmaf(x0.x, A[pp][0], acc[pp]); acc[pp] = fmaf(x0.y, A[pp][1], acc[pp]);
            acc[pp] = fmaf(x0.z, A[pp][2], acc[pp]); acc[pp] = fmaf(x0.w, A[pp][3], acc[pp]);
            acc[pp] = fmaf(x1.x, A[pp][4], acc[pp]); acc[pp] = fmaf(x1.y, A[pp][5], acc[pp]);
            acc[pp] = fmaf(x1.z, A[pp][6], acc[pp]); acc[pp] = fmaf(x1.w, A[pp][7], acc[pp]);
          }
        }
#pragma unroll
        for (int pp = 0; pp < PB2; ++pp) logit[lane * lp + pbase + pp] = acc[pp];
        if (HARD) {
          const int64_t grow = min(row0 + lane, a.n - 1);
#pragma unroll 1
          for (int pp = 0; pp < PB2; ++pp) {
            const int p = pbase + pp;
            float* slot = logit + lane * lp + p;
            *slot = p < a.P ? *slot + gumbel_from_uniform(draw_uniform(a, grow, p)) : -INFINITY;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int vj = t + NT * j;
      if (vj < total) tile[vj] = uu[j] - vv[j];  // q = u - ie
    }
    if (pre) { nxt[t] = pre_u; nxt[TR + t] = pre_i; nxt[2 * TR + t] = pre_e; }
    __syncthreads();
    // ---- stage 2: lane = pair, wave = chunks w, w + NW, ...
    float4 r[CH], nn[CH];
    if (!HARD) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { r[j] = f4zero(); nn[j] = f4zero(); }
      sptr16 ac = as_scalar16(a.AC2) + w * EV;
      const float* lrow = logit + lane * lp;
      for (int p = 0; p < a.P; ++p, ac += NW * EV) {
        v16f V[EV];
#pragma unroll
        for (int e = 0; e < EV; ++e) V[e] = ac[e];
        const float wg = lrow[p];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          constexpr int dummy = 0; (void)dummy;
          const int qa = j, qc = CH + j;
          r[j].x = fmaf(wg, V[qa / 4][(qa % 4) * 4 + 0], r[j].x); r[j].y = fmaf(wg, V[qa / 4][(qa % 4) * 4 + 1], r[j].y);
          r[j].z = fmaf(wg, V[qa / 4][(qa % 4) * 4 + 2], r[j].z); r[j].w = fmaf(wg, V[qa / 4][(qa % 4) * 4 + 3], r[j].w);
          nn[j].x = fmaf(wg, V[qc / 4][(qc % 4) * 4 + 0], nn[j].x); nn[j].y = fmaf(wg, V[qc / 4][(qc % 4) * 4 + 1], nn[j].y);
          nn[j].z = fmaf(wg, V[qc / 4][(qc % 4) * 4 + 2], nn[j].z); nn[j].w = fmaf(wg, V[qc / 4][(qc % 4) * 4 + 3], nn[j].w);
        }
      }
    } else {
      const int ps = row_argmax(logit + lane * lp, a.P);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int c = w + NW * j;
        r[j] = c < nch ? a.Ar[ps * a.dp4 + c] : f4zero();
        nn[j] = c < nch ? a.Cn[ps * a.dp4 + c] : f4zero();
      }
    }
    float4 q[CH];
    float sp = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = w + NW * j;
      q[j] = c < nch ? tile[lane * nch + c] : f4zero();
      sp += dot4(q[j], nn[j]);
    }
    red_s[w * TR + lane] = sp;
    __syncthreads();
    float sfull = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) sfull += red_s[k * TR + lane];
    float zp = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) zp += dist4(fma4(-sfull, nn[j], q[j] + r[j]), l1);
    red_z[w * TR + lane] = zp;
    __syncthreads();
    if (w == 0 && row0 + lane < a.n) {
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) tot += red_z[k * TR + lane];
      a.score[row0 + lane] = tot;
    }
  }
}

template <int NW, int CH>
int launch_pref2(const PrefArgs& a, hipStream_t st, const char* name) {
  const int lp = a.ppad2 | 1;
  const size_t lds = ((size_t)TR * a.nch + 4) * 16 + (size_t)TR * lp * 4 + 2 * (size_t)NW * TR * 4 + 2 * 3 * (size_t)TR * 4;
  if (lds > 160 * 1024) return set_error(KTUP_ERR_UNSUPPORTED, "%s: tile needs %zu B of LDS", name, lds);
  const bool hard = a.gumbel != KTUP_GUMBEL_OFF;
  const void* fn = hard ? (const void*)pref_fwd2_kernel<NW, CH, true> : (const void*)pref_fwd2_kernel<NW, CH, false>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int64_t ntiles = (a.n + TR - 1) / TR;
  // persistent grid = what is actually resident (a queued extra workgroup per CU would run its whole tile loop late)
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NW * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  if (const char* env = getenv("KTUP_PREF_WGS_PER_CU")) per_cu = atoi(env) > 0 ? atoi(env) : per_cu;
  const int grid = grid_for(ntiles, 256 * per_cu);
  if (hard) hipLaunchKernelGGL((pref_fwd2_kernel<NW, CH, true>), dim3(grid), dim3(NW * 64), lds, st, a);
  else hipLaunchKernelGGL((pref_fwd2_kernel<NW, CH, false>), dim3(grid), dim3(NW * 64), lds, st, a);
  return check_launch(name);
}

int dispatch_fwd2(const PrefArgs& a, int d, int n_pref, hipStream_t st, const char* name) {
  const PrefGeom2 g = pref_geom2(d, n_pref);
#define KTUP_F2(NW_, CH_) if (g.NW == NW_ && g.CH == CH_) return launch_pref2<NW_, CH_>(a, st, name);
  KTUP_F2(4, 4) KTUP_F2(5, 4) KTUP_F2(5, 5) KTUP_F2(8, 4) KTUP_F2(8, 5)
#undef KTUP_F2
  return launch_pref2<8, 8>(a, st, name);
}


// =============================================================================================================
// pref_fwd3: soft-gate forward with the two P x d contractions on the fp32-input matrix cores.
// tools/ubench (profiles/r01_ubench.txt): FMAs fed from SGPRs issue at <= 71 TF (37 TF at one wave per SIMD), the
// v_mfma_f32_32x32x2_f32 pipe sustains 150-158 TF at ANY occupancy; the KTUP gate needs 12 kflop per pair.
// Each wave owns 32 pairs from gather to score -- no workgroup barrier after the tables are staged in LDS -- and
// LANE = PAIR in every phase, because the products are computed TRANSPOSED (D = A.B with the table as A, the pairs as B):
//   gather : the wave's 32 rows of U / I / E as one linear run of 16-B chunks; x = u + v -> its LDS tile, q = u - v stays
//            in registers until stage 1 is done, then overwrites x in place;
//   stage 1: logits^T (P x 32 pairs) = Alog (P x d) . X^T.  k-pairing: for k-group g the lanes of half h read the float4
//            [8g + 4h .. +3] of their table row (A) / their pair's x row (B); component c of the float4 pair feeds MFMA c,
//            whose two k's are (8g + c, 8g + 4 + c): one ds_read_b128 per operand feeds 4 MFMAs, no lane selects.
//            Result layout: lane (h, pair) holds the logits of ITS pair for preferences p = (r&3) + 8(r>>2) + 4h;
//   stage 2: n^T (d x 32 pairs) = Cn^T . logits^T and r^T likewise, in 32-coordinate tiles.  Result register r of stage 1
//            IS the B operand of stage-2 MFMA r (k pair = preferences (r&3)+8(r>>2) and +4): no LDS round trip, no shuffle;
//   tail   : lane (h, pair) holds coordinates 32t + (r&3) + 8(r>>2) + 4h of its own pair, so s = q.n and sum f(q + r - s n)
//            are in-lane sums over registers (q read back as conflict-free ds_read_b128) plus ONE cross-half add each.
struct Fwd3Geom {
  int kg, pitchA4, prow, nr, nt, nw, pair, exp;
  unsigned long long* trace;   // debug (KTUP_PREF_TRACE=<file>): s_memtime stamps of workgroup 0, [wave][tile < 8][8 marks]
  size_t table_bytes, wave_bytes;
};
inline Fwd3Geom fwd3_geom(int d, int P) {
  Fwd3Geom g{};
  g.kg = (d + 7) / 8;
  g.pitchA4 = 2 * g.kg + 1;                 // odd number of float4 per logit-table row: conflict-free b128 reads
  g.prow = 8 * ((P + 7) / 8);               // stage-2 table rows (preferences padded to the MFMA k pairing), zero filled
  g.nr = g.prow / 2;                        // stage-2 MFMAs per 32-coordinate tile and table (= stage-1 registers used)
  g.nt = (d + 31) / 32;
  g.table_bytes = (size_t)(P + 1) * g.pitchA4 * 16 + (size_t)2 * g.prow * 128 * 4;
  g.wave_bytes = ((size_t)32 * (d / 4) + 1) * 16 + 3 * 32 * 4 + 16;
  g.wave_bytes = (g.wave_bytes + 15) & ~(size_t)15;
  const size_t budget = 160 * 1024 - g.table_bytes - 64;   // 64 B of pair flags
  g.nw = (int)(budget / g.wave_bytes);
  if (g.nw > 8) g.nw = 8;
  return g;
}

template <int J, int NT2, int NR>
__global__ __launch_bounds__(512) void pref_fwd3_kernel(PrefArgs a, Fwd3Geom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nch = a.nch, d = nch * 4, P = a.P;
  float4* AlogL = reinterpret_cast<float4*>(smem);                                    // [(P + 1)][pitchA4], row P = 0
  float* CnL = reinterpret_cast<float*>(AlogL + (P + 1) * g.pitchA4);                  // [prow][128]
  float* ArL = CnL + g.prow * 128;                                                     // [prow][128]
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5, j = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  char* wbase = reinterpret_cast<char*>(ArL + g.prow * 128) + (size_t)w * g.wave_bytes;
  float4* xt = reinterpret_cast<float4*>(wbase);                                       // [32 * nch] + 1 zero chunk
  int32_t* sid = reinterpret_cast<int32_t*>(xt + 32 * nch + 1);                        // [3][32]
  // matrix-phase token of each wave pair (w, w + nw/2): turn[pair] = side allowed in, done[w] = wave w has left its loop
  volatile int* turn = reinterpret_cast<volatile int*>(reinterpret_cast<char*>(ArL + g.prow * 128) + (size_t)g.nw * g.wave_bytes);
  volatile int* done = turn + 8;
  const int half_nw = g.nw / 2;
  const bool paired = g.pair != 0 && (g.nw & 1) == 0;
  const int pairi = paired ? w % half_nw : 0, side = paired ? w / half_nw : 0, partner = paired ? (w + half_nw) % g.nw : 0;
  // ---- stage the preference tables once per workgroup (zero padded)
  {
    const float* Alog = reinterpret_cast<const float*>(a.Alog);
    const float* Ar = reinterpret_cast<const float*>(a.Ar);
    const float* Cn = reinterpret_cast<const float*>(a.Cn);
    const int dp = a.dp4 * 4, nA = (P + 1) * g.pitchA4 * 4, nT = g.prow * 128;
    float* AlogLf = reinterpret_cast<float*>(AlogL);
    for (int idx = t; idx < nA; idx += blockDim.x) {
      const int p = idx / (g.pitchA4 * 4), k = idx - p * (g.pitchA4 * 4);
      AlogLf[idx] = (p < P && k < d) ? Alog[p * dp + k] : 0.f;
    }
    for (int idx = t; idx < nT; idx += blockDim.x) {
      const int p = idx >> 7, c = idx & 127;
      const bool ok = p < P && c < d;
      CnL[idx] = ok ? Cn[p * dp + c] : 0.f;
      ArL[idx] = ok ? Ar[p * dp + c] : 0.f;
    }
    if (lane == 0) xt[32 * nch] = f4zero();
    if (t < 16) turn[t] = 0;   // turn[0..7] = side 0 first, done[0..7] = 0
  }
  __syncthreads();
  // Waves w and w + nw/2 share a SIMD (a workgroup's waves are dealt to the 4 SIMDs cyclically) and run the same loop.
  // Left alone they march in lockstep -- every wave gathers at once (a ~12 TB/s burst on L2), then every wave wants the
  // matrix pipe at once -- so nothing overlaps (s_memtime trace).  A token per pair makes the two
  // waves alternate: one is in its matrix phases while the other gathers / finishes its tail.
  const bool l1 = a.l1 != 0;
  const int64_t ntiles = (a.n + 31) / 32;
  const int total = 32 * nch;
  const int qstep = 64 / nch, rstep = 64 - qstep * nch;
  const int peff = j < P ? j : P;
  const int64_t wstride = (int64_t)gridDim.x * g.nw;
  bool first = true;
  int it = -1;
  for (int64_t tile_id = (int64_t)blockIdx.x * g.nw + w; tile_id < ntiles; tile_id += wstride) {
    ++it;
    if (g.trace && blockIdx.x == 0 && lane == 0 && it < 8) g.trace[((size_t)w * 8 + it) * 8 + 0] = __builtin_amdgcn_s_memtime();
    const int64_t row0 = tile_id * 32;
    // ---- ids of the wave's 32 pairs: loaded for the first tile here, afterwards prefetched one tile ahead (below)
    if (first && lane < 32) {
      const int64_t gr = row0 + lane;
      const bool ok = gr < a.n;
      const int64_t uid = ok ? a.u_ids[gr] : 0, iid = ok ? a.i_ids[gr] : 0;
      sid[lane] = (int32_t)uid;
      sid[32 + lane] = (int32_t)iid;
      sid[64 + lane] = a.E ? a.item2ent[iid] : 0;
    }
    first = false;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- gather: x -> LDS tile, q kept in registers; all row loads of the tile in flight at once
    float4 q[J];
    {
      int v = lane, row = lane / nch, c = lane - (lane / nch) * nch;
      asm volatile("" : "+v"(v), "+v"(row), "+v"(c));  // opaque per tile: stops LICM from hoisting J x (row, chunk)
                                                       // address sets out of the persistent loop (they get spilled)
      float4 uu[J], vv[J], ee[J];
      int vs[J];
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        vs[jj] = v;
        if (v < total) {
          uu[jj] = a.U[(int64_t)sid[row] * a.ldu4 + c];
          vv[jj] = a.I[(int64_t)sid[32 + row] * a.ldi4 + c];
          ee[jj] = a.E ? a.E[(int64_t)sid[64 + row] * a.lde4 + c] : f4zero();
        } else {
          uu[jj] = f4zero(); vv[jj] = f4zero(); ee[jj] = f4zero();
        }
        v += 64; row += qstep; c += rstep;
        if (c >= nch) { c -= nch; ++row; }
      }
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        const float4 ve = vv[jj] + ee[jj];
        if (vs[jj] < total) xt[vs[jj]] = uu[jj] + ve;
        q[jj] = uu[jj] - ve;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (g.trace && blockIdx.x == 0 && lane == 0 && it < 8) g.trace[((size_t)w * 8 + it) * 8 + 1] = __builtin_amdgcn_s_memtime();
    // ---- next tile's ids (and the dependent item -> entity lookup) travel under the matrix phases of this tile
    int32_t nx_u = 0, nx_i = 0, nx_e = 0;
    const bool pre = lane < 32 && tile_id + wstride < ntiles;
    if (pre) {
      const int64_t gr = (tile_id + wstride) * 32 + lane;
      const bool ok = gr < a.n;
      const int64_t uid = ok ? a.u_ids[gr] : 0, iid = ok ? a.i_ids[gr] : 0;
      nx_u = (int32_t)uid; nx_i = (int32_t)iid;
      nx_e = a.E ? a.item2ent[iid] : 0;
    }
    if (paired) {  // acquire the pair's matrix-phase token (the partner may already have left its loop)
      while (turn[pairi] != side && done[partner] == 0) __builtin_amdgcn_s_sleep(2);
    }
    // ---- stage 1: logits^T = Alog . X^T   (A = table row p = lane & 31, B = x of pair lane & 31)
    v16f lg;
#pragma unroll
    for (int r = 0; r < 16; ++r) lg[r] = 0.f;
    {
      const float4* xb = xt + j * nch + h;
      const float4* ta = AlogL + peff * g.pitchA4 + h;
      float4 av = ta[0], bv = xb[0];
      for (int gk = 0; gk < g.kg; ++gk) {   // operands of group gk + 1 are fetched under the 4 MFMAs of group gk
        const int nx = gk + 1 < g.kg ? 2 * (gk + 1) : 0;
        const float4 an = ta[nx], bn = xb[nx];
        lg = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, lg, 0, 0, 0);
        lg = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, lg, 0, 0, 0);
        lg = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, lg, 0, 0, 0);
        lg = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, lg, 0, 0, 0);
        av = an; bv = bn;
      }
    }
    asm volatile("" :: "v"(lg[0]));
    if (g.trace && blockIdx.x == 0 && lane == 0 && it < 8) g.trace[((size_t)w * 8 + it) * 8 + 2] = __builtin_amdgcn_s_memtime();
    // ---- q overwrites x (every x read above was issued earlier by this same wave)
    {
      int v = lane;
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        if (v < total) xt[v] = q[jj];
        v += 64;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- stage 2a: n^T tiles.  MFMA r: A = Cn[p][coordinate 32t + lane&31] with p = (r&3) + 8(r>>2) + 4h, B = lg[r]
    v16f accN[NT2];
    {
      const float* tbase = CnL + (4 * h) * 128 + j;
      float ta[NR], tn[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) ta[r] = tbase[((r & 3) + 8 * (r >> 2)) * 128];
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accN[nt][r] = 0.f;
        if (nt < g.nt) {
          const int nn = nt + 1 < g.nt ? 32 * (nt + 1) : 0;   // next tile's operands travel under this tile's MFMAs
#pragma unroll
          for (int r = 0; r < NR; ++r) tn[r] = tbase[((r & 3) + 8 * (r >> 2)) * 128 + nn];
#pragma unroll
          for (int r = 0; r < NR; ++r) accN[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[r], lg[r], accN[nt], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < NR; ++r) ta[r] = tn[r];
        }
      }
    }
    asm volatile("" :: "v"(accN[0][0]));
    if (g.trace && blockIdx.x == 0 && lane == 0 && it < 8) g.trace[((size_t)w * 8 + it) * 8 + 3] = __builtin_amdgcn_s_memtime();
    // ---- s = q . n : lane (h, pair) owns coordinates 32t + 8(r>>2) + 4h + (r&3): four float4 of q per tile
    const float4* qrow = xt + j * nch + h;
    float sp = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int ci = 8 * nt + 2 * rq + h;
        if (ci < nch) {
          const float4 qv = qrow[8 * nt + 2 * rq];
          sp = fmaf(qv.x, accN[nt][4 * rq], fmaf(qv.y, accN[nt][4 * rq + 1], fmaf(qv.z, accN[nt][4 * rq + 2], fmaf(qv.w, accN[nt][4 * rq + 3], sp))));
        }
      }
    }
    const float sfull = sp + __shfl_xor(sp, 32, 64);
    if (g.trace && blockIdx.x == 0 && lane == 0 && it < 8) g.trace[((size_t)w * 8 + it) * 8 + 4] = __builtin_amdgcn_s_memtime();
    // ---- stage 2b: r^T tiles and the distance
    float dsum = 0.f;
    {
      const float* tbase = ArL + (4 * h) * 128 + j;
      float ta[NR], tn[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) ta[r] = tbase[((r & 3) + 8 * (r >> 2)) * 128];
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
        if (nt < g.nt) {
          v16f accR;
#pragma unroll
          for (int r = 0; r < 16; ++r) accR[r] = 0.f;
          const int nn = nt + 1 < g.nt ? 32 * (nt + 1) : 0;
#pragma unroll
          for (int r = 0; r < NR; ++r) tn[r] = tbase[((r & 3) + 8 * (r >> 2)) * 128 + nn];
          float4 qv[4];
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) qv[rq] = (8 * nt + 2 * rq + h < nch) ? qrow[8 * nt + 2 * rq] : f4zero();
#pragma unroll
          for (int r = 0; r < NR; ++r) accR = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[r], lg[r], accR, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < NR; ++r) ta[r] = tn[r];
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {   // chunks past the row end have q = r = n = 0: they add f(0) = 0
            dsum += dist1(fmaf(-sfull, accN[nt][4 * rq], qv[rq].x + accR[4 * rq]), l1) +
                    dist1(fmaf(-sfull, accN[nt][4 * rq + 1], qv[rq].y + accR[4 * rq + 1]), l1) +
                    dist1(fmaf(-sfull, accN[nt][4 * rq + 2], qv[rq].z + accR[4 * rq + 2]), l1) +
                    dist1(fmaf(-sfull, accN[nt][4 * rq + 3], qv[rq].w + accR[4 * rq + 3]), l1);
          }
        }
      }
    }
    if (paired && lane == 0) turn[pairi] = 1 - side;  // release: the partner's matrix phases may start
    const float score = dsum + __shfl_xor(dsum, 32, 64);
    if (g.trace && blockIdx.x == 0 && lane == 0 && it < 8) g.trace[((size_t)w * 8 + it) * 8 + 5] = __builtin_amdgcn_s_memtime();

    if (h == 0 && row0 + j < a.n) a.score[row0 + j] = score;
    if (pre) { sid[lane] = nx_u; sid[32 + lane] = nx_i; sid[64 + lane] = nx_e; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (g.trace && blockIdx.x == 0 && lane == 0 && it < 8) g.trace[((size_t)w * 8 + it) * 8 + 6] = __builtin_amdgcn_s_memtime();
  }
  if (paired && lane == 0) { done[w] = 1; turn[pairi] = 1 - side; }
}

template <int J, int NT2, int NR>
int launch_pref3(const PrefArgs& a, const Fwd3Geom& g, hipStream_t st, const char* name) {
  const size_t lds = g.table_bytes + (size_t)g.nw * g.wave_bytes + 64;
  (void)hipFuncSetAttribute((const void*)pref_fwd3_kernel<J, NT2, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int64_t ntiles = (a.n + 31) / 32;
  const int grid = grid_for((ntiles + g.nw - 1) / g.nw, 256);
  Fwd3Geom gg = g;
  const char* tpath = getenv("KTUP_PREF_TRACE");     // debug only: dump s_memtime marks of workgroup 0
  const size_t tbytes = (size_t)8 * 8 * 8 * sizeof(unsigned long long);
  if (tpath && hipMalloc((void**)&gg.trace, tbytes) == hipSuccess) (void)hipMemsetAsync(gg.trace, 0, tbytes, st); else gg.trace = nullptr;
  hipLaunchKernelGGL((pref_fwd3_kernel<J, NT2, NR>), dim3(grid), dim3(g.nw * 64), lds, st, a, gg);
  if (gg.trace) {
    std::vector<unsigned long long> hbuf(8 * 8 * 8);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hbuf.data(), gg.trace, tbytes, hipMemcpyDeviceToHost);
    if (FILE* f = fopen(tpath, "w")) {
      for (int w = 0; w < 8; ++w)
        for (int it = 0; it < 8; ++it) {
          fprintf(f, "w%d it%d", w, it);
          for (int k = 0; k < 8; ++k) fprintf(f, " %llu", hbuf[((size_t)w * 8 + it) * 8 + k]);
          fprintf(f, "\n");
        }
      fclose(f);
    }
    (void)hipFree(gg.trace);
  }
  return check_launch(name);
}

// soft gate, d <= 128, P <= 32 and enough LDS for >= 2 waves; otherwise the caller falls back to pref_fwd2
bool fwd3_supported(const PrefArgs& a, int d, int P, Fwd3Geom* out) {
  if (a.gumbel != KTUP_GUMBEL_OFF || d > 128 || P > 32) return false;
  *out = fwd3_geom(d, P);
  { const char* e = getenv("KTUP_PREF_PAIR"); out->pair = e ? atoi(e) : 0; }
  { const char* e = getenv("KTUP_PREF_EXP"); out->exp = e ? atoi(e) : 0; }
  out->trace = nullptr;
  return out->nw >= 2;
}

template <int J, int NT2>
int dispatch_fwd3_nr(const PrefArgs& a, const Fwd3Geom& g, hipStream_t st, const char* name) {
  switch (g.nr) {   // stage-2 MFMAs per tile and table = 4 * ceil(P / 8)
    case 4: return launch_pref3<J, NT2, 4>(a, g, st, name);
    case 8: return launch_pref3<J, NT2, 8>(a, g, st, name);
    case 12: return launch_pref3<J, NT2, 12>(a, g, st, name);
    default: return launch_pref3<J, NT2, 16>(a, g, st, name);
  }
}

int dispatch_fwd3(const PrefArgs& a, int d, const Fwd3Geom& g, hipStream_t st, const char* name) {
  if (d <= 64) return dispatch_fwd3_nr<8, 2>(a, g, st, name);
  if (d <= 104) return dispatch_fwd3_nr<13, 4>(a, g, st, name);
  return dispatch_fwd3_nr<16, 4>(a, g, st, name);
}


// =============================================================================================================
// pref_fwd4: the same transposed matrix-core formulation as pref_fwd3 on v_mfma_f32_16x16x4_f32, 16 pairs per wave.
// Why: pref_fwd3's s_memtime trace (profiles/r01_fwd3_smemtime_trace.txt) shows a wave's gather (~14 k cycles, running at
// the ~10 B/clk/CU vector-memory rate) and its matrix phases (~20 k) strictly in series, with only 2 waves per SIMD to
// overlap them (12.8 KB LDS tile + 185 VGPRs per wave).  Halving the tile to 16 pairs halves the LDS tile and the
// accumulators (n^T: 7 x 4 registers instead of 4 x 16), so 16 waves (4 per SIMD) are resident per CU, and the finer
// tiles waste less: d=100 is 7 x 16 coordinates (not 4 x 32), and the preferences pack 4 per MFMA exactly (P=20 -> 5).
//   lane l: kq = l >> 4 (k slot 0..3), j = l & 15 (pair for B / D columns, table row for A).
//   stage 1: D1[t][slot i][pair], slot i = 4 kq' + reg  <->  preference p = 16 t + 4 reg + kq' (block-transposed so that a
//            partial last tile spreads its preferences over the 4 k slots); k-group g: lane reads float4 [16g + 4kq .. +3]
//            of its table row (A) / its pair's x row (B) and component c feeds MFMA c;
//   stage 2: MFMA m = 4 t + reg uses B = lg[t][reg] (register, no data movement) and A = T[16t + 4 reg + kq][16 ct + j];
//   tail   : lane (kq, pair) owns coordinates 16 ct + 4 kq + reg = ONE float4 of q per tile; sums over kq by 2 xor-adds.
typedef float v4 __attribute__((ext_vector_type(4)));

struct Fwd4Geom {
  int kg, pitchA4, pt, np, trow, tpitch, ct, nw;
  int exp;   // diagnostics only (KTUP_PREF_EXP): bit0 no row loads, bit1 no stage 1, bit2 no stage 2a, bit3 no stage 2b
  size_t table_bytes, wave_bytes;
};
inline Fwd4Geom fwd4_geom(int d, int P) {
  Fwd4Geom g{};
  g.kg = (d + 15) / 16;                       // stage-1 k groups of 16 coordinates
  g.pitchA4 = 4 * g.kg + 1;                   // odd float4 pitch of the slot-ordered logit table
  g.pt = (P + 15) / 16;                       // preference tiles of 16 slots
  g.np = (P + 3) / 4;                         // stage-2 MFMAs per coordinate tile and table,
  g.np = g.np <= 2 ? g.np : g.np <= 4 ? 4 : g.np <= 5 ? 5 : 8;   // rounded up to an instantiated count (1, 2, 4, 5, 8)
  g.trow = 4 * g.np;                          // rows of the stage-2 tables (zero padded)
  g.ct = (d + 15) / 16;                       // coordinate tiles
  g.tpitch = 16 * g.ct + ((16 * g.ct) % 32 == 0 ? 16 : 0);   // pitch == 16 (mod 32): the two rows a 32-lane group reads never collide
  g.table_bytes = (size_t)g.pt * 16 * g.pitchA4 * 16 + (size_t)2 * g.trow * g.tpitch * 4;
  g.wave_bytes = (((size_t)16 * (d / 4) + 3) * 16 + 3 * 16 * 4 + 15) & ~(size_t)15;
  const size_t budget = 160 * 1024 - g.table_bytes;
  g.nw = (int)(budget / g.wave_bytes);
  if (g.nw > 16) g.nw = 16;
  g.nw &= ~3;                                 // whole waves per SIMD
  return g;
}

template <int J, int CT, int NP>
__global__ __launch_bounds__(1024) void pref_fwd4_kernel(PrefArgs a, Fwd4Geom g) {
  constexpr int PT = (NP + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nch = a.nch, d = nch * 4, P = a.P;
  float4* AlogS = reinterpret_cast<float4*>(smem);                                     // [pt * 16 slots][pitchA4]
  float* CnS = reinterpret_cast<float*>(AlogS + g.pt * 16 * g.pitchA4);                 // [trow][tpitch]
  float* ArS = CnS + g.trow * g.tpitch;                                                 // [trow][tpitch]
  const int t = threadIdx.x, lane = t & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  char* wbase = reinterpret_cast<char*>(ArS + g.trow * g.tpitch) + (size_t)w * g.wave_bytes;
  float4* xt = reinterpret_cast<float4*>(wbase);                                        // [16 * nch] + 3 zero chunks
  int32_t* sid = reinterpret_cast<int32_t*>(xt + 16 * nch + 3);                         // [3][16]
  // ---- stage the tables once per workgroup
  {
    const float* Alog = reinterpret_cast<const float*>(a.Alog);
    const float* Ar = reinterpret_cast<const float*>(a.Ar);
    const float* Cn = reinterpret_cast<const float*>(a.Cn);
    const int dp = a.dp4 * 4, rowf = g.pitchA4 * 4, nA = g.pt * 16 * rowf, nT = g.trow * g.tpitch;
    float* AlogSf = reinterpret_cast<float*>(AlogS);
    for (int idx = t; idx < nA; idx += blockDim.x) {
      const int srow = idx / rowf, k = idx - srow * rowf;
      const int tt = srow >> 4, i = srow & 15;
      const int p = 16 * tt + 4 * (i & 3) + (i >> 2);          // slot -> preference (block transposed)
      AlogSf[idx] = (p < P && k < d) ? Alog[p * dp + k] : 0.f;
    }
    for (int idx = t; idx < nT; idx += blockDim.x) {
      const int p = idx / g.tpitch, c = idx - p * g.tpitch;
      const bool ok = p < P && c < d;
      CnS[idx] = ok ? Cn[p * dp + c] : 0.f;
      ArS[idx] = ok ? Ar[p * dp + c] : 0.f;
    }
    if (lane < 3) xt[16 * nch + lane] = f4zero();
  }
  __syncthreads();
  if (g.exp & 0xF0) {   // diagnostics: de-synchronise the waves of a SIMD (they start every phase together otherwise)
    const int hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_HW_ID[3:0] = wave slot
    const int slot = (g.exp & 128) ? (hw & 3) : (w >> 2);
    if (g.exp & 16) {
      if (slot == 1) __builtin_amdgcn_s_setprio(1);
      else if (slot == 2) __builtin_amdgcn_s_setprio(2);
      else if (slot == 3) __builtin_amdgcn_s_setprio(3);
    }
    const int reps = slot * ((g.exp & 32) ? 1 : 0) + slot * ((g.exp & 64) ? 2 : 0);
    for (int r = 0; r < reps; ++r) __builtin_amdgcn_s_sleep(64);   // 64 x 64 clk = 4096 clk each
  }
  const bool l1 = a.l1 != 0;
  const int64_t ntiles = (a.n + 15) / 16;
  const int total = 16 * nch;
  const int qstep = 64 / nch, rstep = 64 - qstep * nch;
  const int64_t wstride = (int64_t)gridDim.x * g.nw;
  bool first = true;
  for (int64_t tile_id = (int64_t)blockIdx.x * g.nw + w; tile_id < ntiles; tile_id += wstride) {
    const int64_t row0 = tile_id * 16;
    if (first && lane < 16) {
      const int64_t gr = row0 + lane;
      const bool ok = gr < a.n;
      const int64_t uid = ok ? a.u_ids[gr] : 0, iid = ok ? a.i_ids[gr] : 0;
      sid[lane] = (int32_t)uid;
      sid[16 + lane] = (int32_t)iid;
      sid[32 + lane] = a.E ? a.item2ent[iid] : 0;
    }
    first = false;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- gather: x -> LDS tile, q in registers
    float4 q[J];
    {
      int v = lane, row = lane / nch, c = lane - (lane / nch) * nch;
      asm volatile("" : "+v"(v), "+v"(row), "+v"(c));   // opaque per tile (LICM would hoist and spill the address sets)
      float4 uu[J], vv[J], ee[J];
      int vs[J];
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        vs[jj] = v;
        if (v < total && !(g.exp & 1)) {
          uu[jj] = a.U[(int64_t)sid[row] * a.ldu4 + c];
          vv[jj] = a.I[(int64_t)sid[16 + row] * a.ldi4 + c];
          ee[jj] = a.E ? a.E[(int64_t)sid[32 + row] * a.lde4 + c] : f4zero();
        } else {
          uu[jj] = f4zero(); vv[jj] = f4zero(); ee[jj] = f4zero();
        }
        v += 64; row += qstep; c += rstep;
        if (c >= nch) { c -= nch; ++row; }
      }
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        const float4 ve = vv[jj] + ee[jj];
        if (vs[jj] < total) xt[vs[jj]] = uu[jj] + ve;
        q[jj] = uu[jj] - ve;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- next tile's ids travel under the matrix phases
    int32_t nx_u = 0, nx_i = 0, nx_e = 0;
    const bool pre = lane < 16 && tile_id + wstride < ntiles;
    if (pre) {
      const int64_t gr = (tile_id + wstride) * 16 + lane;
      const bool ok = gr < a.n;
      const int64_t uid = ok ? a.u_ids[gr] : 0, iid = ok ? a.i_ids[gr] : 0;
      nx_u = (int32_t)uid; nx_i = (int32_t)iid;
      nx_e = a.E ? a.item2ent[iid] : 0;
    }
    // ---- stage 1: logits^T, PT independent accumulator chains
    v4 lg[PT];
#pragma unroll
    for (int tt = 0; tt < PT; ++tt) lg[tt] = (v4){0.f, 0.f, 0.f, 0.f};
    {
      const float4* xb = xt + j * nch + kq;
      const float4* ta = AlogS + j * g.pitchA4 + kq;
      for (int gk = 0; gk < ((g.exp & 2) ? 0 : g.kg); ++gk) {
        const float4 bv = xb[4 * gk];
        float4 av[PT];
#pragma unroll
        for (int tt = 0; tt < PT; ++tt) av[tt] = ta[tt * 16 * g.pitchA4 + 4 * gk];
#pragma unroll
        for (int tt = 0; tt < PT; ++tt) lg[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt].x, bv.x, lg[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < PT; ++tt) lg[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt].y, bv.y, lg[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < PT; ++tt) lg[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt].z, bv.z, lg[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < PT; ++tt) lg[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt].w, bv.w, lg[tt], 0, 0, 0);
      }
    }
    // ---- q overwrites x
    {
      int v = lane;
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        if (v < total) xt[v] = q[jj];
        v += 64;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- stage 2a: n^T coordinate tiles (two tiles in flight: independent accumulator chains)
    v4 accN[CT];
    const float* tn0 = CnS + kq * g.tpitch + j;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      accN[ct] = (v4){0.f, 0.f, 0.f, 0.f};
      if (ct < g.ct && !(g.exp & 4)) {
        float ta[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) ta[m] = tn0[(16 * (m >> 2) + 4 * (m & 3)) * g.tpitch + 16 * ct];
#pragma unroll
        for (int m = 0; m < NP; ++m) accN[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[m], lg[m >> 2][m & 3], accN[ct], 0, 0, 0);
      }
    }
    // ---- s = q . n
    const float4* qrow = xt + j * nch + kq;
    float sp = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (4 * ct + kq < nch) {
        const float4 qv = qrow[4 * ct];
        sp = fmaf(qv.x, accN[ct][0], fmaf(qv.y, accN[ct][1], fmaf(qv.z, accN[ct][2], fmaf(qv.w, accN[ct][3], sp))));
      }
    }
    sp += __shfl_xor(sp, 16, 64);
    const float sfull = sp + __shfl_xor(sp, 32, 64);
    // ---- stage 2b: r^T tiles and the distance
    float dsum = 0.f;
    const float* tr0 = ArS + kq * g.tpitch + j;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (ct < g.ct && !(g.exp & 8)) {
        float ta[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) ta[m] = tr0[(16 * (m >> 2) + 4 * (m & 3)) * g.tpitch + 16 * ct];
        const float4 qv = (4 * ct + kq < nch) ? qrow[4 * ct] : f4zero();
        v4 accR = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NP; ++m) accR = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[m], lg[m >> 2][m & 3], accR, 0, 0, 0);
        dsum += dist1(fmaf(-sfull, accN[ct][0], qv.x + accR[0]), l1) + dist1(fmaf(-sfull, accN[ct][1], qv.y + accR[1]), l1) +
                dist1(fmaf(-sfull, accN[ct][2], qv.z + accR[2]), l1) + dist1(fmaf(-sfull, accN[ct][3], qv.w + accR[3]), l1);
      }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    const float score = dsum + __shfl_xor(dsum, 32, 64);
    if (kq == 0 && row0 + j < a.n) a.score[row0 + j] = score;
    if (pre) { sid[lane] = nx_u; sid[16 + lane] = nx_i; sid[32 + lane] = nx_e; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int J, int CT, int NP>
int launch_pref4(const PrefArgs& a, const Fwd4Geom& g, hipStream_t st, const char* name) {
  const size_t lds = g.table_bytes + (size_t)g.nw * g.wave_bytes;
  (void)hipFuncSetAttribute((const void*)pref_fwd4_kernel<J, CT, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int64_t ntiles = (a.n + 15) / 16;
  const int grid = grid_for((ntiles + g.nw - 1) / g.nw, 256);
  hipLaunchKernelGGL((pref_fwd4_kernel<J, CT, NP>), dim3(grid), dim3(g.nw * 64), lds, st, a, g);
  return check_launch(name);
}

template <int J, int CT>
int dispatch_fwd4_np(const PrefArgs& a, const Fwd4Geom& g, hipStream_t st, const char* name) {
  if (g.np <= 1) return launch_pref4<J, CT, 1>(a, g, st, name);
  if (g.np <= 2) return launch_pref4<J, CT, 2>(a, g, st, name);
  if (g.np <= 4) return launch_pref4<J, CT, 4>(a, g, st, name);
  if (g.np <= 5) return launch_pref4<J, CT, 5>(a, g, st, name);
  return launch_pref4<J, CT, 8>(a, g, st, name);
}

// soft gate, d <= 128, P <= 32
bool fwd4_supported(const PrefArgs& a, int d, int P, Fwd4Geom* out) {
  if (a.gumbel != KTUP_GUMBEL_OFF || d > 128 || P > 32) return false;
  *out = fwd4_geom(d, P);
  { const char* e = getenv("KTUP_PREF_EXP"); out->exp = e ? atoi(e) : 0; }
  return out->nw >= 4;
}

int dispatch_fwd4(const PrefArgs& a, int d, const Fwd4Geom& g, hipStream_t st, const char* name) {
  if (d <= 64) return dispatch_fwd4_np<4, 4>(a, g, st, name);
  if (d <= 112) return dispatch_fwd4_np<7, 7>(a, g, st, name);
  return dispatch_fwd4_np<8, 8>(a, g, st, name);
}

template <int CH, int NW>
int launch_pref(bool bwd, const PrefArgs& a, hipStream_t st, const char* name) {
  const int64_t ntiles = (a.n + TR - 1) / TR;
  size_t lds = Tile<CH, NW>::bytes(a.nch, a.lp);
  const size_t bwd_fixed = (size_t)2 * TR * a.lp * 4 + 16;
  constexpr int KSL = (CH % 4 == 0) ? 4 : 1;            // sliced variant instantiated for the wide geometries only
  bool sliced = false;
  if (bwd) {
    const size_t whole = lds + bwd_fixed + (size_t)3 * TR * a.nch * 16;
    sliced = whole > 160 * 1024 && KSL > 1;
    lds += bwd_fixed + (size_t)3 * TR * (sliced ? NW * (CH / KSL) : a.nch) * 16;
  }
  if (lds > 160 * 1024) return set_error(KTUP_ERR_UNSUPPORTED, "%s: tile needs %zu B of LDS", name, lds);
  const int grid = grid_for(ntiles, 256 * 4);
  if (bwd && sliced) {
    (void)hipFuncSetAttribute((const void*)pref_bwd_kernel<CH, NW, KSL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((pref_bwd_kernel<CH, NW, KSL>), dim3(grid), dim3(NW * 64), lds, st, a);
  } else if (bwd) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)pref_bwd_kernel<CH, NW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((pref_bwd_kernel<CH, NW, 1>), dim3(grid), dim3(NW * 64), lds, st, a);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)pref_fwd_kernel<CH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((pref_fwd_kernel<CH, NW>), dim3(grid), dim3(NW * 64), lds, st, a);
  }
  return check_launch(name);
}

int run_pref(bool bwd, const char* name, const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E,
             int64_t lde, const int32_t* item2ent, int64_t ent_pad, const float* pref_ws, int n_pref, int d,
             const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform,
             uint64_t seed, uint64_t offset, float* score, const float* gscore, float* gU, float* gI, float* gE,
             float* gA, float* gC, void* stream) {
  KTUP_REQUIRE(n >= 0, "%s: negative row count", name);
  if (n == 0) return KTUP_OK;
  const PrefGeom g = pref_geom(d, n_pref);
  if (!g.ok)
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size must be a multiple of 4 in [4, 256] (got %d), n_pref > 0", name, d);
  KTUP_REQUIRE(U && I && pref_ws && u_ids && i_ids, "%s: null pointer argument", name);
  KTUP_REQUIRE((E == nullptr) == (item2ent == nullptr), "%s: E and item2ent must be given together", name);
  KTUP_REQUIRE(aligned16(U) && aligned16(I) && aligned16(E) && aligned16(pref_ws), "%s: tables must be 16-byte aligned", name);
  KTUP_REQUIRE(ldu % 4 == 0 && ldi % 4 == 0 && (!E || lde % 4 == 0), "%s: row pitches must be multiples of 4 floats", name);
  KTUP_REQUIRE(gumbel_mode >= KTUP_GUMBEL_OFF && gumbel_mode <= KTUP_GUMBEL_PHILOX_DEV, "%s: bad gumbel_mode %d", name, gumbel_mode);
  KTUP_REQUIRE((gumbel_mode != KTUP_GUMBEL_INPUT && gumbel_mode != KTUP_GUMBEL_PHILOX_DEV) || uniform,
               "%s: KTUP_GUMBEL_INPUT / KTUP_GUMBEL_PHILOX_DEV need the `uniform` pointer", name);
  PrefArgs a{};
  a.U = reinterpret_cast<const float4*>(U); a.I = reinterpret_cast<const float4*>(I); a.E = reinterpret_cast<const float4*>(E);
  a.ldu4 = ldu / 4; a.ldi4 = ldi / 4; a.lde4 = lde / 4;
  a.item2ent = item2ent;
  const float* base = pref_ws;
  a.Alog = reinterpret_cast<const float4*>(base);
  a.Ar = reinterpret_cast<const float4*>(base + (size_t)g.ppad * g.dp);
  a.Cn = reinterpret_cast<const float4*>(base + (size_t)(g.ppad + n_pref) * g.dp);
  a.P = n_pref; a.ppad = g.ppad; a.lp = g.ppad | 1; a.nch = d / 4; a.dp4 = g.dp / 4;
  const PrefGeom2 g2 = pref_geom2(d, n_pref);
  a.Alog2 = base + ws1_floats(g, n_pref);
  a.AC2 = a.Alog2 + (size_t)g2.ppad2 * g2.dpa;
  a.dpa16 = g2.dpa / 16; a.ppad2 = g2.ppad2;
  a.u_ids = u_ids; a.i_ids = i_ids; a.n = n; a.l1 = l1; a.gumbel = gumbel_mode; a.uniform = uniform;
  a.seed = seed; a.offset = offset; a.score = score;
  a.gscore = gscore; a.gU = gU; a.gI = gI; a.gE = gE; a.gA = gA; a.gC = gC; a.ent_pad = ent_pad;
  a.alpha_beta = E ? 0.5f : 1.0f;
  if (bwd) {
    KTUP_REQUIRE(gscore && gU && gI && gA && gC && (!E || gE), "%s: null gradient pointer", name);
    const int nt = g.NW * 64;
    KTUP_REQUIRE(nt >= d, "%s: internal geometry error", name);
    const int ngrp = nt / d, pg = (n_pref + ngrp - 1) / ngrp;
    if (pg > 16) return set_error(KTUP_ERR_UNSUPPORTED, "%s: n_pref %d too large for d=%d (max %d)", name, n_pref, d, 16 * ngrp);
  } else {
    KTUP_REQUIRE(score, "%s: null score pointer", name);
  }
  hipStream_t st = (hipStream_t)stream;
  if (!bwd) {  // KTUP_PREF_FWD selects the forward variant (A/B measurements); default = tuned kernel, one pair per lane
    const char* env = getenv("KTUP_PREF_FWD");
    const int variant = env ? atoi(env) : 7;   // 0 = first kernel, 2 = SGPR-FMA kernel, 3 = 32x32x2 MFMA, 4 = 16x16x4 MFMA (run-time geometry), 7 (default) = compile-time-geometry kernel first
    if (variant == 7) {
      const int rc = pref_fwd_mc(reinterpret_cast<const float*>(a.U), a.ldu4 * 4, reinterpret_cast<const float*>(a.I), a.ldi4 * 4,
                                 reinterpret_cast<const float*>(a.E), a.lde4 * 4, a.item2ent, reinterpret_cast<const float*>(a.Alog),
                                 reinterpret_cast<const float*>(a.Ar), reinterpret_cast<const float*>(a.Cn), a.dp4 * 4, n_pref, d,
                                 a.u_ids, a.i_ids, a.n, a.l1, a.gumbel, a.uniform, a.seed, a.offset, a.score, st, name);
      if (rc != 1) return rc;
    }
    Fwd4Geom g4;
    if ((variant == 4 || variant == 7) && fwd4_supported(a, d, n_pref, &g4)) return dispatch_fwd4(a, d, g4, st, name);
    Fwd3Geom g3;
    if (variant >= 3 && fwd3_supported(a, d, n_pref, &g3)) return dispatch_fwd3(a, d, g3, st, name);
    if (variant != 0) return dispatch_fwd2(a, d, n_pref, st, name);
  }
  if (bwd) {   // matrix-core backward (KTUP_PREF_BWD=0 keeps the first kernel for A/B runs)
    const char* env = getenv("KTUP_PREF_BWD");
    if (!env || atoi(env) != 0) {
      const int rc = pref_bwd_mc(reinterpret_cast<const float*>(a.U), a.ldu4 * 4, reinterpret_cast<const float*>(a.I), a.ldi4 * 4,
                                 reinterpret_cast<const float*>(a.E), a.lde4 * 4, a.item2ent, a.ent_pad,
                                 reinterpret_cast<const float*>(a.Alog), reinterpret_cast<const float*>(a.Ar),
                                 reinterpret_cast<const float*>(a.Cn), a.dp4 * 4, a.alpha_beta, n_pref, d, a.u_ids, a.i_ids, a.n, a.l1,
                                 a.gumbel, a.uniform, a.seed, a.offset, a.gscore, a.gU, a.gI, a.gE, a.gA, a.gC, st, name);
      if (rc != 1) return rc;
    }
  }
  if (g.CH == 4 && g.NW == 4) return launch_pref<4, 4>(bwd, a, st, name);
  if (g.CH == 7 && g.NW == 4) return launch_pref<7, 4>(bwd, a, st, name);
  if (g.CH == 8 && g.NW == 4) return launch_pref<8, 4>(bwd, a, st, name);
  return launch_pref<8, 8>(bwd, a, st, name);
}

}  // namespace

extern "C" size_t ktup_pref_workspace_bytes(int d, int n_pref) {
  const PrefGeom g = pref_geom(d, n_pref);
  return g.ok ? ws_floats(g, d, n_pref) * sizeof(float) : 0;
}

extern "C" int ktup_pref_prepare(const float* pref, const float* pref_norm, const float* rel, const float* norm, int64_t ld,
                                 int n_pref, int d, float* ws, void* stream) {
  const PrefGeom g = pref_geom(d, n_pref);
  if (!g.ok)
    return set_error(KTUP_ERR_UNSUPPORTED, "ktup_pref_prepare: embedding_size must be a multiple of 4 in [4, 256] (got %d)", d);
  KTUP_REQUIRE(pref && pref_norm && ws, "ktup_pref_prepare: null pointer argument");
  KTUP_REQUIRE((rel == nullptr) == (norm == nullptr), "ktup_pref_prepare: rel and norm must be given together");
  KTUP_REQUIRE(ld >= d, "ktup_pref_prepare: pitch %lld < d", (long long)ld);
  KTUP_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 63u) == 0, "ktup_pref_prepare: workspace must be 64-byte aligned");
  const int total = (int)ws_floats(g, d, n_pref);
  hipLaunchKernelGGL(pref_prepare_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, pref, pref_norm, rel,
                     norm, ld, n_pref, d, g.dp, g.ppad, pref_geom2(d, n_pref), (int64_t)ws1_floats(g, n_pref), ws);
  return check_launch("ktup_pref_prepare");
}

extern "C" int ktup_score_tup_fwd(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* pref_ws, int n_pref,
                                  int d, const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                                  const float* uniform, uint64_t seed, uint64_t offset, float* score, void* stream) {
  return run_pref(false, "ktup_score_tup_fwd", U, ldu, I, ldi, nullptr, 0, nullptr, -1, pref_ws, n_pref, d, u_ids, i_ids, n,
                  l1, gumbel_mode, uniform, seed, offset, score, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int ktup_score_ktup_fwd(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                   const int32_t* item2ent, const float* pref_ws, int n_pref, int d, const int64_t* u_ids,
                                   const int64_t* i_ids, int64_t n, int l1, int gumbel_mode, const float* uniform,
                                   uint64_t seed, uint64_t offset, float* score, void* stream) {
  KTUP_REQUIRE(E && item2ent, "ktup_score_ktup_fwd: E and item2ent are required (use ktup_score_tup_fwd for TUP)");
  return run_pref(false, "ktup_score_ktup_fwd", U, ldu, I, ldi, E, lde, item2ent, -1, pref_ws, n_pref, d, u_ids, i_ids, n, l1,
                  gumbel_mode, uniform, seed, offset, score, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int ktup_score_tup_bwd(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* pref_ws, int n_pref,
                                  int d, const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                                  const float* uniform, uint64_t seed, uint64_t offset, const float* gscore, float* gU,
                                  float* gI, float* gA, float* gC, void* stream) {
  return run_pref(true, "ktup_score_tup_bwd", U, ldu, I, ldi, nullptr, 0, nullptr, -1, pref_ws, n_pref, d, u_ids, i_ids, n, l1,
                  gumbel_mode, uniform, seed, offset, nullptr, gscore, gU, gI, nullptr, gA, gC, stream);
}

extern "C" int ktup_score_ktup_bwd(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                   const int32_t* item2ent, int64_t ent_pad, const float* pref_ws, int n_pref, int d,
                                   const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                                   const float* uniform, uint64_t seed, uint64_t offset, const float* gscore, float* gU,
                                   float* gI, float* gE, float* gA, float* gC, void* stream) {
  KTUP_REQUIRE(E && item2ent, "ktup_score_ktup_bwd: E and item2ent are required (use ktup_score_tup_bwd for TUP)");
  return run_pref(true, "ktup_score_ktup_bwd", U, ldu, I, ldi, E, lde, item2ent, ent_pad, pref_ws, n_pref, d, u_ids, i_ids, n,
                  l1, gumbel_mode, uniform, seed, offset, nullptr, gscore, gU, gI, gE, gA, gC, stream);
}
