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
                                    int d, int dp, int ppad, float* __restrict__ ws) {
  const float beta = rel ? 0.5f : 1.0f;
  const int total = (ppad + 2 * P) * dp;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    float v = 0.f;
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
  }
}

struct PrefArgs {
  const float4 *U, *I, *E;      // tables as float4 (E may be null)
  int64_t ldu4, ldi4, lde4;     // pitches in float4
  const int32_t* item2ent;      // null for TUP
  const float4 *Alog, *Ar, *Cn; // prepared tables, pitch dp4
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
             float* gA, float* gC, void* stream, int64_t n_user_rows = 0, int64_t n_item_rows = 0, void* bws = nullptr) {
  KTUP_REQUIRE(n >= 0, "%s: negative row count", name);
  if (n == 0) return KTUP_OK;
  const PrefGeom g = pref_geom(d, n_pref);
  const bool rows = !g.ok && pref_row_covers(d, n_pref);      // wider than the tile kernels hold: one wave per pair (ktup_score_pref_row.hip)
  if (!g.ok && !rows)
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size must be a positive multiple of 4 (got %d), 0 < n_pref (<= 128 beyond 256 columns)", name, d);
  KTUP_REQUIRE(U && I && pref_ws && u_ids && i_ids, "%s: null pointer argument", name);
  KTUP_REQUIRE((E == nullptr) == (item2ent == nullptr), "%s: E and item2ent must be given together", name);
  KTUP_REQUIRE(aligned16(U) && aligned16(I) && aligned16(E) && aligned16(pref_ws), "%s: tables must be 16-byte aligned", name);
  KTUP_REQUIRE(ldu % 4 == 0 && ldi % 4 == 0 && (!E || lde % 4 == 0), "%s: row pitches must be multiples of 4 floats", name);
  KTUP_REQUIRE(gumbel_mode >= KTUP_GUMBEL_OFF && gumbel_mode <= KTUP_GUMBEL_PHILOX_DEV, "%s: bad gumbel_mode %d", name, gumbel_mode);
  KTUP_REQUIRE((gumbel_mode != KTUP_GUMBEL_INPUT && gumbel_mode != KTUP_GUMBEL_PHILOX_DEV) || uniform,
               "%s: KTUP_GUMBEL_INPUT / KTUP_GUMBEL_PHILOX_DEV need the `uniform` pointer", name);
  if (rows) {
    if (bwd) KTUP_REQUIRE(gscore && gU && gI && gA && gC && (!E || gE), "%s: null gradient pointer", name);
    else KTUP_REQUIRE(score, "%s: null score pointer", name);
    return pref_row(bwd, name, U, ldu, I, ldi, E, lde, item2ent, ent_pad, pref_ws, n_pref, d, u_ids, i_ids, n, 0, 0, l1, gumbel_mode,
                    uniform, seed, offset, score, gscore, gU, gI, gE, gA, gC, (hipStream_t)stream);
  }
  PrefArgs a{};
  a.U = reinterpret_cast<const float4*>(U); a.I = reinterpret_cast<const float4*>(I); a.E = reinterpret_cast<const float4*>(E);
  a.ldu4 = ldu / 4; a.ldi4 = ldi / 4; a.lde4 = lde / 4;
  a.item2ent = item2ent;
  const float* base = pref_ws;
  a.Alog = reinterpret_cast<const float4*>(base);
  a.Ar = reinterpret_cast<const float4*>(base + (size_t)g.ppad * g.dp);
  a.Cn = reinterpret_cast<const float4*>(base + (size_t)(g.ppad + n_pref) * g.dp);
  a.P = n_pref; a.ppad = g.ppad; a.lp = g.ppad | 1; a.nch = d / 4; a.dp4 = g.dp / 4;
  a.u_ids = u_ids; a.i_ids = i_ids; a.n = n; a.l1 = l1; a.gumbel = gumbel_mode; a.uniform = uniform;
  a.seed = seed; a.offset = offset; a.score = score;
  a.gscore = gscore; a.gU = gU; a.gI = gI; a.gE = gE; a.gA = gA; a.gC = gC; a.ent_pad = ent_pad;
  a.alpha_beta = E ? 0.5f : 1.0f;
  if (bwd) {
    KTUP_REQUIRE(gscore && gU && gI && gA && gC && (!E || gE), "%s: null gradient pointer", name);
    const int nt = g.NW * 64;
    KTUP_REQUIRE(nt >= d, "%s: internal geometry error", name);
    const int ngrp = nt / d, pg = (n_pref + ngrp - 1) / ngrp;
    if (pg > 16)     // more preferences than the table-gradient pass holds per thread (16 x the d-column groups of a workgroup: 32 at
      return pref_row(true, name, U, ldu, I, ldi, E, lde, item2ent, ent_pad, pref_ws, n_pref, d, u_ids, i_ids, n, 0, 0, l1, gumbel_mode,
                      uniform, seed, offset, nullptr, gscore, gU, gI, gE, gA, gC, (hipStream_t)stream, g.ppad, g.dp);   // d = 100): one wave per pair
  } else {
    KTUP_REQUIRE(score, "%s: null score pointer", name);
  }
  hipStream_t st = (hipStream_t)stream;
  if (opt_pref_mc()) {   // matrix-core kernels (compile-time geometry) for the shapes they cover; 1 = not covered
    if (bwd) {
      // large batches with a workspace: per-pair row gradients by plain stores + a reduction by sorted segments instead of
      // d float atomics per gathered row (ktup_segreduce.hip)
      const size_t gbytes = (((size_t)n * d * sizeof(float)) + 255) & ~(size_t)255;
      float* GU = nullptr; float* GV = nullptr; char* swsU = nullptr; char* swsI = nullptr;
      if (bws && n_user_rows > 0 && n_item_rows > 0 && opt_seg_bwd_min() > 0 && n >= opt_seg_bwd_min() && (d == 64 || d == 100 || d == 128 || d == 256)) {
        GU = reinterpret_cast<float*>(bws);
        GV = reinterpret_cast<float*>(reinterpret_cast<char*>(bws) + gbytes);
        swsU = reinterpret_cast<char*>(bws) + 2 * gbytes;
        swsI = swsU + seg_ws_bytes(n, n_user_rows);
        if (!seg_covers(GU, d, d, n, n, n_user_rows, gU, ldu, nullptr, nullptr, 0, swsU) ||
            !seg_covers(GV, d, d, n, n, n_item_rows, gI, ldi, E ? item2ent : nullptr, gE, lde, swsI))
          return set_error(KTUP_ERR_UNSUPPORTED, "%s: segment reduction does not cover this shape", name);
      }
      // the two counting sorts depend on the ids alone: they run on the library's side stream beside the gradient kernel
      hipStream_t side = nullptr;
      if (GU) {
        side = fork_side(st);
        hipStream_t ss = side ? side : st;
        int r1 = seg_sort(u_ids, nullptr, n, n, n_user_rows, swsU, ss, name);
        if (r1 == KTUP_OK) r1 = seg_sort(i_ids, nullptr, n, n, n_item_rows, swsI, ss, name);
        if (r1 != KTUP_OK) { join_side(st, side); return r1; }
      }
      const int rc = pref_bwd_mc(reinterpret_cast<const float*>(a.U), a.ldu4 * 4, reinterpret_cast<const float*>(a.I), a.ldi4 * 4,
                                 reinterpret_cast<const float*>(a.E), a.lde4 * 4, a.item2ent, a.ent_pad, reinterpret_cast<const float*>(a.Alog),
                                 reinterpret_cast<const float*>(a.Ar), reinterpret_cast<const float*>(a.Cn), a.dp4 * 4, a.alpha_beta, n_pref, d,
                                 a.u_ids, a.i_ids, a.n, a.l1, a.gumbel, a.uniform, a.seed, a.offset, a.gscore, a.gU, a.gI, a.gE, a.gA, a.gC, st,
                                 name, GU, GV);
      join_side(st, side);
      if (rc == KTUP_OK && GU) {
        // the two reductions touch disjoint tables (gU | gI, gE): side by side, each alone is bound by the latency of its row gathers
        hipStream_t s2 = fork_side(st);
        int r2 = seg_apply(GU, d, d, n, n, n, n_user_rows, gU, ldu, nullptr, -1, nullptr, 0, swsU, s2 ? s2 : st, name);
        if (r2 == KTUP_OK) r2 = seg_apply(GV, d, d, n, n, n, n_item_rows, gI, ldi, E ? item2ent : nullptr, ent_pad, gE, lde, swsI, st, name);
        join_side(st, s2);
        return r2;
      }
      if (rc != 1) return rc;
      if (GU) {    // the matrix-core kernel does not cover this shape after all: fall through to the atomics kernels
        GU = GV = nullptr;
      }
    } else {
      const int rc = pref_fwd_mc(reinterpret_cast<const float*>(a.U), a.ldu4 * 4, reinterpret_cast<const float*>(a.I), a.ldi4 * 4,
                                 reinterpret_cast<const float*>(a.E), a.lde4 * 4, a.item2ent, reinterpret_cast<const float*>(a.Alog),
                                 reinterpret_cast<const float*>(a.Ar), reinterpret_cast<const float*>(a.Cn), a.dp4 * 4, n_pref, d, a.u_ids,
                                 a.i_ids, a.n, a.l1, a.gumbel, a.uniform, a.seed, a.offset, a.score, st, name);
      if (rc != 1) return rc;
    }
  }
  // generic kernels: any d % 4 == 0 up to 256, both gates
  if (g.CH == 4 && g.NW == 4) return launch_pref<4, 4>(bwd, a, st, name);
  if (g.CH == 7 && g.NW == 4) return launch_pref<7, 4>(bwd, a, st, name);
  if (g.CH == 8 && g.NW == 4) return launch_pref<8, 4>(bwd, a, st, name);
  return launch_pref<8, 8>(bwd, a, st, name);
}

}  // namespace

extern "C" size_t ktup_pref_workspace_bytes(int d, int n_pref) {
  const PrefGeom g = pref_geom(d, n_pref);
  if (!g.ok) return pref_row_covers(d, n_pref) ? pref_row_ws_floats(d, n_pref) * sizeof(float) : 0;
  return ws_floats(g, n_pref) * sizeof(float);
}

extern "C" int ktup_pref_prepare(const float* pref, const float* pref_norm, const float* rel, const float* norm, int64_t ld,
                                 int n_pref, int d, float* ws, void* stream) {
  PrefGeom g = pref_geom(d, n_pref);
  if (!g.ok && pref_row_covers(d, n_pref)) { g.dp = d; g.ppad = n_pref; g.ok = true; }     // plain [P][d] blocks for the row kernels
  if (!g.ok)
    return set_error(KTUP_ERR_UNSUPPORTED, "ktup_pref_prepare: embedding_size must be a positive multiple of 4 (got %d), 0 < n_pref (<= 128 beyond 256 columns)", d);
  KTUP_REQUIRE(pref && pref_norm && ws, "ktup_pref_prepare: null pointer argument");
  KTUP_REQUIRE((rel == nullptr) == (norm == nullptr), "ktup_pref_prepare: rel and norm must be given together");
  KTUP_REQUIRE(ld >= d, "ktup_pref_prepare: pitch %lld < d", (long long)ld);
  KTUP_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 63u) == 0, "ktup_pref_prepare: workspace must be 64-byte aligned");
  const int total = (g.ppad + 2 * n_pref) * g.dp;
  hipLaunchKernelGGL(pref_prepare_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, pref, pref_norm, rel,
                     norm, ld, n_pref, d, g.dp, g.ppad, ws);
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

// Backward with a workspace: for n >= option seg_bwd_min (default 8192) and the matrix-core shapes, the row gradients are
// written per pair and reduced per table row by sorted segments (ktup_segreduce.hip) instead of float atomics; otherwise exactly
// ktup_score_{tup,ktup}_bwd.  n_user_rows / n_item_rows: rows of U / I (the key range of the counting sort).
extern "C" size_t ktup_score_pref_bwd_workspace_bytes(int64_t n, int d, int64_t n_user_rows, int64_t n_item_rows) {
  if (n <= 0 || d <= 0 || n_user_rows <= 0 || n_item_rows <= 0) return 0;
  if (opt_seg_bwd_min() <= 0 || n < opt_seg_bwd_min() || !(d == 64 || d == 100 || d == 128 || d == 256)) return 0;
  const size_t gbytes = (((size_t)n * d * sizeof(float)) + 255) & ~(size_t)255;
  return 2 * gbytes + seg_ws_bytes(n, n_user_rows) + seg_ws_bytes(n, n_item_rows);     // both sorts are alive at once
}

extern "C" int ktup_score_tup_bwd_ws(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* pref_ws, int n_pref,
                                     int d, const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                                     const float* uniform, uint64_t seed, uint64_t offset, const float* gscore, float* gU,
                                     float* gI, float* gA, float* gC, int64_t n_user_rows, int64_t n_item_rows, void* ws, void* stream) {
  return run_pref(true, "ktup_score_tup_bwd_ws", U, ldu, I, ldi, nullptr, 0, nullptr, -1, pref_ws, n_pref, d, u_ids, i_ids, n, l1,
                  gumbel_mode, uniform, seed, offset, nullptr, gscore, gU, gI, nullptr, gA, gC, stream, n_user_rows, n_item_rows, ws);
}

extern "C" int ktup_score_ktup_bwd_ws(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                      const int32_t* item2ent, int64_t ent_pad, const float* pref_ws, int n_pref, int d,
                                      const int64_t* u_ids, const int64_t* i_ids, int64_t n, int l1, int gumbel_mode,
                                      const float* uniform, uint64_t seed, uint64_t offset, const float* gscore, float* gU,
                                      float* gI, float* gE, float* gA, float* gC, int64_t n_user_rows, int64_t n_item_rows, void* ws,
                                      void* stream) {
  KTUP_REQUIRE(E && item2ent, "ktup_score_ktup_bwd_ws: E and item2ent are required (use ktup_score_tup_bwd_ws for TUP)");
  return run_pref(true, "ktup_score_ktup_bwd_ws", U, ldu, I, ldi, E, lde, item2ent, ent_pad, pref_ws, n_pref, d, u_ids, i_ids, n,
                  l1, gumbel_mode, uniform, seed, offset, nullptr, gscore, gU, gI, gE, gA, gC, stream, n_user_rows, n_item_rows, ws);
}
