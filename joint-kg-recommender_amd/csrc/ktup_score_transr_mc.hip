// K4 forward on the matrix cores: TransR score  sum_k dist( M_r (h - t) + r )_k   (transR.py:65-78 + utils/misc.py:21-26).
//
// The reference gathers one d x d projection matrix PER TRIPLE (index_select on a (R x d^2) table: 40 KB per triple at
// d = 100, although only R = 20 distinct matrices exist) and runs a batched mat-vec.  Here the batch is bucketed by
// relation (counting sort of the triple indices, three tiny launches into caller scratch), and a workgroup owns up to
// 1024 triples of ONE relation: M_r is staged into LDS once and the projection becomes
//     Y^T (d x 16 triples) = M_r (d x d) . Q^T (d x 16),   Q = H - T
// on v_mfma_f32_16x16x4_f32 per wave: A = rows of M_r from LDS, B = the wave's Q tile from LDS (held in registers across
// the output tiles), D = 4 output coordinates per lane.  4 d^2 flop per triple (12.3 GFLOP at 307,200 triples) puts this
// kernel on the fp32 matrix pipe, not on HBM: ~175 MFMAs per 16 triples.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ktup_common.h"
#include "ktup_lane_swap.h"
#include "ktup_pref_geom.h"

namespace ktup {
namespace {

constexpr int ST = 1024;   // triples per workgroup pass (one relation)

struct Buckets {           // int32 scratch laid out by transr_mc_workspace_bytes
  int32_t *cnt, *cursor, *off, *tile_off, *perm;
};

KTUP_DEV Buckets carve(void* ws, int64_t n, int n_rel) {
  Buckets b;
  int32_t* p = reinterpret_cast<int32_t*>(ws);
  b.cnt = p; p += n_rel;
  b.cursor = p; p += n_rel;
  b.off = p; p += n_rel + 1;
  b.tile_off = p; p += n_rel + 1;
  b.perm = p;
  return b;
}

// Workgroup-local histograms in LDS first: 307,200 triples on 20 relations would otherwise serialise on 20 L2 atomics.
__global__ __launch_bounds__(256) void bucket_hist_kernel(const int64_t* __restrict__ r, int64_t n, int n_rel, void* ws) {
  extern __shared__ int32_t lcnt[];
  const Buckets b = carve(ws, n, n_rel);
  for (int k = threadIdx.x; k < n_rel; k += 256) lcnt[k] = 0;
  __syncthreads();
  const int64_t chunk = (n + gridDim.x - 1) / gridDim.x, lo = (int64_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&lcnt[(int)r[i]], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < n_rel; k += 256)
    if (lcnt[k]) atomicAdd(&b.cnt[k], lcnt[k]);
}

__global__ void bucket_scan_kernel(int64_t n, int n_rel, void* ws) {   // one thread: n_rel <= 4096
  const Buckets b = carve(ws, n, n_rel);
  int o = 0, t = 0;
  for (int k = 0; k < n_rel; ++k) {
    b.off[k] = o; b.tile_off[k] = t;
    o += b.cnt[k];
    t += (b.cnt[k] + ST - 1) / ST;
  }
  b.off[n_rel] = o; b.tile_off[n_rel] = t;
}

// Same chunking as the histogram: a workgroup reserves one range per relation (one global atomic each) and its threads
// take slots inside it with LDS atomics.  Order inside a bucket is irrelevant: rows are scored independently.
__global__ __launch_bounds__(256) void bucket_scatter_kernel(const int64_t* __restrict__ r, int64_t n, int n_rel, void* ws) {
  extern __shared__ int32_t lds[];
  int32_t* lcnt = lds;             // [n_rel] local counts, then local cursors
  int32_t* lbase = lds + n_rel;    // [n_rel] start of this workgroup's range inside the bucket
  const Buckets b = carve(ws, n, n_rel);
  for (int k = threadIdx.x; k < n_rel; k += 256) lcnt[k] = 0;
  __syncthreads();
  const int64_t chunk = (n + gridDim.x - 1) / gridDim.x, lo = (int64_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&lcnt[(int)r[i]], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < n_rel; k += 256) {
    lbase[k] = lcnt[k] ? b.off[k] + atomicAdd(&b.cursor[k], lcnt[k]) : 0;
    lcnt[k] = 0;
  }
  __syncthreads();
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const int rr = (int)r[i];
    b.perm[lbase[rr] + atomicAdd(&lcnt[rr], 1)] = (int32_t)i;
  }
}

template <int NCH_>
struct RGeom {
  static constexpr int NCH = NCH_, D = 4 * NCH;
  static constexpr int KG = (D + 15) / 16, CT = KG;
  static constexpr bool TAIL1 = NCH - 4 * (KG - 1) == 1;
  static constexpr int KGF = TAIL1 ? KG - 1 : KG;
  static constexpr int J = (16 * NCH + 63) / 64;
  static constexpr int TOTAL = 16 * NCH;
  static constexpr int PITCH4 = 4 * KG + 1;                 // odd float4 pitch of the staged M_r rows
  static constexpr int M_F4 = 16 * CT * PITCH4;             // rows padded to 16 CT with zeros
  static constexpr int XT_F4 = 16 * NCH + 3;
  static constexpr size_t WAVE_BYTES = ((size_t)XT_F4 * 16 + 2 * 16 * 4 + 15) & ~(size_t)15;
  static constexpr size_t FIXED = (size_t)M_F4 * 16 + (size_t)4 * CT * 16;
  static constexpr int NW = FIXED + 16 * WAVE_BYTES <= 160 * 1024 ? 16 : 8;   // waves per workgroup, 16 triples per wave tile
  static constexpr size_t LDS = FIXED + NW * WAVE_BYTES;
};

struct RArgs {
  const v4* E; uint32_t lde4;
  const float* R; int64_t ldr;
  const float* M; int64_t ldm;
  const int64_t *h, *t;
  int64_t n; int n_rel; int l1;
  float* score;
  void* ws;
};

template <typename G>
__global__ __launch_bounds__(G::NW * 64) void transr_fwd_mc_kernel(RArgs a) {
  constexpr int NW = G::NW;
  constexpr int NCH = G::NCH, D = G::D, KG = G::KG, CT = G::CT, KGF = G::KGF, J = G::J, TOTAL = G::TOTAL, PITCH4 = G::PITCH4;
  constexpr bool TAIL1 = G::TAIL1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* Ms = reinterpret_cast<v4*>(smem);                                   // [16 CT rows][PITCH4]
  v4* rS = Ms + G::M_F4;                                                  // [4 CT] relation vector, zero padded
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* wbase = reinterpret_cast<char*>(rS + 4 * CT) + (size_t)w * G::WAVE_BYTES;
  v4* xt = reinterpret_cast<v4*>(wbase);                                  // [16 * NCH] + 3 zero chunks
  int32_t* sid = reinterpret_cast<int32_t*>(xt + G::XT_F4);               // [2][16]  head / tail ids
  const Buckets b = carve(a.ws, a.n, a.n_rel);
  if (lane < 3) xt[16 * NCH + lane] = (v4){0.f, 0.f, 0.f, 0.f};
  int grow[J], gc[J];
#pragma unroll
  for (int jj = 0; jj < J; ++jj) {
    const int e = lane + 64 * jj;
    const bool past = e >= TOTAL;
    grow[jj] = past ? 0 : e / NCH;
    gc[jj] = past ? 0 : e % NCH;
  }
  const bool last_ok = lane + 64 * (J - 1) < TOTAL;
  const v4* xb = xt + j * NCH + kq;
  const v4* mrow = Ms + j * PITCH4 + kq;
  const bool l1 = a.l1 != 0;
  const int ntiles = b.tile_off[a.n_rel];
  int staged = -1;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // relation of this pass: the last rr with tile_off[rr] <= tile (binary search, wave uniform)
    int lo = 0, hi = a.n_rel - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (b.tile_off[mid] <= tile) lo = mid; else hi = mid - 1;
    }
    const int rr = lo;
    const int first = b.off[rr] + (tile - b.tile_off[rr]) * ST;
    const int count = min(ST, b.off[rr + 1] - first);
    if (rr != staged) {                                                    // stage M_r (rows >= d and columns >= d are zero) and r
      __syncthreads();
      const float* Mg = a.M + (int64_t)rr * a.ldm;
      float* Msf = reinterpret_cast<float*>(Ms);
      for (int idx = tid; idx < G::M_F4 * 4; idx += NW * 64) {
        const int row = idx / (PITCH4 * 4), k = idx - row * (PITCH4 * 4);
        Msf[idx] = (row < D && k < D) ? Mg[row * D + k] : 0.f;
      }
      float* rSf = reinterpret_cast<float*>(rS);
      for (int idx = tid; idx < 16 * CT; idx += NW * 64) rSf[idx] = (a.R && idx < D) ? a.R[(int64_t)rr * a.ldr + idx] : 0.f;   // R == NULL: no translation
      __syncthreads();
      staged = rr;
    }
    for (int sub = w; sub * 16 < count; sub += NW) {
      const int base = first + sub * 16;
      int my = -1;
      if (lane < 16) {
        if (sub * 16 + lane < count) my = b.perm[base + lane];
        const int src = my >= 0 ? my : b.perm[base];                       // tail lanes re-read the tile's first triple
        sid[lane] = (int32_t)a.h[src];
        sid[16 + lane] = a.t ? (int32_t)a.t[src] : 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      {                                                                    // q = h - t -> LDS tile
        v4 hh[J], tt[J];
#pragma unroll
        for (int jj = 0; jj < J; ++jj) {
          asm volatile("" : "+v"(gc[jj]));
          const uint32_t ih = (uint32_t)sid[grow[jj]], it = (uint32_t)sid[16 + grow[jj]];
          hh[jj] = a.E[(uint64_t)ih * a.lde4 + (uint32_t)gc[jj]];
          tt[jj] = a.t ? a.E[(uint64_t)it * a.lde4 + (uint32_t)gc[jj]] : (v4){0.f, 0.f, 0.f, 0.f};   // t == NULL: project h alone
        }
#pragma unroll
        for (int jj = 0; jj < J; ++jj) {
          if (jj < J - 1 || last_ok) xt[lane + 64 * jj] = hh[jj] + (-tt[jj]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // B operands of this wave's 16 triples, kept across the output tiles
      v4 bq[KGF];
#pragma unroll
      for (int g = 0; g < KGF; ++g) {
        bq[g] = xb[4 * g];
        if (4 * g + 3 >= NCH) {
          if (4 * g + kq >= NCH) bq[g] = (v4){0.f, 0.f, 0.f, 0.f};
        }
      }
      float btail = 0.f;
      if (TAIL1) btail = reinterpret_cast<const float*>(xt + j * NCH + 4 * KGF)[kq];
      v4 dacc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        // y^T tile: coordinates 16 ct + 4 kq + reg of triple j, starting from the relation vector
        v4 acc = rS[4 * ct + kq];
#pragma unroll
        for (int g = 0; g < KGF; ++g) {
          const v4 av = mrow[ct * 16 * PITCH4 + 4 * g];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bq[g][c], acc, 0, 0, 0);
        }
        if (TAIL1) {
          const float as = reinterpret_cast<const float*>(Ms + (ct * 16 + j) * PITCH4 + 4 * KGF)[kq];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(as, btail, acc, 0, 0, 0);
        }
        // rows >= d of M and entries >= d of r are zero, so padded coordinates contribute dist(0) = 0
        if (l1) dacc += __builtin_elementwise_abs(acc);
        else dacc = __builtin_elementwise_fma(acc, acc, dacc);
      }
      const float s = allsum_kq((dacc[0] + dacc[1]) + (dacc[2] + dacc[3]));
      // lane (kq, j): D rows are coordinates, columns triples -> every kq lane of column j holds the same total
      if (lane < 16 && my >= 0) a.score[my] = s;                            // lanes 0-15 = (kq 0, triple lane)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <typename G>
int launch(const RArgs& a, hipStream_t st, const char* name) {
  static_assert(G::LDS <= 160 * 1024, "LDS budget");
  (void)hipFuncSetAttribute((const void*)transr_fwd_mc_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
  const int64_t max_tiles = a.n / ST + a.n_rel;
  hipLaunchKernelGGL((transr_fwd_mc_kernel<G>), dim3(grid_for(max_tiles, 256)), dim3(G::NW * 64), G::LDS, st, a);
  return check_launch(name);
}

// ---------------------------------------------------------------------------------------------------------------- backward
// K4 backward on the matrix cores (autograd of transR.py:65-78 + utils/misc.py:21-26 in the forward's M (h - t) + r form):
//     gy = g * ddist(y),  gR[r] += gy,  gM[r] += gy (x) q,  gq = M_r^T gy,  gE[h] += gq,  gE[t] -= gq        (q = h - t)
// The generic kernel streams M_r per triple and adds d^2 atomics per triple to gM[r]: 20 ms at 307,200 triples against a
// 0.15 ms forward.  Here the batch is bucketed by relation like the forward and a workgroup walks consecutive 1024-triple passes:
//   phase 1 (per wave, 16 triples)  y^T = M_r q^T + r as in the forward; gy^T stays in the D registers;
//   phase 2 (per wave)              gq^T = M_r^T gy^T: the D registers of phase 1 ARE the B operands (k runs over coordinates
//                                   16 ct + 4 kq + reg, matched by A = a column walk of the staged M_r); the rows go out as
//                                   atomics (small batches) or plain stores for the segment reduction (large ones);
//   phase 3 (per workgroup round)   gM_r += gy^T (d x 16 NW) . q (16 NW x d): every wave leaves gy / q of its 16 triples in LDS,
//                                   and wave w owns the output row tiles ct_a = w, w + NW, ... over ALL the round's triples
//                                   (K = 16 NW), accumulating in registers across rounds and passes of the same relation; a
//                                   ones column gives gR.  One flush of d x d atomics per (workgroup, relation) run.
template <int NCH_>
struct RBGeom {
  static constexpr int NCH = NCH_, D = 4 * NCH;
  static constexpr int KG = (D + 15) / 16, CT = KG;
  static constexpr bool TAIL1 = NCH - 4 * (KG - 1) == 1;
  static constexpr int KGF = TAIL1 ? KG - 1 : KG;
  static constexpr int J = (16 * NCH + 63) / 64;
  static constexpr int TOTAL = 16 * NCH;
  static constexpr int PITCH4 = 4 * KG + 1, PITCHF = 4 * PITCH4;      // staged M_r rows: PITCHF = 20 (mod 32) floats at d = 100
  static constexpr int M_F4 = 16 * CT * PITCH4;
  static constexpr int GYP = (16 * CT) % 32 == 16 ? 16 * CT : 16 * CT + 16;   // float pitch of the q / gy tiles: 16 (mod 32)
  static constexpr int XP4 = GYP / 4;
  static constexpr size_t TILE_BYTES = (size_t)16 * GYP * 4;
  static constexpr size_t WAVE_BYTES = 2 * TILE_BYTES + 64 * 4;      // q tile, gy tile, ids (head, tail, triple) + upstream gradient
  static constexpr size_t FIXED = (size_t)M_F4 * 16 + (size_t)4 * CT * 16;
  static constexpr int NW = FIXED + CT * WAVE_BYTES <= 160 * 1024 ? CT : (CT + 2) / 2 + (FIXED + ((CT + 2) / 2 + 1) * WAVE_BYTES <= 160 * 1024 ? 1 : 0);
  static constexpr int S = (CT + NW - 1) / NW;                        // output row tiles per wave in phase 3
  static constexpr size_t LDS = FIXED + NW * WAVE_BYTES;
};

struct RBArgs {
  const v4* E; uint32_t lde4;
  const float* R; int64_t ldr;
  const float* M; int64_t ldm;
  const int64_t *h, *t;
  int64_t n; int n_rel; int l1;
  const float* gscore;
  float *gE, *gR, *gM;
  float* G;                      // != NULL: gq rows are stored here ([n][d]) instead of added to gE (segment reduction follows)
  void* ws;
};

template <typename G, bool ROWOUT>
__global__ __launch_bounds__(G::NW * 64) void transr_bwd_mc_kernel(RBArgs a) {
  constexpr int NW = G::NW, S = G::S;
  constexpr int NCH = G::NCH, D = G::D, CT = G::CT, KGF = G::KGF, J = G::J, TOTAL = G::TOTAL, PITCH4 = G::PITCH4, PITCHF = G::PITCHF;
  constexpr int GYP = G::GYP, XP4 = G::XP4;
  constexpr bool TAIL1 = G::TAIL1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* Ms = reinterpret_cast<v4*>(smem);                                   // [16 CT rows][PITCH4]
  const float* Msf = reinterpret_cast<const float*>(Ms);
  v4* rS = Ms + G::M_F4;                                                  // [4 CT] relation vector, zero padded
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* tiles = reinterpret_cast<char*>(rS + 4 * CT);
  auto qtile = [&](int ww) { return reinterpret_cast<v4*>(tiles + (size_t)ww * G::WAVE_BYTES); };
  auto gtile = [&](int ww) { return reinterpret_cast<v4*>(tiles + (size_t)ww * G::WAVE_BYTES + G::TILE_BYTES); };
  v4* xt = qtile(w);                                                      // [16 triples][XP4] q = h - t, zero beyond d
  v4* gyt = gtile(w);                                                     // [16 triples][XP4] gy
  int32_t* sid = reinterpret_cast<int32_t*>(tiles + (size_t)w * G::WAVE_BYTES + 2 * G::TILE_BYTES);   // [16] head, [16] tail, [16] triple
  float* gup = reinterpret_cast<float*>(sid + 48);                        // [16] upstream gradient (0 for tail slots)
  const Buckets b = carve(a.ws, a.n, a.n_rel);
  for (int idx = lane; idx < 2 * 16 * XP4; idx += 64) xt[idx] = (v4){0.f, 0.f, 0.f, 0.f};   // both tiles: finite from the start
  int grow[J], gc[J];
#pragma unroll
  for (int jj = 0; jj < J; ++jj) {
    const int e = lane + 64 * jj;
    const bool past = e >= TOTAL;
    grow[jj] = past ? 0 : e / NCH;
    gc[jj] = past ? 0 : e % NCH;
  }
  const bool last_ok = lane + 64 * (J - 1) < TOTAL;
  const v4* xb = xt + j * XP4 + kq;
  const v4* mrow = Ms + j * PITCH4 + kq;
  const bool l1 = a.l1 != 0;
  const int ntiles = b.tile_off[a.n_rel];
  const int t_lo = (int)((int64_t)blockIdx.x * ntiles / gridDim.x), t_hi = (int)((int64_t)(blockIdx.x + 1) * ntiles / gridDim.x);
  v4 accM[S][CT], accR[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    accR[s] = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < CT; ++cb) accM[s][cb] = (v4){0.f, 0.f, 0.f, 0.f};
  }
  const float one0 = j == 0 ? 1.f : 0.f;
  auto flush = [&](int rel) {                                             // this wave's row tiles of gM[rel], and gR[rel]
    float* gm = a.gM + (int64_t)rel * a.ldm;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int ca = w + s * NW;
      if (ca < CT) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int i = 16 * ca + 4 * kq + reg;
          if (i < D) {
#pragma unroll
            for (int cb = 0; cb < CT; ++cb) {
              const int k = 16 * cb + j;
              const float v = accM[s][cb][reg];
              if (k < D && v != 0.f) atomicAdd(gm + (int64_t)i * D + k, v);
            }
            if (j == 0 && accR[s][reg] != 0.f) atomicAdd(a.gR + (int64_t)rel * a.ldr + i, accR[s][reg]);
          }
        }
      }
      accR[s] = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cb = 0; cb < CT; ++cb) accM[s][cb] = (v4){0.f, 0.f, 0.f, 0.f};
    }
  };
  int staged = -1;
  for (int tile = t_lo; tile < t_hi; ++tile) {
    int lo = 0, hi = a.n_rel - 1;                                         // relation of this pass (wave uniform)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (b.tile_off[mid] <= tile) lo = mid; else hi = mid - 1;
    }
    const int rr = lo;
    const int first = b.off[rr] + (tile - b.tile_off[rr]) * ST;
    const int count = min(ST, b.off[rr + 1] - first);
    if (rr != staged) {
      if (staged >= 0) flush(staged);
      __syncthreads();
      const float* Mg = a.M + (int64_t)rr * a.ldm;
      float* Mw = reinterpret_cast<float*>(Ms);
      for (int idx = tid; idx < G::M_F4 * 4; idx += NW * 64) {
        const int row = idx / PITCHF, k = idx - row * PITCHF;
        Mw[idx] = (row < D && k < D) ? Mg[row * D + k] : 0.f;
      }
      float* rSf = reinterpret_cast<float*>(rS);
      for (int idx = tid; idx < 16 * CT; idx += NW * 64) rSf[idx] = idx < D ? a.R[(int64_t)rr * a.ldr + idx] : 0.f;
      __syncthreads();
      staged = rr;
    }
    for (int sub0 = 0; sub0 * 16 < count; sub0 += NW) {
      const int sub = sub0 + w;
      const bool valid = sub * 16 < count;                                // wave uniform
      if (valid) {
        const int base = first + sub * 16;
        if (lane < 16) {
          const bool on = sub * 16 + lane < count;
          const int my = on ? b.perm[base + lane] : -1;
          const int src = on ? my : b.perm[base];                         // tail slots re-read the tile's first triple, with g = 0
          sid[lane] = (int32_t)a.h[src];
          sid[16 + lane] = (int32_t)a.t[src];
          sid[32 + lane] = my;
          gup[lane] = on ? a.gscore[my] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {                                                                  // q = h - t -> LDS tile
          v4 hh[J], tt[J];
#pragma unroll
          for (int jj = 0; jj < J; ++jj) {
            asm volatile("" : "+v"(gc[jj]));
            const uint32_t ih = (uint32_t)sid[grow[jj]], it = (uint32_t)sid[16 + grow[jj]];
            hh[jj] = a.E[(uint64_t)ih * a.lde4 + (uint32_t)gc[jj]];
            tt[jj] = a.E[(uint64_t)it * a.lde4 + (uint32_t)gc[jj]];
          }
#pragma unroll
          for (int jj = 0; jj < J; ++jj) {
            if (jj < J - 1 || last_ok) xt[grow[jj] * XP4 + gc[jj]] = hh[jj] + (-tt[jj]);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 1: y^T tiles, then gy^T in the same registers (lane (kq, j): coordinates 16 ct + 4 kq + reg of triple j)
        v4 bq[KGF];
#pragma unroll
        for (int g = 0; g < KGF; ++g) bq[g] = xb[4 * g];                   // chunks beyond d are zero in the tile
        float btail = 0.f;
        if (TAIL1) btail = reinterpret_cast<const float*>(xt + j * XP4 + 4 * KGF)[kq];
        const float gj = gup[j];
        v4 gy[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          v4 acc = rS[4 * ct + kq];
#pragma unroll
          for (int g = 0; g < KGF; ++g) {
            const v4 av = mrow[ct * 16 * PITCH4 + 4 * g];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bq[g][c], acc, 0, 0, 0);
          }
          if (TAIL1) {
            const float as = reinterpret_cast<const float*>(Ms + (ct * 16 + j) * PITCH4 + 4 * KGF)[kq];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(as, btail, acc, 0, 0, 0);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = gj * ddist1(acc[c], l1);   // padded coordinates: y = 0 -> 0
          gy[ct] = acc;
          gyt[j * XP4 + 4 * ct + kq] = acc;
        }
        // ---- phase 2: gq^T = M^T gy^T; A = M[16 ct + 4 kq + reg][16 ct2 + j] (a column walk of the staged rows)
        const int32_t my_j = sid[32 + j];
        const int64_t hj = sid[j], tj = sid[16 + j];
#pragma unroll
        for (int ct2 = 0; ct2 < CT; ++ct2) {
          v4 gq = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const float am = Msf[(16 * ct + 4 * kq + reg) * PITCHF + 16 * ct2 + j];
              gq = __builtin_amdgcn_mfma_f32_16x16x4f32(am, gy[ct][reg], gq, 0, 0, 0);
            }
          }
          const int c0 = 16 * ct2 + 4 * kq;
          if (my_j >= 0 && c0 < D) {
            if constexpr (ROWOUT) {
              *reinterpret_cast<v4*>(a.G + (int64_t)my_j * D + c0) = gq;
            } else {
              atomic_add4(a.gE + hj * (int64_t)a.lde4 * 4 + c0, make_float4(gq[0], gq[1], gq[2], gq[3]));
              atomic_add4(a.gE + tj * (int64_t)a.lde4 * 4 + c0, make_float4(-gq[0], -gq[1], -gq[2], -gq[3]));
            }
          }
        }
      } else {
        for (int idx = lane; idx < 16 * XP4; idx += 64) gyt[idx] = (v4){0.f, 0.f, 0.f, 0.f};   // nothing from this wave in this round
      }
      __syncthreads();
      // ---- phase 3: gM row tiles of this wave over the round's NW x 16 triples; ones column -> gR
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int ca = w + s * NW;
        if (ca < CT) {
          for (int ww = 0; ww < NW; ++ww) {
            if ((sub0 + ww) * 16 >= count) break;                          // later waves had no triples either
            const float* gf = reinterpret_cast<const float*>(gtile(ww));
            const float* qf = reinterpret_cast<const float*>(qtile(ww));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const float av = gf[(4 * kk + kq) * GYP + 16 * ca + j];
              accR[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, one0, accR[s], 0, 0, 0);
#pragma unroll
              for (int cb = 0; cb < CT; ++cb) {
                const float bv = qf[(4 * kk + kq) * GYP + 16 * cb + j];
                accM[s][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accM[s][cb], 0, 0, 0);
              }
            }
          }
        }
      }
      __syncthreads();                                                     // the next round rewrites the tiles
    }
  }
  if (staged >= 0) flush(staged);
}

template <typename G>
int launch_bwd(const RBArgs& a, hipStream_t st, const char* name) {
  static_assert(G::LDS <= 160 * 1024, "LDS budget");
  const int64_t max_tiles = a.n / ST + a.n_rel;
  const dim3 grid(grid_for(max_tiles, 256)), block(G::NW * 64);
  if (a.G) {
    (void)hipFuncSetAttribute((const void*)transr_bwd_mc_kernel<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    hipLaunchKernelGGL((transr_bwd_mc_kernel<G, true>), grid, block, G::LDS, st, a);
  } else {
    (void)hipFuncSetAttribute((const void*)transr_bwd_mc_kernel<G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    hipLaunchKernelGGL((transr_bwd_mc_kernel<G, false>), grid, block, G::LDS, st, a);
  }
  return check_launch(name);
}

int bucket(const int64_t* r, int64_t n, int64_t n_rel, void* ws, hipStream_t st) {
  if (hipMemsetAsync(ws, 0, (size_t)2 * n_rel * sizeof(int32_t), st) != hipSuccess) return 1;   // cnt, cursor
  const int g1 = grid_for((n + 2047) / 2048, 512);
  hipLaunchKernelGGL(bucket_hist_kernel, dim3(g1), dim3(256), (size_t)n_rel * 4, st, r, n, (int)n_rel, ws);
  hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1), 0, st, n, (int)n_rel, ws);
  hipLaunchKernelGGL(bucket_scatter_kernel, dim3(g1), dim3(256), (size_t)n_rel * 8, st, r, n, (int)n_rel, ws);
  return 0;
}

}  // namespace

size_t transr_mc_workspace_bytes(int64_t n, int64_t n_rel) {
  return (size_t)(4 * n_rel + 2 + n) * sizeof(int32_t);
}

// Returns KTUP_OK / an error, or 1 when the shape is not covered (the caller runs the generic kernel).
int transr_fwd_mc(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm, int64_t n_rel, int d,
                  const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1, float* score, void* ws, hipStream_t st,
                  const char* name) {
  if (!ws || n_rel <= 0 || n_rel > 4096 || n >= (1ll << 31) || (d != 64 && d != 100 && d != 128)) return 1;
  if ((lde & 3) || !aligned16(E) || (lde >> 2) > 0xffffffffll) return 1;
  if (bucket(r, n, n_rel, ws, st)) return check_launch(name);
  RArgs a{reinterpret_cast<const v4*>(E), (uint32_t)(lde >> 2), R, ldr, M, ldm, h, t, n, (int)n_rel, l1, score, ws};
  if (d == 64) return launch<RGeom<16>>(a, st, name);
  if (d == 100) return launch<RGeom<25>>(a, st, name);
  return launch<RGeom<32>>(a, st, name);
}

}  // namespace ktup

namespace ktup {

// gE / gR / gM accumulate.  G != NULL: the entity-row gradients gq are stored to G ([n][d]; the caller reduces them by sorted
// segments: gE[h] += G, gE[t] -= G) instead of added with atomics.  Returns KTUP_OK / an error, or 1 when the shape is not covered.
int transr_bwd_mc(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm, int64_t n_rel, int d,
                  const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1, const float* gscore, float* gE, float* gR,
                  float* gM, float* G, void* ws, hipStream_t st, const char* name) {
  if (!ws || n_rel <= 0 || n_rel > 4096 || n >= (1ll << 31) || (d != 64 && d != 100 && d != 128) || ldm < (int64_t)d * d) return 1;
  if ((lde & 3) || !aligned16(E) || !aligned16(gE) || (lde >> 2) > 0xffffffffll || (G && !aligned16(G))) return 1;
  if (bucket(r, n, n_rel, ws, st)) return check_launch(name);
  RBArgs a{reinterpret_cast<const v4*>(E), (uint32_t)(lde >> 2), R, ldr, M, ldm, h, t, n, (int)n_rel, l1, gscore, gE, gR, gM, G, ws};
  if (d == 64) return launch_bwd<RBGeom<16>>(a, st, name);
  if (d == 100) return launch_bwd<RBGeom<25>>(a, st, name);
  return launch_bwd<RBGeom<32>>(a, st, name);
}

}  // namespace ktup
