// K1-K4: fused gather + score (and its backward) for BPRMF, TransE, TransH, TransR.
//
// Layout / mapping (gather-bound, HBM roofline): a row of d fp32 is d/4 16-byte chunks.  G = 16/32/64
// consecutive lanes of a wave own one scored row (d=64 -> 16 lanes, d=100 -> 25 of 32 lanes, d=256 -> 64),
// so every wave-level load is a run of contiguous 16-B pieces per row; the L1/L2 sum is reduced with
// xor-shuffles inside the lane group and one 4-byte score is stored per row.  The reference instead runs
// index_select x3 + ~4-10 elementwise kernels, each materialising a (B x d) temporary.
// When a pointer or pitch is not 16-B aligned (or d % 4 != 0) the same code runs with 4-byte lanes.
#include "ktup_rows.h"
#include "ktup_lane_swap.h"
#include "ktup_pref_geom.h"

using namespace ktup;

namespace {

// ------------------------------------------------------------------------------ K1 BPRMF
struct BprmfFwd {
  const float *U, *I; int64_t ldu, ldi; const int64_t *u, *i; float* score;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    V a[CPL], b[CPL];
    cx.load(a, U + u[row] * ldu);
    cx.load(b, I + i[row] * ldi);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s += vdot(a[j], b[j]);
    s = group_sum<G>(s);
    if (cx.lane == 0) score[row] = s;
  }
};
struct BprmfBwd {
  const float *U, *I; int64_t ldu, ldi; const int64_t *u, *i; const float* gs; float *gU, *gI;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t ur = u[row], ir = i[row];
    V a[CPL], b[CPL];
    cx.load(a, U + ur * ldu);
    cx.load(b, I + ir * ldi);
    const float g = gs[row];
#pragma unroll
    for (int j = 0; j < CPL; ++j) { V t = a[j]; a[j] = vscale(g, b[j]); b[j] = vscale(g, t); }
    cx.scatter_add(gU + ur * ldu, a);
    cx.scatter_add(gI + ir * ldi, b);
  }
};

// ------------------------------------------------------------------------------ K2 TransE
struct TranseFwd {
  const float *E, *R; int64_t lde, ldr; const int64_t *h, *t, *r; bool l1; float* score;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    V a[CPL], b[CPL], c[CPL];
    cx.load(a, E + h[row] * lde);
    cx.load(b, E + t[row] * lde);
    cx.load(c, R + r[row] * ldr);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s += vdist(vsub(vadd(a[j], c[j]), b[j]), l1);  // (h + r) - t, transE.py:57
    s = group_sum<G>(s);
    if (cx.lane == 0) score[row] = s;
  }
};
struct TranseBwd {
  const float *E, *R; int64_t lde, ldr; const int64_t *h, *t, *r; bool l1; const float* gs; float *gE, *gR;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t hr = h[row], tr = t[row], rr = r[row];
    V a[CPL], b[CPL], c[CPL];
    cx.load(a, E + hr * lde);
    cx.load(b, E + tr * lde);
    cx.load(c, R + rr * ldr);
    const float g = gs[row];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      a[j] = vscale(g, vddist(vsub(vadd(a[j], c[j]), b[j]), l1));  // gz
      b[j] = vscale(-1.f, a[j]);
    }
    cx.scatter_add(gE + hr * lde, a);
    cx.scatter_add(gE + tr * lde, b);
    cx.scatter_add(gR + rr * ldr, a);
  }
};

// ------------------------------------------------------------------------------ K3 TransH
// z = (h - (h.w)w) + r - (t - (t.w)w),  w = norm row of the relation (NOT re-normalised, misc.py:18-19)
struct TranshFwd {
  const float *E, *R, *Nm; int64_t lde, ldr, ldn; const int64_t *h, *t, *r; bool l1; float* score;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t rr = r[row];
    V a[CPL], b[CPL], c[CPL], w[CPL];
    cx.load(a, E + h[row] * lde);
    cx.load(b, E + t[row] * lde);
    cx.load(c, R + rr * ldr);
    cx.load(w, Nm + rr * ldn);
    float dh = 0.f, dt = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { dh += vdot(a[j], w[j]); dt += vdot(b[j], w[j]); }
    dh = group_sum<G>(dh);
    dt = group_sum<G>(dt);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const V ph = vfma(-dh, w[j], a[j]), pt = vfma(-dt, w[j], b[j]);
      s += vdist(vsub(vadd(ph, c[j]), pt), l1);
    }
    s = group_sum<G>(s);
    if (cx.lane == 0) score[row] = s;
  }
};
// K3 forward with both relation tables staged in LDS (n_rel * d small: 16 KB at ml1m).  The generic row kernel gathers
// 4 rows per triple through the vector-memory path, which is what bounds it (tools/gather_bench.hip: ~16-20 B/clk/CU of
// scattered 400-byte rows); r and w come from LDS here, so only h and t (the algorithmic bytes) are gathered.
template <int G>
__global__ __launch_bounds__(256) void transh_fwd_lds_kernel(TranshFwd op, int nch, int n_rel, int64_t n) {
  extern __shared__ float4 ktup_transh_tabs[];
  float4* Rs = ktup_transh_tabs;                 // [n_rel][nch]
  float4* Ws = ktup_transh_tabs + n_rel * nch;
  for (int idx = threadIdx.x; idx < n_rel * nch; idx += 256) {
    const int row = idx / nch, c = idx - row * nch;
    Rs[idx] = reinterpret_cast<const float4*>(op.R + (int64_t)row * op.ldr)[c];
    Ws[idx] = reinterpret_cast<const float4*>(op.Nm + (int64_t)row * op.ldn)[c];
  }
  __syncthreads();
  const int lane = threadIdx.x % G;
  constexpr int GPB = 256 / G;
  const bool on = lane < nch;
  for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / G; row < n; row += (int64_t)gridDim.x * GPB) {
    const int rr = (int)op.r[row];
    float4 a = f4zero(), b = f4zero(), c = f4zero(), w = f4zero();
    if (on) {
      a = reinterpret_cast<const float4*>(op.E + op.h[row] * op.lde)[lane];
      b = reinterpret_cast<const float4*>(op.E + op.t[row] * op.lde)[lane];
      c = Rs[rr * nch + lane];
      w = Ws[rr * nch + lane];
    }
    const float dh = group_sum<G>(dot4(a, w)), dt = group_sum<G>(dot4(b, w));
    const float4 ph = fma4(-dh, w, a), pt = fma4(-dt, w, b);
    const float s = group_sum<G>(dist4((ph + c) - pt, op.l1));
    if (lane == 0) op.score[row] = s;
  }
}

// K3 forward, wave-tile form for large batches: a wave gathers the h and t rows of 16 triples with the linear (row, chunk)
// lane mapping of pref_fwd_mc (every lane busy, 2 x J float4 loads in flight per lane; the lane-group kernels above keep
// 25 of 32 lanes busy at d = 100 and 2 loads in flight), puts q = h - t into an LDS tile, and lane (kq, triple) then walks
// chunks kq, kq + 4, ... of its triple against the LDS-resident relation rows.
template <int NCH, bool TRANSH>
__global__ __launch_bounds__(1024) void transh_fwd_tile_kernel(TranshFwd op, int n_rel, int64_t n) {   // TRANSH false: TransE (op.Nm unused)
  typedef float v4 __attribute__((ext_vector_type(4)));
  constexpr int J = (16 * NCH + 63) / 64, TOTAL = 16 * NCH, P4 = NCH | 1, NW = 16;
  constexpr int CPL = (NCH + 3) / 4;                                  // chunks per lane in the compute mapping
  extern __shared__ __attribute__((aligned(16))) char ktup_transh_tile_smem[];
  v4* Rs = reinterpret_cast<v4*>(ktup_transh_tile_smem);              // [n_rel][P4]
  v4* Ws = Rs + n_rel * P4;
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  v4* QT = Ws + n_rel * P4 + w * (16 * P4 + 12);                      // per wave: q = h - t tile [16][P4], ids [3][16]
  int32_t* sid = reinterpret_cast<int32_t*>(QT + 16 * P4);
  for (int idx = tid; idx < n_rel * NCH; idx += NW * 64) {
    const int row = idx / NCH, c = idx - row * NCH;
    Rs[row * P4 + c] = *reinterpret_cast<const v4*>(op.R + (int64_t)row * op.ldr + 4 * c);
    if (TRANSH) Ws[row * P4 + c] = *reinterpret_cast<const v4*>(op.Nm + (int64_t)row * op.ldn + 4 * c);
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
  const v4* E4 = reinterpret_cast<const v4*>(op.E);
  const uint32_t lde4 = (uint32_t)(op.lde >> 2);
  const int64_t ntiles = (n + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * NW + w; tile < ntiles; tile += (int64_t)gridDim.x * NW) {
    const int64_t row0 = tile * 16;
    if (lane < 16) {
      const int64_t gr = min(row0 + lane, n - 1);
      sid[lane] = (int32_t)op.h[gr]; sid[16 + lane] = (int32_t)op.t[gr]; sid[32 + lane] = (int32_t)op.r[gr];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      v4 hh[J], tt[J];
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        asm volatile("" : "+v"(gc[jj]));
        hh[jj] = E4[(uint64_t)(uint32_t)sid[grow[jj]] * lde4 + (uint32_t)gc[jj]];
        tt[jj] = E4[(uint64_t)(uint32_t)sid[16 + grow[jj]] * lde4 + (uint32_t)gc[jj]];
      }
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        if (jj < J - 1 || last_ok) QT[grow[jj] * P4 + gc[jj]] = hh[jj] + (-tt[jj]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // z = (h - (h.w) w) + r - (t - (t.w) w) = q + r - (q.w) w with q = h - t (the projection is linear)
    const int rr = sid[32 + j];
    const v4* qrow = QT + j * P4;
    const v4* wrow = Ws + rr * P4;
    const v4* rrow = Rs + rr * P4;
    v4 qv[CPL], wv[CPL];
    v4 da = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = kq + 4 * k;
      qv[k] = da * 0.f; wv[k] = qv[k];
      if (c < NCH) { qv[k] = qrow[c]; if (TRANSH) { wv[k] = wrow[c]; da += qv[k] * wv[k]; } }
    }
    const float sq = TRANSH ? ktup::allsum_kq((da[0] + da[1]) + (da[2] + da[3])) : 0.f;
    v4 acc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = kq + 4 * k;
      if (c < NCH) {
        const v4 z = TRANSH ? (qv[k] + rrow[c]) - sq * wv[k] : qv[k] + rrow[c];
        if (op.l1) acc += __builtin_elementwise_abs(z);
        else acc += z * z;
      }
    }
    const float s = ktup::allsum_kq((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane < 16 && row0 + lane < n) op.score[row0 + lane] = s;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}


// K1 forward, wave-tile form for large batches (same reasoning as transh_fwd_tile_kernel): the u (.) i products of 16 pairs go
// through an LDS tile and are row-summed by lane (kq, pair).
template <int NCH>
__global__ __launch_bounds__(1024) void bprmf_fwd_tile_kernel(BprmfFwd op, int64_t n) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  constexpr int J = (16 * NCH + 63) / 64, TOTAL = 16 * NCH, P4 = NCH | 1, NW = 16, CPL = (NCH + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char ktup_bprmf_tile_smem[];
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  v4* PT = reinterpret_cast<v4*>(ktup_bprmf_tile_smem) + w * (16 * P4 + 8);
  int32_t* sid = reinterpret_cast<int32_t*>(PT + 16 * P4);            // [2][16]
  int grow[J], gc[J];
#pragma unroll
  for (int jj = 0; jj < J; ++jj) {
    const int e = lane + 64 * jj;
    const bool past = e >= TOTAL;
    grow[jj] = past ? 0 : e / NCH;
    gc[jj] = past ? 0 : e % NCH;
  }
  const bool last_ok = lane + 64 * (J - 1) < TOTAL;
  const v4* U4 = reinterpret_cast<const v4*>(op.U);
  const v4* I4 = reinterpret_cast<const v4*>(op.I);
  const uint32_t ldu4 = (uint32_t)(op.ldu >> 2), ldi4 = (uint32_t)(op.ldi >> 2);
  const int64_t ntiles = (n + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * NW + w; tile < ntiles; tile += (int64_t)gridDim.x * NW) {
    const int64_t row0 = tile * 16;
    if (lane < 16) {
      const int64_t gr = min(row0 + lane, n - 1);
      sid[lane] = (int32_t)op.u[gr]; sid[16 + lane] = (int32_t)op.i[gr];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      v4 uu[J], ii[J];
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        asm volatile("" : "+v"(gc[jj]));
        uu[jj] = U4[(uint64_t)(uint32_t)sid[grow[jj]] * ldu4 + (uint32_t)gc[jj]];
        ii[jj] = I4[(uint64_t)(uint32_t)sid[16 + grow[jj]] * ldi4 + (uint32_t)gc[jj]];
      }
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        if (jj < J - 1 || last_ok) PT[grow[jj] * P4 + gc[jj]] = uu[jj] * ii[jj];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    v4 acc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = kq + 4 * k;
      if (c < NCH) acc += PT[j * P4 + c];
    }
    const float s = ktup::allsum_kq((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane < 16 && row0 + lane < n) op.score[row0 + lane] = s;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int NCH>
int launch_bprmf_tile(const BprmfFwd& op, int64_t n, hipStream_t st) {
  const size_t lds = (size_t)16 * (16 * (NCH | 1) + 8) * 16;
  (void)hipFuncSetAttribute((const void*)bprmf_fwd_tile_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(bprmf_fwd_tile_kernel<NCH>, dim3(grid_for(((n + 15) / 16 + 15) / 16, 256)), dim3(1024), lds, st, op, n);
  return check_launch("ktup_score_bprmf_fwd");
}


inline bool tile_forward_applies(int64_t n, int64_t n_rel, int d, int64_t lde) {      // large batches, instantiated sizes
  return n >= 65536 && n_rel > 0 && (d == 64 || d == 100 || d == 128) && (lde >> 2) <= 0xffffffffll;
}

// Returns 1 when the relation tables do not fit LDS next to the 16 wave tiles.
template <bool TRANSH>
int launch_tile_forward(const TranshFwd& op, int d, int64_t n_rel, int64_t n, hipStream_t st, const char* name) {
  const int nch = d / 4, p4 = nch | 1;
  const size_t lds = ((size_t)2 * n_rel * p4 + (size_t)16 * (16 * p4 + 12)) * 16;
  if (lds > 160 * 1024) return 1;
  const int grid = grid_for(((n + 15) / 16 + 15) / 16, 256);
#define KTUP_TILE(NCH)                                                                                                              \
  {                                                                                                                                 \
    (void)hipFuncSetAttribute((const void*)transh_fwd_tile_kernel<NCH, TRANSH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((transh_fwd_tile_kernel<NCH, TRANSH>), dim3(grid), dim3(1024), lds, st, op, (int)n_rel, n);                    \
  }
  if (d == 64) KTUP_TILE(16) else if (d == 100) KTUP_TILE(25) else KTUP_TILE(32)
#undef KTUP_TILE
  return check_launch(name);
}

// With q = h - t, s = q.w, a = gz.w :  gh = gz - a w, gt = -gh, gr = gz, gw = -s gz - a q.
struct TranshBwd {
  const float *E, *R, *Nm; int64_t lde, ldr, ldn; const int64_t *h, *t, *r; bool l1; const float* gs;
  float *gE, *gR, *gN;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t hr = h[row], tr = t[row], rr = r[row];
    V a[CPL], b[CPL], c[CPL], w[CPL];
    cx.load(a, E + hr * lde);
    cx.load(b, E + tr * lde);
    cx.load(c, R + rr * ldr);
    cx.load(w, Nm + rr * ldn);
    float dh = 0.f, dt = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { dh += vdot(a[j], w[j]); dt += vdot(b[j], w[j]); }
    dh = group_sum<G>(dh);
    dt = group_sum<G>(dt);
    const float g = gs[row];
    float aw = 0.f;
    V gz[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const V ph = vfma(-dh, w[j], a[j]), pt = vfma(-dt, w[j], b[j]);
      gz[j] = vscale(g, vddist(vsub(vadd(ph, c[j]), pt), l1));
      aw += vdot(gz[j], w[j]);
    }
    aw = group_sum<G>(aw);
    const float sq = dh - dt;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const V q = vsub(a[j], b[j]);
      a[j] = vfma(-aw, w[j], gz[j]);                         // gh
      b[j] = vscale(-1.f, a[j]);                             // gt
      w[j] = vfma(-aw, q, vscale(-sq, gz[j]));               // gw
    }
    cx.scatter_add(gE + hr * lde, a);
    cx.scatter_add(gE + tr * lde, b);
    cx.scatter_add(gR + rr * ldr, gz);
    cx.scatter_add(gN + rr * ldn, w);
  }
};

// ------------------------------------------------------------------------------ K4 TransR
// One wave per triple.  q = h - t is staged in LDS; lane j owns output coordinate j of M_r q (+ r_j).
// (Computes M(h - t) where the reference computes Mh - Mt, transR.py:71-72: same value up to fp32 rounding.)
// First version: the d x d matrix of the triple's relation is streamed from L2 per triple; bucketing triples
// by relation so a workgroup keeps M_r in LDS is the planned optimisation (only R distinct matrices exist).
template <bool BWD>
__global__ __launch_bounds__(256) void transr_kernel(const float* __restrict__ E, int64_t lde, const float* __restrict__ R,
                                                     int64_t ldr, const float* __restrict__ M, int64_t ldm, int d,
                                                     const int64_t* __restrict__ h, const int64_t* __restrict__ t,
                                                     const int64_t* __restrict__ r, int64_t n, bool l1,
                                                     float* __restrict__ score, const float* __restrict__ gs,
                                                     float* gE, float* gR, float* gM) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* q = smem + wv * 2 * d;  // q[d] then gz[d] (backward)
  float* gzs = q + d;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < n; row += (int64_t)gridDim.x * 4) {
    const int64_t hr = h[row], tr = t[row], rr = r[row];
    for (int k = lane; k < d; k += 64) q[k] = E[hr * lde + k] - E[tr * lde + k];
    __builtin_amdgcn_wave_barrier();
    const float* Mr = M + rr * ldm;
    float part = 0.f;
    const float g = BWD ? gs[row] : 0.f;
    for (int j = lane; j < d; j += 64) {
      const float* mrow = Mr + (int64_t)j * d;
      float acc = 0.f;
      for (int k = 0; k < d; ++k) acc = fmaf(mrow[k], q[k], acc);
      const float z = acc + R[rr * ldr + j];
      if (BWD) {
        const float gz = g * ddist1(z, l1);
        gzs[j] = gz;
        atomicAdd(gR + rr * ldr + j, gz);
      } else {
        part += dist1(z, l1);
      }
    }
    if (!BWD) {
      part = group_sum<64>(part);
      if (lane == 0) score[row] = part;
    } else {
      __builtin_amdgcn_wave_barrier();
      // lane k: gq_k = sum_j M[j][k] gz_j (coalesced over k); gM[j][k] += gz_j q_k
      float* gMr = gM + rr * ldm;
      for (int k = lane; k < d; k += 64) {
        float gq = 0.f;
        const float qk = q[k];
        for (int j = 0; j < d; ++j) {
          const float gz = gzs[j];
          gq = fmaf(Mr[(int64_t)j * d + k], gz, gq);
          atomicAdd(gMr + (int64_t)j * d + k, gz * qk);
        }
        atomicAdd(gE + hr * lde + k, gq);
        atomicAdd(gE + tr * lde + k, -gq);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int check_common(const char* name, int d, int64_t n) {
  if (d <= 0) return set_error(KTUP_ERR_INVALID_ARG, "%s: embedding_size must be positive (got %d)", name, d);
  if (n < 0) return set_error(KTUP_ERR_INVALID_ARG, "%s: negative row count", name);
  return KTUP_OK;
}

}  // namespace

#define KTUP_NONNULL(name, p) KTUP_REQUIRE((p) != nullptr || n == 0, name ": null pointer argument '" #p "'")

extern "C" int ktup_score_bprmf_fwd(const float* U, int64_t ldu, const float* I, int64_t ldi, int d, const int64_t* u_ids,
                                    const int64_t* i_ids, int64_t n, float* score, void* stream) {
  if (int e = check_common("ktup_score_bprmf_fwd", d, n)) return e;
  KTUP_NONNULL("ktup_score_bprmf_fwd", U); KTUP_NONNULL("ktup_score_bprmf_fwd", I);
  KTUP_NONNULL("ktup_score_bprmf_fwd", u_ids); KTUP_NONNULL("ktup_score_bprmf_fwd", i_ids);
  KTUP_NONNULL("ktup_score_bprmf_fwd", score);
  BprmfFwd op{U, I, ldu, ldi, u_ids, i_ids, score};
  const bool v4ok = can_vec4(d, {U, I}, {ldu, ldi});
  if (v4ok && n >= 65536 && (ldu >> 2) <= 0xffffffffll && (ldi >> 2) <= 0xffffffffll) {      // large batches: wave tiles
    if (d == 64) return launch_bprmf_tile<16>(op, n, (hipStream_t)stream);
    if (d == 100) return launch_bprmf_tile<25>(op, n, (hipStream_t)stream);
    if (d == 128) return launch_bprmf_tile<32>(op, n, (hipStream_t)stream);
  }
  return launch_rows(op, d, v4ok, n, (hipStream_t)stream, "ktup_score_bprmf_fwd");
}

extern "C" int ktup_score_bprmf_bwd(const float* U, int64_t ldu, const float* I, int64_t ldi, int d, const int64_t* u_ids,
                                    const int64_t* i_ids, int64_t n, const float* gscore, float* gU, float* gI,
                                    void* stream) {
  if (int e = check_common("ktup_score_bprmf_bwd", d, n)) return e;
  KTUP_NONNULL("ktup_score_bprmf_bwd", U); KTUP_NONNULL("ktup_score_bprmf_bwd", I);
  KTUP_NONNULL("ktup_score_bprmf_bwd", gscore); KTUP_NONNULL("ktup_score_bprmf_bwd", gU);
  KTUP_NONNULL("ktup_score_bprmf_bwd", gI);
  BprmfBwd op{U, I, ldu, ldi, u_ids, i_ids, gscore, gU, gI};
  return launch_rows(op, d, can_vec4(d, {U, I, gU, gI}, {ldu, ldi}), n, (hipStream_t)stream, "ktup_score_bprmf_bwd");
}

extern "C" int ktup_score_transe_fwd(const float* E, int64_t lde, const float* R, int64_t ldr, int64_t n_rel, int d,
                                     const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1, float* score,
                                     void* stream) {
  if (int e = check_common("ktup_score_transe_fwd", d, n)) return e;
  KTUP_NONNULL("ktup_score_transe_fwd", E); KTUP_NONNULL("ktup_score_transe_fwd", R);
  KTUP_NONNULL("ktup_score_transe_fwd", h); KTUP_NONNULL("ktup_score_transe_fwd", t);
  KTUP_NONNULL("ktup_score_transe_fwd", r); KTUP_NONNULL("ktup_score_transe_fwd", score);
  TranseFwd op{E, R, lde, ldr, h, t, r, l1 != 0, score};
  const bool v4ok = can_vec4(d, {E, R}, {lde, ldr});
  if (v4ok && tile_forward_applies(n, n_rel, d, lde)) {
    const TranshFwd top{E, R, nullptr, lde, ldr, 0, h, t, r, l1 != 0, score};
    const int rc = launch_tile_forward<false>(top, d, n_rel, n, (hipStream_t)stream, "ktup_score_transe_fwd");
    if (rc != 1) return rc;
  }
  return launch_rows(op, d, v4ok, n, (hipStream_t)stream, "ktup_score_transe_fwd");
}

extern "C" int ktup_score_transe_bwd(const float* E, int64_t lde, const float* R, int64_t ldr, int d, const int64_t* h,
                                     const int64_t* t, const int64_t* r, int64_t n, int l1, const float* gscore,
                                     float* gE, float* gR, void* stream) {
  if (int e = check_common("ktup_score_transe_bwd", d, n)) return e;
  KTUP_NONNULL("ktup_score_transe_bwd", E); KTUP_NONNULL("ktup_score_transe_bwd", R);
  KTUP_NONNULL("ktup_score_transe_bwd", gscore); KTUP_NONNULL("ktup_score_transe_bwd", gE);
  KTUP_NONNULL("ktup_score_transe_bwd", gR);
  TranseBwd op{E, R, lde, ldr, h, t, r, l1 != 0, gscore, gE, gR};
  return launch_rows(op, d, can_vec4(d, {E, R, gE, gR}, {lde, ldr}), n, (hipStream_t)stream, "ktup_score_transe_bwd");
}

extern "C" int ktup_score_transh_fwd(const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                                     int64_t n_rel, int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n,
                                     int l1, float* score, void* stream) {
  if (int e = check_common("ktup_score_transh_fwd", d, n)) return e;
  KTUP_NONNULL("ktup_score_transh_fwd", E); KTUP_NONNULL("ktup_score_transh_fwd", R);
  KTUP_NONNULL("ktup_score_transh_fwd", Nrm); KTUP_NONNULL("ktup_score_transh_fwd", h);
  KTUP_NONNULL("ktup_score_transh_fwd", t); KTUP_NONNULL("ktup_score_transh_fwd", r);
  KTUP_NONNULL("ktup_score_transh_fwd", score);
  TranshFwd op{E, R, Nrm, lde, ldr, ldn, h, t, r, l1 != 0, score};
  const bool v4ok = can_vec4(d, {E, R, Nrm}, {lde, ldr, ldn});
  const int64_t tab_bytes = 2 * n_rel * (int64_t)d * 4;
  if (v4ok && tile_forward_applies(n, n_rel, d, lde)) {
    const int rc = launch_tile_forward<true>(op, d, n_rel, n, (hipStream_t)stream, "ktup_score_transh_fwd");
    if (rc != 1) return rc;
  }
  if (n > 0 && v4ok && d <= 256 && n_rel > 0 && tab_bytes <= 16 * 1024 && n >= 4 * n_rel) {   // 8 workgroups per CU keep their tables
    const int nch = d / 4;
    const int G = nch <= 16 ? 16 : nch <= 32 ? 32 : 64;
    const int grid = grid_for((n + (256 / G) - 1) / (256 / G));
    hipStream_t st = (hipStream_t)stream;
    if (G == 16) hipLaunchKernelGGL(transh_fwd_lds_kernel<16>, dim3(grid), dim3(256), tab_bytes, st, op, nch, (int)n_rel, n);
    else if (G == 32) hipLaunchKernelGGL(transh_fwd_lds_kernel<32>, dim3(grid), dim3(256), tab_bytes, st, op, nch, (int)n_rel, n);
    else hipLaunchKernelGGL(transh_fwd_lds_kernel<64>, dim3(grid), dim3(256), tab_bytes, st, op, nch, (int)n_rel, n);
    return check_launch("ktup_score_transh_fwd");
  }
  return launch_rows(op, d, v4ok, n, (hipStream_t)stream, "ktup_score_transh_fwd");
}

extern "C" int ktup_score_transh_bwd(const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                                     int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                                     const float* gscore, float* gE, float* gR, float* gN, void* stream) {
  if (int e = check_common("ktup_score_transh_bwd", d, n)) return e;
  KTUP_NONNULL("ktup_score_transh_bwd", E); KTUP_NONNULL("ktup_score_transh_bwd", R);
  KTUP_NONNULL("ktup_score_transh_bwd", Nrm); KTUP_NONNULL("ktup_score_transh_bwd", gscore);
  KTUP_NONNULL("ktup_score_transh_bwd", gE); KTUP_NONNULL("ktup_score_transh_bwd", gR);
  KTUP_NONNULL("ktup_score_transh_bwd", gN);
  TranshBwd op{E, R, Nrm, lde, ldr, ldn, h, t, r, l1 != 0, gscore, gE, gR, gN};
  return launch_rows(op, d, can_vec4(d, {E, R, Nrm, gE, gR, gN}, {lde, ldr, ldn}), n, (hipStream_t)stream,
                     "ktup_score_transh_bwd");
}

// ---------------------------------------------------------------------------------------------------------------------------
// K1-K3 backward for large batches: per-row gradient vectors by plain stores + reduction by sorted segments (ktup_segreduce.hip)
// instead of d float atomics per gathered row, which serialise at ~40 G atomics/s when hundreds of rows share a table row
// (round 1: 2.8 ms for 307,200 triples against a 22 us forward).  The relation-side tables (a few rows that EVERY triple hits)
// accumulate in LDS per workgroup and reach memory with one atomic per element and workgroup.
//   TransE : G[k] = gz            gE[h] += G, gE[t] -= G (one sort over h ++ t with signs), gR via LDS
//   TransH : G[k] = gh (gt = -gh) gE likewise;  gR (= gz) and gN (= gw) via LDS
//   BPRMF  : GU[k] = g v, GI[k] = g u   two sorts
struct KgSegArgs {
  const float *E, *R, *Nm; int64_t lde, ldr, ldn;
  const int64_t *h, *t, *r; int64_t n; int nch, d, n_rel; bool l1;
  const float* gs; float* G; float *gR, *gN;
};

// PRIV: every lane group accumulates the relation-side gradients in its OWN LDS copy with float4 read-modify-writes (a wave's LDS
// operations execute in order, and the lanes of a group own distinct addresses).  Shared accumulators with ds_add_f32 ran at
// 0.3 lane-atomics per clock and CU -- the two lane groups of a wave usually hit the same relation row -- and were 150 of the
// kernel's 168 us at 307,200 triples; they remain for relation tables too large for 256 / GL copies.
template <int GL, bool TRANSH, bool PRIV>
__global__ __launch_bounds__(256) void kg_bwd_rowout_kernel(KgSegArgs a) {
  extern __shared__ __attribute__((aligned(16))) float kacc[];    // [PRIV ? 256 / GL : 1][(TRANSH ? 2 : 1)][n_rel * d]
  const int relems = a.n_rel * a.d;
  constexpr int GPB = 256 / GL;
  constexpr int NT = TRANSH ? 2 : 1;
  for (int i = threadIdx.x; i < (PRIV ? GPB : 1) * NT * relems; i += 256) kacc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x % GL;
  float* mine = kacc + (PRIV ? (threadIdx.x / GL) * NT * relems : 0);
  const bool on = lane < a.nch;
  // two triples per lane group and trip: both id loads, then all eight row loads, are in flight together (one triple per trip
  // left the kernel waiting on two dependent memory round trips per 1.6 KB moved)
  const int64_t stride = (int64_t)gridDim.x * GPB;
  for (int64_t k0 = (int64_t)blockIdx.x * GPB + threadIdx.x / GL; k0 < a.n; k0 += 2 * stride) {
    int64_t kk[2], rr[2], hi[2], ti[2];
    bool live[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      kk[x] = k0 + x * stride;
      live[x] = kk[x] < a.n;
      const int64_t kc = live[x] ? kk[x] : k0;
      rr[x] = a.r[kc]; hi[x] = a.h[kc]; ti[x] = a.t[kc];
    }
    float4 hh[2], tt[2], c[2], w[2];
    float g[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      hh[x] = tt[x] = c[x] = w[x] = f4zero();
      if (on) {
        hh[x] = reinterpret_cast<const float4*>(a.E + hi[x] * a.lde)[lane];
        tt[x] = reinterpret_cast<const float4*>(a.E + ti[x] * a.lde)[lane];
        c[x] = reinterpret_cast<const float4*>(a.R + rr[x] * a.ldr)[lane];
        if (TRANSH) w[x] = reinterpret_cast<const float4*>(a.Nm + rr[x] * a.ldn)[lane];
      }
      g[x] = a.gs[live[x] ? kk[x] : k0];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      float4 gz, gh, gw = f4zero();
      if (TRANSH) {                                                  // the arithmetic of TranshBwd
        const float dh = group_sum<GL>(dot4(hh[x], w[x])), dt = group_sum<GL>(dot4(tt[x], w[x]));
        const float4 ph = fma4(-dh, w[x], hh[x]), pt = fma4(-dt, w[x], tt[x]);
        gz = g[x] * ddist4((ph + c[x]) - pt, a.l1);
        const float aw = group_sum<GL>(dot4(gz, w[x]));
        gh = fma4(-aw, w[x], gz);
        gw = fma4(-aw, hh[x] - tt[x], (-(dh - dt)) * gz);
      } else {
        gz = g[x] * ddist4((hh[x] + c[x]) - tt[x], a.l1);
        gh = gz;
      }
      if (on && live[x]) {
        reinterpret_cast<float4*>(a.G + kk[x] * a.d)[lane] = gh;
        float* r0 = mine + rr[x] * a.d + 4 * lane;
        if (PRIV) {
          float4* q0 = reinterpret_cast<float4*>(r0);
          *q0 = *q0 + gz;
          if (TRANSH) {
            float4* q1 = reinterpret_cast<float4*>(r0 + relems);
            *q1 = *q1 + gw;
          }
        } else {
          atomicAdd(r0 + 0, gz.x); atomicAdd(r0 + 1, gz.y); atomicAdd(r0 + 2, gz.z); atomicAdd(r0 + 3, gz.w);
          if (TRANSH) {
            float* w0 = r0 + relems;
            atomicAdd(w0 + 0, gw.x); atomicAdd(w0 + 1, gw.y); atomicAdd(w0 + 2, gw.z); atomicAdd(w0 + 3, gw.w);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < relems; i += 256) {
    const int row = i / a.d, col = i - row * a.d;
    float vr = 0.f, vn = 0.f;
#pragma unroll
    for (int g = 0; g < (PRIV ? GPB : 1); ++g) {
      vr += kacc[g * NT * relems + i];
      if (TRANSH) vn += kacc[g * NT * relems + relems + i];
    }
    if (vr != 0.f) atomicAdd(a.gR + (int64_t)row * a.ldr + col, vr);
    if (TRANSH && vn != 0.f) atomicAdd(a.gN + (int64_t)row * a.ldn + col, vn);
  }
}

struct BprmfSegArgs {
  const float *U, *I; int64_t ldu, ldi; const int64_t *u, *i; int64_t n; int nch, d; const float* gs; float *GU, *GI;
};
template <int GL>
__global__ __launch_bounds__(256) void bprmf_bwd_rowout_kernel(BprmfSegArgs a) {
  constexpr int GPB = 256 / GL;
  const int lane = threadIdx.x % GL;
  if (lane >= a.nch) return;
  for (int64_t k = (int64_t)blockIdx.x * GPB + threadIdx.x / GL; k < a.n; k += (int64_t)gridDim.x * GPB) {
    const float4 uu = reinterpret_cast<const float4*>(a.U + a.u[k] * a.ldu)[lane];
    const float4 vv = reinterpret_cast<const float4*>(a.I + a.i[k] * a.ldi)[lane];
    const float g = a.gs[k];
    reinterpret_cast<float4*>(a.GU + k * a.d)[lane] = g * vv;
    reinterpret_cast<float4*>(a.GI + k * a.d)[lane] = g * uu;
  }
}

// the segment route applies: a workspace, a large batch, float4 rows, relation tables that fit the LDS accumulators
static bool seg_route(const void* ws, int64_t n, int d, int64_t n_rows, bool vec4, size_t lds_bytes) {
  return ws && n_rows > 0 && ktup::opt_seg_bwd_min() > 0 && n >= ktup::opt_seg_bwd_min() && vec4 && d <= 256 && lds_bytes <= 64 * 1024 &&
         n < (1ll << 30);
}
static size_t g_bytes(int64_t n, int d) { return (((size_t)n * d * sizeof(float)) + 255) & ~(size_t)255; }

static int transr_launch(bool bwd, const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm,
                         int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1, float* score,
                         const float* gs, float* gE, float* gR, float* gM, void* stream, const char* name) {
  if (int e = check_common(name, d, n)) return e;
  KTUP_REQUIRE(n == 0 || (E && R && M && h && t && r), "%s: null pointer argument", name);
  KTUP_REQUIRE(ldm >= (int64_t)d * d, "%s: projection pitch %lld < d*d", name, (long long)ldm);
  if (n == 0) return KTUP_OK;
  const size_t lds = (size_t)4 * 2 * d * sizeof(float);
  KTUP_REQUIRE(lds <= 64 * 1024, "%s: embedding_size %d too large", name, d);
  const int grid = grid_for((n + 3) / 4);
  if (bwd) {
    KTUP_REQUIRE(gs && gE && gR && gM, "%s: null gradient pointer", name);
    hipLaunchKernelGGL(transr_kernel<true>, dim3(grid), dim3(256), lds, (hipStream_t)stream, E, lde, R, ldr, M, ldm, d,
                       h, t, r, n, l1 != 0, nullptr, gs, gE, gR, gM);
  } else {
    KTUP_REQUIRE(score, "%s: null score pointer", name);
    hipLaunchKernelGGL(transr_kernel<false>, dim3(grid), dim3(256), lds, (hipStream_t)stream, E, lde, R, ldr, M, ldm, d,
                       h, t, r, n, l1 != 0, score, nullptr, nullptr, nullptr, nullptr);
  }
  return check_launch(name);
}

extern "C" size_t ktup_score_transr_workspace_bytes(int64_t n, int64_t n_rel) {
  return n > 0 && n_rel > 0 ? ktup::transr_mc_workspace_bytes(n, n_rel) : 0;
}

extern "C" int ktup_score_transr_fwd(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm,
                                     int64_t n_rel, int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n,
                                     int l1, float* score, void* ws, void* stream) {
  if (ws && n > 0 && E && R && M && h && t && r && score && ldm >= (int64_t)d * d) {
    const int rc = ktup::transr_fwd_mc(E, lde, R, ldr, M, ldm, n_rel, d, h, t, r, n, l1, score, ws, (hipStream_t)stream,
                                       "ktup_score_transr_fwd");
    if (rc != 1) return rc;
  }
  return transr_launch(false, E, lde, R, ldr, M, ldm, d, h, t, r, n, l1, score, nullptr, nullptr, nullptr, nullptr,
                       stream, "ktup_score_transr_fwd");
}

extern "C" int ktup_score_transr_bwd(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm,
                                     int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                                     const float* gscore, float* gE, float* gR, float* gM, void* stream) {
  return transr_launch(true, E, lde, R, ldr, M, ldm, d, h, t, r, n, l1, nullptr, gscore, gE, gR, gM, stream,
                       "ktup_score_transr_bwd");
}

// K4 backward with caller scratch: relation-bucketed on the matrix cores (d in {64, 100, 128}); from option seg_bwd_min rows on,
// the entity-row gradients go through the segment reduction instead of 2 d float atomics per triple.  Other shapes, or no
// scratch: exactly ktup_score_transr_bwd.
static size_t transr_bucket_bytes(int64_t n, int64_t n_rel) { return (ktup::transr_mc_workspace_bytes(n, n_rel) + 255) & ~(size_t)255; }
static bool transr_seg(int64_t n, int64_t n_ent) { return n_ent > 0 && ktup::opt_seg_bwd_min() > 0 && n >= ktup::opt_seg_bwd_min(); }

extern "C" size_t ktup_score_transr_bwd_workspace_bytes(int64_t n, int d, int64_t n_ent, int64_t n_rel) {
  if (n <= 0 || n_rel <= 0 || (d != 64 && d != 100 && d != 128)) return 0;
  return transr_bucket_bytes(n, n_rel) + (transr_seg(n, n_ent) ? g_bytes(n, d) + ktup::seg_ws_bytes(2 * n, n_ent) : 0);
}

extern "C" int ktup_score_transr_bwd_ws(const float* E, int64_t lde, const float* R, int64_t ldr, const float* M, int64_t ldm,
                                        int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                                        const float* gscore, float* gE, float* gR, float* gM, int64_t n_ent, int64_t n_rel, void* ws,
                                        void* stream) {
  const char* name = "ktup_score_transr_bwd_ws";
  if (ws && n > 0 && n_rel > 0 && E && R && M && h && t && r && gscore && gE && gR && gM && ktup::opt_pref_mc()) {
    hipStream_t st = (hipStream_t)stream;
    const bool seg = transr_seg(n, n_ent);
    char* base = reinterpret_cast<char*>(ws) + transr_bucket_bytes(n, n_rel);
    float* G = seg ? reinterpret_cast<float*>(base) : nullptr;
    void* sws = base + g_bytes(n, d);
    hipStream_t side = nullptr;
    if (seg) {     // heads then tails in ONE counting sort (gE[h] += G, gE[t] -= G), on the side stream beside the gradient kernel
      if (!ktup::seg_covers(G, d, d, n, 2 * n, n_ent, gE, lde, nullptr, nullptr, 0, sws))
        return set_error(KTUP_ERR_UNSUPPORTED, "%s: segment reduction does not cover this shape", name);
      side = ktup::fork_side(st);
      if (int e = ktup::seg_sort(h, t, n, 2 * n, n_ent, sws, side ? side : st, name)) { ktup::join_side(st, side); return e; }
    }
    int rc = ktup::transr_bwd_mc(E, lde, R, ldr, M, ldm, n_rel, d, h, t, r, n, l1, gscore, gE, gR, gM, G, ws, st, name);
    ktup::join_side(st, side);
    if (rc == KTUP_OK && seg) rc = ktup::seg_apply(G, d, d, n, 2 * n, n, n_ent, gE, lde, nullptr, -1, nullptr, 0, sws, st, name);
    if (rc != 1) return rc;
  }
  return transr_launch(true, E, lde, R, ldr, M, ldm, d, h, t, r, n, l1, nullptr, gscore, gE, gR, gM, stream, name);
}

// ---- backward with caller scratch: for n >= option seg_bwd_min (default 8192) the row gradients are written per row and summed
// per table row by sorted segments (ktup_segment_reduce_rows) instead of float atomics; otherwise exactly the *_bwd entry points.
extern "C" size_t ktup_score_kg_bwd_workspace_bytes(int64_t n, int d, int64_t n_ent) {
  if (n <= 0 || d <= 0 || d % 4 || n_ent <= 0 || ktup::opt_seg_bwd_min() <= 0 || n < ktup::opt_seg_bwd_min()) return 0;
  return g_bytes(n, d) + ktup::seg_ws_bytes(2 * n, n_ent);
}

static int kg_bwd_seg(bool transh, const char* name, const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                      int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1, const float* gscore, float* gE,
                      float* gR, float* gN, int64_t n_ent, int64_t n_rel, void* ws, hipStream_t st) {
  float* G = reinterpret_cast<float*>(ws);
  char* sws = reinterpret_cast<char*>(ws) + g_bytes(n, d);
  // ids of the two roles, heads then tails, in ONE counting sort (gE[h] += G, gE[t] -= G); it depends on the ids alone and
  // runs on the library's side stream beside the gradient kernel
  if (!ktup::seg_covers(G, d, d, n, 2 * n, n_ent, gE, lde, nullptr, nullptr, 0, sws))
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: segment reduction does not cover this shape", name);
  hipStream_t side = ktup::fork_side(st);
  if (int e = ktup::seg_sort(h, t, n, 2 * n, n_ent, sws, side ? side : st, name)) { ktup::join_side(st, side); return e; }
  KgSegArgs a{E, R, Nrm, lde, ldr, ldn, h, t, r, n, d / 4, d, (int)n_rel, l1 != 0, gscore, G, gR, gN};
  const size_t lds = (size_t)(transh ? 2 : 1) * n_rel * d * sizeof(float);
  const int nch = d / 4;
#define KTUP_KGSEG(GL)                                                                                                     \
  {                                                                                                                        \
    const int grid = grid_for((n + (256 / GL) - 1) / (256 / GL), 256 * 2);   /* every workgroup ends with n_rel x d atomics */ \
    const size_t plds = lds * (256 / GL);                                                                                  \
    const bool priv = plds <= 144 * 1024;                                                                                  \
    if (priv) {                                                                                                            \
      if (transh) {                                                                                                        \
        (void)hipFuncSetAttribute((const void*)kg_bwd_rowout_kernel<GL, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds); \
        hipLaunchKernelGGL((kg_bwd_rowout_kernel<GL, true, true>), dim3(grid), dim3(256), plds, st, a);                    \
      } else {                                                                                                             \
        (void)hipFuncSetAttribute((const void*)kg_bwd_rowout_kernel<GL, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds); \
        hipLaunchKernelGGL((kg_bwd_rowout_kernel<GL, false, true>), dim3(grid), dim3(256), plds, st, a);                   \
      }                                                                                                                    \
    } else {                                                                                                               \
      if (transh) hipLaunchKernelGGL((kg_bwd_rowout_kernel<GL, true, false>), dim3(grid), dim3(256), lds, st, a);          \
      else hipLaunchKernelGGL((kg_bwd_rowout_kernel<GL, false, false>), dim3(grid), dim3(256), lds, st, a);                \
    }                                                                                                                      \
  }
  if (nch <= 16) KTUP_KGSEG(16) else if (nch <= 32) KTUP_KGSEG(32) else KTUP_KGSEG(64)
#undef KTUP_KGSEG
  ktup::join_side(st, side);
  if (int e = check_launch(name)) return e;
  return ktup::seg_apply(G, d, d, n, 2 * n, n, n_ent, gE, lde, nullptr, -1, nullptr, 0, sws, st, name);
}

extern "C" int ktup_score_transe_bwd_ws(const float* E, int64_t lde, const float* R, int64_t ldr, int d, const int64_t* h,
                                        const int64_t* t, const int64_t* r, int64_t n, int l1, const float* gscore, float* gE,
                                        float* gR, int64_t n_ent, int64_t n_rel, void* ws, void* stream) {
  const char* name = "ktup_score_transe_bwd_ws";
  const bool v4ok = can_vec4(d, {E, R, gE, gR}, {lde, ldr});
  if (!seg_route(ws, n, d, n_ent, v4ok, (size_t)n_rel * d * 4) || n_rel <= 0)
    return ktup_score_transe_bwd(E, lde, R, ldr, d, h, t, r, n, l1, gscore, gE, gR, stream);
  if (int e = check_common(name, d, n)) return e;
  KTUP_REQUIRE(E && R && h && t && r && gscore && gE && gR, "%s: null pointer argument", name);
  return kg_bwd_seg(false, name, E, lde, R, ldr, nullptr, 0, d, h, t, r, n, l1, gscore, gE, gR, nullptr, n_ent, n_rel, ws, (hipStream_t)stream);
}

extern "C" int ktup_score_transh_bwd_ws(const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                                        int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t n, int l1,
                                        const float* gscore, float* gE, float* gR, float* gN, int64_t n_ent, int64_t n_rel, void* ws,
                                        void* stream) {
  const char* name = "ktup_score_transh_bwd_ws";
  const bool v4ok = can_vec4(d, {E, R, Nrm, gE, gR, gN}, {lde, ldr, ldn});
  if (!seg_route(ws, n, d, n_ent, v4ok, (size_t)2 * n_rel * d * 4) || n_rel <= 0)
    return ktup_score_transh_bwd(E, lde, R, ldr, Nrm, ldn, d, h, t, r, n, l1, gscore, gE, gR, gN, stream);
  if (int e = check_common(name, d, n)) return e;
  KTUP_REQUIRE(E && R && Nrm && h && t && r && gscore && gE && gR && gN, "%s: null pointer argument", name);
  return kg_bwd_seg(true, name, E, lde, R, ldr, Nrm, ldn, d, h, t, r, n, l1, gscore, gE, gR, gN, n_ent, n_rel, ws, (hipStream_t)stream);
}

extern "C" size_t ktup_score_bprmf_bwd_workspace_bytes(int64_t n, int d, int64_t n_users, int64_t n_items) {
  if (n <= 0 || d <= 0 || d % 4 || n_users <= 0 || n_items <= 0 || ktup::opt_seg_bwd_min() <= 0 || n < ktup::opt_seg_bwd_min()) return 0;
  return 2 * g_bytes(n, d) + ktup::seg_ws_bytes(n, n_users) + ktup::seg_ws_bytes(n, n_items);
}

extern "C" int ktup_score_bprmf_bwd_ws(const float* U, int64_t ldu, const float* I, int64_t ldi, int d, const int64_t* u_ids,
                                       const int64_t* i_ids, int64_t n, const float* gscore, float* gU, float* gI, int64_t n_users,
                                       int64_t n_items, void* ws, void* stream) {
  const char* name = "ktup_score_bprmf_bwd_ws";
  const bool v4ok = can_vec4(d, {U, I, gU, gI}, {ldu, ldi});
  if (!seg_route(ws, n, d, n_users, v4ok, 0) || n_items <= 0)
    return ktup_score_bprmf_bwd(U, ldu, I, ldi, d, u_ids, i_ids, n, gscore, gU, gI, stream);
  if (int e = check_common(name, d, n)) return e;
  KTUP_REQUIRE(U && I && u_ids && i_ids && gscore && gU && gI, "%s: null pointer argument", name);
  hipStream_t st = (hipStream_t)stream;
  float* GU = reinterpret_cast<float*>(ws);
  float* GI = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + g_bytes(n, d));
  char* swsU = reinterpret_cast<char*>(ws) + 2 * g_bytes(n, d);
  char* swsI = swsU + ktup::seg_ws_bytes(n, n_users);
  if (!ktup::seg_covers(GU, d, d, n, n, n_users, gU, ldu, nullptr, nullptr, 0, swsU) || !ktup::seg_covers(GI, d, d, n, n, n_items, gI, ldi, nullptr, nullptr, 0, swsI))
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: segment reduction does not cover this shape", name);
  hipStream_t side = ktup::fork_side(st);
  {
    hipStream_t ss = side ? side : st;
    int e = ktup::seg_sort(u_ids, nullptr, n, n, n_users, swsU, ss, name);
    if (e == KTUP_OK) e = ktup::seg_sort(i_ids, nullptr, n, n, n_items, swsI, ss, name);
    if (e != KTUP_OK) { ktup::join_side(st, side); return e; }
  }
  BprmfSegArgs a{U, I, ldu, ldi, u_ids, i_ids, n, d / 4, d, gscore, GU, GI};
  const int nch = d / 4;
#define KTUP_BSEG(GL)                                                                                         \
  {                                                                                                           \
    hipLaunchKernelGGL((bprmf_bwd_rowout_kernel<GL>), dim3(grid_for((n + (256 / GL) - 1) / (256 / GL), 256 * 8)), dim3(256), 0, st, a); \
  }
  if (nch <= 16) KTUP_BSEG(16) else if (nch <= 32) KTUP_BSEG(32) else KTUP_BSEG(64)
#undef KTUP_BSEG
  ktup::join_side(st, side);
  if (int e = check_launch(name)) return e;
  hipStream_t s2 = ktup::fork_side(st);      // disjoint tables: the two reductions run side by side
  int rc = ktup::seg_apply(GU, d, d, n, n, n, n_users, gU, ldu, nullptr, -1, nullptr, 0, swsU, s2 ? s2 : st, name);
  if (rc == KTUP_OK) rc = ktup::seg_apply(GI, d, d, n, n, n, n_items, gI, ldi, nullptr, -1, nullptr, 0, swsI, st, name);
  ktup::join_side(st, s2);
  return rc;
}
