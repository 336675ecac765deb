// K15 / K16, soft gate, squared-L2: the all-item TUP / KTUP scores as six small GEMMs on the matrix cores.
//
// After the gate decomposition of ktup_eval.hip (pref_project_kernel) every user b has  AU = u + RU, u, NU  and every item j
// has  C0 = v - RV, v, NV  (v = item (+ entity) row), and
//     a = AU_b - C0_j,   n = NU_b + NV_j,   q = u_b - v_j,   s = q . n,   score = | a - s n |^2          (transUP.py:84-102,
// jTransUP.py:163-191 evaluate the same expression after materialising ~8 (B x N x d) tensors).  For the squared distance
//     score = |a|^2 - 2 s (a . n) + s^2 |n|^2
// and every term is bilinear in (user vector, item vector):
//     s     = u.NU + [u.NV] - [NU.v] - v.NV
//     |a|^2 = |AU|^2 - 2 [AU.C0] + |C0|^2
//     a . n = AU.NU + [AU.NV] - [NU.C0] - C0.NV
//     |n|^2 = |NU|^2 + 2 [NU.NV] + |NV|^2
// The six bracketed (users x items) products are (B x d).(d x N) GEMMs -- the "full-catalogue eval GEMM" -- computed here on
// v_mfma_f32_16x16x4_f32 with fp32 accumulation; the other eight terms are per-user / per-item scalars.  A workgroup stages
// UB users and 64 items (three vectors each) in LDS once; each wave owns a 16 x 16 (user x item) tile and all six
// accumulators.  The L1 distance does not expand this way and stays on the VALU kernel (pairs_kernel<2>).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ktup_common.h"
#include "ktup_lane_swap.h"
#include "ktup_pref_geom.h"

namespace ktup {
namespace {

constexpr int IB = 64;   // items per workgroup

template <int NCH_, int UB_>
struct EGeom {
  static constexpr int NCH = NCH_, UB = UB_, D = 4 * NCH;
  static constexpr int KG = (D + 15) / 16;
  static constexpr bool TAIL1 = NCH - 4 * (KG - 1) == 1;       // d % 16 == 4: the last chunk goes through one b32-operand MFMA
  static constexpr int KGF = TAIL1 ? KG - 1 : KG;
  static_assert(TAIL1 || NCH % 4 == 0, "k groups must be whole (d % 16 in {0, 4})");
  static constexpr int P4 = NCH | 1;                           // odd float4 row pitch: conflict-free b128 operand reads
  static constexpr int NW = (UB / 16) * (IB / 16);
  static constexpr int ROWS = UB + IB;
  static constexpr size_t LDS = (size_t)ROWS * 3 * P4 * 16 + (size_t)ROWS * 4 * 4;
};

struct EArgs {
  const float *QW, *C0, *C1, *C2;   // users: rows of 3 d floats [AU | u | NU]; items: three (n_items x d) arrays
  int64_t nq, n_items;
  float* out; int64_t ldo;
};

template <typename G>
__global__ __launch_bounds__(G::NW * 64) void pairs_l2_mc_kernel(EArgs a) {
  constexpr int NCH = G::NCH, UB = G::UB, D = G::D, KGF = G::KGF, P4 = G::P4, NW = G::NW, ROWS = G::ROWS;
  constexpr bool TAIL1 = G::TAIL1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* X = reinterpret_cast<v4*>(smem);                        // [ROWS][3][P4]: rows 0..UB-1 users, UB.. items
  float* sc = reinterpret_cast<float*>(X + ROWS * 3 * P4);    // [ROWS][4] scalars
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t u0 = (int64_t)blockIdx.x * UB, i0 = (int64_t)blockIdx.y * IB;
  // ---- stage: users [AU | u | NU], items [C0 | v | NV]; rows past the end are zero
  for (int idx = tid; idx < ROWS * 3 * NCH; idx += NW * 64) {
    const int row = idx / (3 * NCH), rem = idx - row * (3 * NCH), vec = rem / NCH, c = rem - vec * NCH;
    v4 val = (v4){0.f, 0.f, 0.f, 0.f};
    if (row < UB) {
      if (u0 + row < a.nq) val = *reinterpret_cast<const v4*>(a.QW + ((u0 + row) * 3 + vec) * D + 4 * c);
    } else if (i0 + row - UB < a.n_items) {
      const float* src = vec == 0 ? a.C0 : vec == 1 ? a.C1 : a.C2;
      val = *reinterpret_cast<const v4*>(src + (i0 + row - UB) * D + 4 * c);
    }
    X[(row * 3 + vec) * P4 + c] = val;
  }
  __syncthreads();
  // ---- per-row scalars, 8 lanes per row.  users: u.NU, |AU|^2, AU.NU, |NU|^2;  items: v.NV, |C0|^2, C0.NV, |NV|^2
  for (int row = tid >> 3; row < ROWS; row += (NW * 64) >> 3) {
    const v4* r0 = X + (row * 3 + 0) * P4;
    const v4* r1 = r0 + P4;
    const v4* r2 = r1 + P4;
    v4 s0 = (v4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    for (int c = tid & 7; c < NCH; c += 8) {
      const v4 x0 = r0[c], x1 = r1[c], x2 = r2[c];
      s0 += x1 * x2; s1 += x0 * x0; s2 += x0 * x2; s3 += x2 * x2;
    }
    float f0 = (s0[0] + s0[1]) + (s0[2] + s0[3]), f1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    float f2 = (s2[0] + s2[1]) + (s2[2] + s2[3]), f3 = (s3[0] + s3[1]) + (s3[2] + s3[3]);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      f0 += __shfl_xor(f0, m, 64); f1 += __shfl_xor(f1, m, 64); f2 += __shfl_xor(f2, m, 64); f3 += __shfl_xor(f3, m, 64);
    }
    if ((tid & 7) == 0) { sc[row * 4 + 0] = f0; sc[row * 4 + 1] = f1; sc[row * 4 + 2] = f2; sc[row * 4 + 3] = f3; }
  }
  __syncthreads();
  // ---- this wave's 16 x 16 tile: users are MFMA rows (A operand), items columns (B operand)
  const int ut = w / (IB / 16), it = w % (IB / 16);
  const v4* ua = X + ((16 * ut + j) * 3) * P4 + kq;           // lane (kq, row j): AU at +0, u at +P4, NU at +2 P4
  const v4* ib = X + ((UB + 16 * it + j) * 3) * P4 + kq;      // lane (kq, col j): C0, v, NV
  v4 uNV = (v4){0.f, 0.f, 0.f, 0.f}, NUv = uNV, AUC0 = uNV, AUNV = uNV, NUC0 = uNV, NUNV = uNV;
#pragma unroll
  for (int g = 0; g < KGF; ++g) {
    const v4 aAU = ua[4 * g], au = ua[P4 + 4 * g], aNU = ua[2 * P4 + 4 * g];
    const v4 bC0 = ib[4 * g], bv = ib[P4 + 4 * g], bNV = ib[2 * P4 + 4 * g];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uNV = __builtin_amdgcn_mfma_f32_16x16x4f32(au[c], bNV[c], uNV, 0, 0, 0);
      NUv = __builtin_amdgcn_mfma_f32_16x16x4f32(aNU[c], bv[c], NUv, 0, 0, 0);
      AUC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aAU[c], bC0[c], AUC0, 0, 0, 0);
      AUNV = __builtin_amdgcn_mfma_f32_16x16x4f32(aAU[c], bNV[c], AUNV, 0, 0, 0);
      NUC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aNU[c], bC0[c], NUC0, 0, 0, 0);
      NUNV = __builtin_amdgcn_mfma_f32_16x16x4f32(aNU[c], bNV[c], NUNV, 0, 0, 0);
    }
  }
  if (TAIL1) {                                                // coordinates 16 KGF + kq
    const float* uf = reinterpret_cast<const float*>(X + ((16 * ut + j) * 3) * P4 + 4 * KGF) + kq;
    const float* jf = reinterpret_cast<const float*>(X + ((UB + 16 * it + j) * 3) * P4 + 4 * KGF) + kq;
    const float aAU = uf[0], au = uf[4 * P4], aNU = uf[8 * P4];
    const float bC0 = jf[0], bv = jf[4 * P4], bNV = jf[8 * P4];
    uNV = __builtin_amdgcn_mfma_f32_16x16x4f32(au, bNV, uNV, 0, 0, 0);
    NUv = __builtin_amdgcn_mfma_f32_16x16x4f32(aNU, bv, NUv, 0, 0, 0);
    AUC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aAU, bC0, AUC0, 0, 0, 0);
    AUNV = __builtin_amdgcn_mfma_f32_16x16x4f32(aAU, bNV, AUNV, 0, 0, 0);
    NUC0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aNU, bC0, NUC0, 0, 0, 0);
    NUNV = __builtin_amdgcn_mfma_f32_16x16x4f32(aNU, bNV, NUNV, 0, 0, 0);
  }
  // ---- epilogue: lane (kq, j) holds users 16 ut + 4 kq + reg (reg = 0..3) x item 16 it + j
  const v4 isc = *reinterpret_cast<const v4*>(sc + (UB + 16 * it + j) * 4);          // v.NV, |C0|^2, C0.NV, |NV|^2
  const int64_t item = i0 + 16 * it + j;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int ur = 16 * ut + 4 * kq + reg;
    const v4 usc = *reinterpret_cast<const v4*>(sc + ur * 4);                           // u.NU, |AU|^2, AU.NU, |NU|^2
    const float s = (usc[0] + uNV[reg]) - (NUv[reg] + isc[0]);
    const float aa = fmaf(-2.f, AUC0[reg], usc[1] + isc[1]);
    const float an = (usc[2] + AUNV[reg]) - (NUC0[reg] + isc[2]);
    const float nn = fmaf(2.f, NUNV[reg], usc[3] + isc[3]);
    const float score = fmaf(s * s, nn, fmaf(-2.f * s, an, aa));
    if (u0 + ur < a.nq && item < a.n_items) a.out[(u0 + ur) * a.ldo + item] = score;
  }
}

template <typename G>
int launch(const EArgs& a, hipStream_t st, const char* name) {
  static_assert(G::LDS <= 160 * 1024, "LDS budget");
  (void)hipFuncSetAttribute((const void*)pairs_l2_mc_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
  const dim3 grid((unsigned)((a.nq + G::UB - 1) / G::UB), (unsigned)((a.n_items + IB - 1) / IB));
  hipLaunchKernelGGL((pairs_l2_mc_kernel<G>), grid, dim3(G::NW * 64), G::LDS, st, a);
  return check_launch(name);
}

// ---------------------------------------------------------------------------------------------------------------------------
// K12 / K13, squared L2: all-entity TransE / TransH scores.  Per query b (kg_query_prep_kernel): c_b (translated, projected
// query) and, for TransH, the hyperplane normal w_b; per candidate: the entity row e_j.
//   TransE (transE.py:65-105):   score = |c - e|^2                   = |c|^2 - 2 [c.e] + |e|^2
//   TransH (transH.py:73-121):   score = |c - e + (e.w) w|^2         = |c|^2 + |e|^2 - 2 [c.e] + [e.w] (2 c.w + [e.w] (|w|^2 - 2))
// (the reference projects every candidate on every query's hyperplane: three B x E x d tensors).  One or two GEMMs [.] on the
// matrix cores, the rest per-row scalars.  L1 stays on pairs_kernel<0/1>.
template <int NCH_, bool TRANSH_>
struct KGeom {
  static constexpr int NCH = NCH_, D = 4 * NCH;
  static constexpr bool TRANSH = TRANSH_;
  static constexpr int KG = (D + 15) / 16;
  static constexpr bool TAIL1 = NCH - 4 * (KG - 1) == 1;
  static constexpr int KGF = TAIL1 ? KG - 1 : KG;
  static_assert(TAIL1 || NCH % 4 == 0, "k groups must be whole (d % 16 in {0, 4})");
  static constexpr int P4 = NCH | 1;
  static constexpr int QV = TRANSH ? 2 : 1;                    // vectors per query
  static constexpr int UB = 64, NW = 16;
  static constexpr size_t LDS = (size_t)(UB * QV + IB) * P4 * 16 + (size_t)(UB * 4 + IB) * 4;
};

struct KArgs {
  const float* QW; int dq;          // queries: rows of 3 dq floats, slot 0 = c, slot 2 = w
  const float* C; int64_t ldc;      // candidates
  int64_t nq, n_cand;
  float* out; int64_t ldo;
  // TransR (all three non-NULL): slot 0 holds c' = M_r^T c, the query's |c|^2 comes from qcc and the candidate term is
  // |M_r e|^2 = cnorm[qrel[query] * n_cand + candidate]  (|c - M_r e|^2 = |c|^2 - 2 c'.e + |M_r e|^2)
  const float* qcc; const float* cnorm; const int32_t* qrel;
};

template <typename G>
__global__ __launch_bounds__(G::NW * 64) void pairs_kg_l2_mc_kernel(KArgs a) {
  constexpr int NCH = G::NCH, UB = G::UB, KGF = G::KGF, P4 = G::P4, NW = G::NW, QV = G::QV;
  constexpr bool TAIL1 = G::TAIL1, TRANSH = G::TRANSH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4* Q = reinterpret_cast<v4*>(smem);                         // [UB][QV][P4]
  v4* Cd = Q + UB * QV * P4;                                   // [IB][P4]
  float* qs = reinterpret_cast<float*>(Cd + IB * P4);          // [UB][4]: |c|^2, c.w, |w|^2
  float* cs = qs + UB * 4;                                     // [IB]:    |e|^2
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, j = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order: the hardware deals consecutive workgroups round-robin to the 8 XCDs (each with its own 4 MB L2).
  // The entity table (5.9 MB at ml1m) does not fit one L2, so every XCD gets a contiguous BAND of candidate tiles (all of
  // its workgroups then re-read the same ~0.75 MB of candidates and the whole, small, query block) instead of all of them.
  // The launch pads grid.x to 8 bands of nxp tiles; tiles past the last real one exit here.
  const int nxp = gridDim.x >> 3;
  const int bid = blockIdx.y * gridDim.x + blockIdx.x, xcd = bid & 7, local = bid >> 3;
  const int tx = xcd * nxp + local % nxp, ty = local / nxp;
  if ((int64_t)tx * IB >= a.n_cand) return;
  const int64_t u0 = (int64_t)ty * UB, i0 = (int64_t)tx * IB;
  for (int idx = tid; idx < UB * QV * NCH; idx += NW * 64) {
    const int row = idx / (QV * NCH), rem = idx - row * (QV * NCH), vec = rem / NCH, c = rem - vec * NCH;
    v4 val = (v4){0.f, 0.f, 0.f, 0.f};
    if (u0 + row < a.nq) val = *reinterpret_cast<const v4*>(a.QW + ((u0 + row) * 3 + 2 * vec) * a.dq + 4 * c);
    Q[(row * QV + vec) * P4 + c] = val;
  }
  for (int idx = tid; idx < IB * NCH; idx += NW * 64) {
    const int row = idx / NCH, c = idx - row * NCH;
    v4 val = (v4){0.f, 0.f, 0.f, 0.f};
    if (i0 + row < a.n_cand) val = *reinterpret_cast<const v4*>(a.C + (i0 + row) * a.ldc + 4 * c);
    Cd[row * P4 + c] = val;
  }
  __syncthreads();
  for (int row = tid >> 3; row < UB + IB; row += (NW * 64) >> 3) {   // 8 lanes per row
    v4 s0 = (v4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0;
    if (row < UB) {
      const v4* r0 = Q + row * QV * P4;
      for (int c = tid & 7; c < NCH; c += 8) {
        const v4 x0 = r0[c];
        s0 += x0 * x0;
        if (TRANSH) { const v4 x1 = r0[P4 + c]; s1 += x0 * x1; s2 += x1 * x1; }
      }
    } else {
      const v4* r0 = Cd + (row - UB) * P4;
      for (int c = tid & 7; c < NCH; c += 8) { const v4 x0 = r0[c]; s0 += x0 * x0; }
    }
    float f0 = (s0[0] + s0[1]) + (s0[2] + s0[3]), f1 = (s1[0] + s1[1]) + (s1[2] + s1[3]), f2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) { f0 += __shfl_xor(f0, m, 64); f1 += __shfl_xor(f1, m, 64); f2 += __shfl_xor(f2, m, 64); }
    if ((tid & 7) == 0) {
      if (row < UB) { qs[row * 4 + 0] = f0; qs[row * 4 + 1] = f1; qs[row * 4 + 2] = f2; }
      else cs[row - UB] = f0;
    }
  }
  __syncthreads();
  const int ut = w >> 2, it = w & 3;
  const v4* qa = Q + ((16 * ut + j) * QV) * P4 + kq;
  const v4* cb = Cd + (16 * it + j) * P4 + kq;
  v4 ce = (v4){0.f, 0.f, 0.f, 0.f}, we = ce;
#pragma unroll
  for (int g = 0; g < KGF; ++g) {
    const v4 ac = qa[4 * g], be = cb[4 * g];
    v4 aw = ac;
    if (TRANSH) aw = qa[P4 + 4 * g];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ce = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[c], be[c], ce, 0, 0, 0);
      if (TRANSH) we = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[c], be[c], we, 0, 0, 0);
    }
  }
  if (TAIL1) {
    const float* qf = reinterpret_cast<const float*>(Q + ((16 * ut + j) * QV) * P4 + 4 * KGF) + kq;
    const float be = (reinterpret_cast<const float*>(Cd + (16 * it + j) * P4 + 4 * KGF) + kq)[0];
    ce = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[0], be, ce, 0, 0, 0);
    if (TRANSH) we = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[4 * P4], be, we, 0, 0, 0);
  }
  const float ee = cs[16 * it + j];
  const int64_t cand = i0 + 16 * it + j;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int ur = 16 * ut + 4 * kq + reg;
    float cc = qs[ur * 4 + 0], en = ee;
    if (a.cnorm && u0 + ur < a.nq && cand < a.n_cand) {
      cc = a.qcc[u0 + ur];
      en = a.cnorm[(int64_t)a.qrel[u0 + ur] * a.n_cand + cand];
    }
    float score = fmaf(-2.f, ce[reg], cc + en);
    if (TRANSH) {
      const float cw = qs[ur * 4 + 1], ww = qs[ur * 4 + 2], ew = we[reg];
      score = fmaf(ew, fmaf(ew, ww - 2.f, 2.f * cw), score);
    }
    if (u0 + ur < a.nq && cand < a.n_cand) a.out[(u0 + ur) * a.ldo + cand] = score;
  }
}

template <typename G>
int launch_kg(const KArgs& a, hipStream_t st, const char* name) {
  (void)hipFuncSetAttribute((const void*)pairs_kg_l2_mc_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
  const unsigned nx = (unsigned)((a.n_cand + IB - 1) / IB), nxp = (nx + 7) / 8;       // 8 XCD bands of nxp candidate tiles
  const dim3 grid(8 * nxp, (unsigned)((a.nq + G::UB - 1) / G::UB));
  hipLaunchKernelGGL((pairs_kg_l2_mc_kernel<G>), grid, dim3(G::NW * 64), G::LDS, st, a);
  return check_launch(name);
}

template <bool TRANSH>
int dispatch_kg(const KArgs& a, int d, hipStream_t st, const char* name) {
  switch (d) {
    case 20: return launch_kg<KGeom<5, TRANSH>>(a, st, name);
    case 36: return launch_kg<KGeom<9, TRANSH>>(a, st, name);
    case 64: return launch_kg<KGeom<16, TRANSH>>(a, st, name);
    case 100: return launch_kg<KGeom<25, TRANSH>>(a, st, name);
    case 128: return launch_kg<KGeom<32, TRANSH>>(a, st, name);
    default: return 1;
  }
}

}  // namespace

// Returns KTUP_OK / an error, or 1 when d is not an instantiated size (the caller runs the VALU kernel).
int pairs_l2_mc(const float* QW, const float* C0, const float* C1, const float* C2, int d, int64_t nq, int64_t n_items, float* out,
                int64_t ldo, hipStream_t st, const char* name) {
  if (!aligned16(QW) || !aligned16(C0) || !aligned16(C1) || !aligned16(C2)) return 1;
  if ((n_items + IB - 1) / IB > 65535) return 1;
  const EArgs a{QW, C0, C1, C2, nq, n_items, out, ldo};
  switch (d) {
    case 20: return launch<EGeom<5, 64>>(a, st, name);
    case 36: return launch<EGeom<9, 64>>(a, st, name);
    case 64: return launch<EGeom<16, 64>>(a, st, name);
    case 100: return launch<EGeom<25, 64>>(a, st, name);
    case 128: return launch<EGeom<32, 32>>(a, st, name);
    default: return 1;
  }
}

// model: 0 TransE, 1 TransH.  Same contract as pairs_l2_mc.
int pairs_kg_l2_mc(int model, const float* QW, int dq, const float* C, int64_t ldc, int d, int64_t nq, int64_t n_cand, float* out,
                   int64_t ldo, hipStream_t st, const char* name, const float* qcc, const float* cnorm, const int32_t* qrel) {
  if (!aligned16(QW) || !aligned16(C) || (ldc & 3) || dq != d) return 1;
  if ((nq + 63) / 64 > 65535) return 1;
  const KArgs a{QW, dq, C, ldc, nq, n_cand, out, ldo, qcc, cnorm, qrel};
  return model == 1 ? dispatch_kg<true>(a, d, st, name) : dispatch_kg<false>(a, d, st, name);
}

}  // namespace ktup
