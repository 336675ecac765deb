// Row-kernel framework shared by the fused gather+score kernels (K1-K3) and the regularisers (K10):
// G = 16/32/64 consecutive lanes own one gathered row, loads are runs of contiguous 16-B (or 4-B) pieces,
// reductions are xor-shuffles inside the lane group.
#pragma once
#include <initializer_list>

#include "ktup_common.h"

namespace ktup {

// ---- tiny vector abstraction: V = float4 (fast path) or float --------------------------------
KTUP_DEV void vzero(float& a) { a = 0.f; }
KTUP_DEV void vzero(float4& a) { a = f4zero(); }
KTUP_DEV float vadd(float a, float b) { return a + b; }
KTUP_DEV float4 vadd(float4 a, float4 b) { return a + b; }
KTUP_DEV float vsub(float a, float b) { return a - b; }
KTUP_DEV float4 vsub(float4 a, float4 b) { return a - b; }
KTUP_DEV float vscale(float s, float a) { return s * a; }
KTUP_DEV float4 vscale(float s, float4 a) { return s * a; }
KTUP_DEV float vfma(float s, float a, float c) { return fmaf(s, a, c); }
KTUP_DEV float4 vfma(float s, float4 a, float4 c) { return fma4(s, a, c); }
KTUP_DEV float vdot(float a, float b) { return a * b; }
KTUP_DEV float vdot(float4 a, float4 b) { return dot4(a, b); }
KTUP_DEV float vdist(float z, bool l1) { return dist1(z, l1); }
KTUP_DEV float vdist(float4 z, bool l1) { return dist4(z, l1); }
KTUP_DEV float vddist(float z, bool l1) { return ddist1(z, l1); }
KTUP_DEV float4 vddist(float4 z, bool l1) { return ddist4(z, l1); }
KTUP_DEV void vatomic(float* p, float v) { atomicAdd(p, v); }
KTUP_DEV void vatomic(float* p, float4 v) { atomic_add4(p, v); }
template <typename V> struct VW;
template <> struct VW<float> { static constexpr int W = 1; };
template <> struct VW<float4> { static constexpr int W = 4; };

template <typename V, int G, int CPL>
struct RowCtx {
  int nch;   // chunks of V per row
  int lane;  // lane inside the G-lane group
  KTUP_DEV void load(V (&x)[CPL], const float* row) const {
    const V* p = reinterpret_cast<const V*>(row);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + j * G;
      if (c < nch) x[j] = p[c]; else vzero(x[j]);
    }
  }
  KTUP_DEV void scatter_add(float* row, const V (&g)[CPL]) const {
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + j * G;
      if (c < nch) vatomic(row + (int64_t)c * VW<V>::W, g[j]);
    }
  }
};

template <typename V, int G, int CPL, typename Op>
__global__ __launch_bounds__(256) void row_kernel(Op op, int nch, int64_t n) {
  RowCtx<V, G, CPL> cx{nch, (int)(threadIdx.x % G)};
  constexpr int GPB = 256 / G;  // rows in flight per workgroup
  for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / G; row < n; row += (int64_t)gridDim.x * GPB)
    op.template run<V, G, CPL>(cx, row);
}

template <typename Op>
int launch_rows(const Op& op, int d, bool vec4, int64_t n, hipStream_t st, const char* name) {
  if (n == 0) return KTUP_OK;
  const int nch = vec4 ? d / 4 : d;
#define KTUP_L(V, G, CPL)                                                                          \
  {                                                                                                \
    const int grid = grid_for((n + (256 / G) - 1) / (256 / G));                                   \
    hipLaunchKernelGGL((row_kernel<V, G, CPL, Op>), dim3(grid), dim3(256), 0, st, op, nch, n);    \
    return check_launch(name);                                                                     \
  }
  if (vec4) {
    if (nch <= 16) KTUP_L(float4, 16, 1)
    if (nch <= 32) KTUP_L(float4, 32, 1)
    if (nch <= 64) KTUP_L(float4, 64, 1)
    if (nch <= 128) KTUP_L(float4, 64, 2)
    if (nch <= 256) KTUP_L(float4, 64, 4)
  } else {
    if (nch <= 16) KTUP_L(float, 16, 1)
    if (nch <= 32) KTUP_L(float, 32, 1)
    if (nch <= 64) KTUP_L(float, 64, 1)
    if (nch <= 128) KTUP_L(float, 64, 2)
    if (nch <= 256) KTUP_L(float, 64, 4)
  }
#undef KTUP_L
  return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size %d too large for the row kernels", name, d);
}

inline bool can_vec4(int d, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds) {
  if (d % 4) return false;
  for (const void* p : ptrs) if (p && !aligned16(p)) return false;
  for (int64_t l : lds) if (l % 4) return false;
  return true;
}


// Block-wide sum of one float per thread; the total is valid in thread 0.
KTUP_DEV float block_sum_256(float v) {
  __shared__ float red[4];
  v = group_sum<64>(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return threadIdx.x == 0 ? red[0] + red[1] + red[2] + red[3] : 0.f;
}

// Same row mapping, but every row contributes one float (returned by op.run, meaningful on the group's lane 0)
// to a scalar: per-thread partials -> one block reduction -> ONE atomic per workgroup.
template <typename V, int G, int CPL, typename Op>
__global__ __launch_bounds__(256) void row_reduce_kernel(Op op, int nch, int64_t n, float* __restrict__ out, int serial) {
  RowCtx<V, G, CPL> cx{nch, (int)(threadIdx.x % G)};
  constexpr int GPB = 256 / G;
  float part = 0.f;
  // serial (option `deterministic`): ONE lane group walks every row, so a row listed several times receives its gradient adds in
  // program order and the value is summed in row order
  const int64_t row0 = serial ? (threadIdx.x < G ? 0 : n) : (int64_t)blockIdx.x * GPB + threadIdx.x / G;
  const int64_t rstep = serial ? 1 : (int64_t)gridDim.x * GPB;
  for (int64_t row = row0; row < n; row += rstep) {
    const float v = op.template run<V, G, CPL>(cx, row);
    if (cx.lane == 0) part += v;
  }
  part = block_sum_256(part);
  if (threadIdx.x == 0) atomicAdd(out, part);
}

template <typename Op>
int launch_rows_reduce(const Op& op, int d, bool vec4, int64_t n, float* out, hipStream_t st, const char* name, bool zero_first = true) {
  if (zero_first && hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return check_launch(name);
  if (n == 0) return KTUP_OK;
  const int nch = vec4 ? d / 4 : d;
#define KTUP_L(V, G, CPL)                                                                                   \
  {                                                                                                         \
    const int serial = opt_deterministic() != 0;                                                            \
    const int grid = serial ? 1 : grid_for((n + (256 / G) - 1) / (256 / G), 128);                           \
    hipLaunchKernelGGL((row_reduce_kernel<V, G, CPL, Op>), dim3(grid), dim3(256), 0, st, op, nch, n, out, serial);  \
    return check_launch(name);                                                                              \
  }
  if (vec4) {
    if (nch <= 16) KTUP_L(float4, 16, 1)
    if (nch <= 32) KTUP_L(float4, 32, 1)
    if (nch <= 64) KTUP_L(float4, 64, 1)
    if (nch <= 128) KTUP_L(float4, 64, 2)
    if (nch <= 256) KTUP_L(float4, 64, 4)
  } else {
    if (nch <= 16) KTUP_L(float, 16, 1)
    if (nch <= 32) KTUP_L(float, 32, 1)
    if (nch <= 64) KTUP_L(float, 64, 1)
    if (nch <= 128) KTUP_L(float, 64, 2)
    if (nch <= 256) KTUP_L(float, 64, 4)
  }
#undef KTUP_L
  return set_error(KTUP_ERR_UNSUPPORTED, "%s: embedding_size %d too large for the row kernels", name, d);
}

}  // namespace ktup
