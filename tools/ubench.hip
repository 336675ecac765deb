// Micro-benchmarks of the primitives the preference-gate kernels are built from (run on the GPU box):
// FMA issue rate with VGPR vs SGPR vs freshly s_loaded operands, packed FMA, LDS-broadcast operands, MFMA f32.
//   hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v16f* sptr16;

constexpr int NACC = 32;

// A: all-VGPR fma
__global__ __launch_bounds__(256) void k_fma_vgpr(float* out, int iters, float seed) {
  float acc[NACC], m[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = threadIdx.x * 0.001f + i; m[i] = seed + i * 1e-3f + threadIdx.x * 1e-6f; }
  const float x = seed * 0.5f + threadIdx.x * 1e-7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = fmaf(x, m[i], acc[i]);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// B: one SGPR operand (values come from a uniform table loaded once)
__global__ __launch_bounds__(256) void k_fma_sgpr(float* out, int iters, const float* tab) {
  float acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 0.001f + i;
  const sptr16 t = (sptr16)(uintptr_t)tab;
  const v16f a = t[0], b = t[1];
  const float x = threadIdx.x * 1e-7f + 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fmaf(x, a[i], acc[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[16 + i] = fmaf(x, b[i], acc[16 + i]);
    asm volatile("" ::: "memory");
  }
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// C: SGPR operands re-loaded (s_load_dwordx16 x2) every iteration, REP fmas per loaded value
template <int REP>
__global__ __launch_bounds__(256) void k_fma_sload(float* out, int iters, const float* tab, int stride16) {
  float acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 0.001f + i;
  sptr16 t = (sptr16)(uintptr_t)tab;
  float x[REP];
  for (int r = 0; r < REP; ++r) x[r] = threadIdx.x * 1e-7f + 0.5f + r;
  for (int it = 0; it < iters; ++it) {
    const v16f a = t[0], b = t[1];
    t += stride16;
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(x[r], a[i], acc[i]);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[16 + i] = fmaf(x[r], b[i], acc[16 + i]);
    }
  }
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// D: packed fma, VGPR operands
__global__ __launch_bounds__(256) void k_pkfma_vgpr(float* out, int iters, float seed) {
  v2f acc[NACC / 2], m[NACC / 2];
  for (int i = 0; i < NACC / 2; ++i) { acc[i] = (v2f){threadIdx.x * 0.001f + i, 1.f}; m[i] = (v2f){seed + i * 1e-3f, seed + threadIdx.x * 1e-6f}; }
  const v2f x = (v2f){seed * 0.5f + threadIdx.x * 1e-7f, seed};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC / 2; ++i) acc[i] = __builtin_elementwise_fma(x, m[i], acc[i]);
  }
  float s = 0; for (int i = 0; i < NACC / 2; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// E: operand broadcast from LDS (same address for all lanes), one ds_read_b128 per 4 fmas x REP rows
template <int REP>
__global__ __launch_bounds__(256) void k_fma_lds(float* out, int iters, const float* tab) {
  __shared__ float4 lt[512];
  for (int i = threadIdx.x; i < 512; i += 256) lt[i] = reinterpret_cast<const float4*>(tab)[i];
  __syncthreads();
  float4 acc[REP][8];
  for (int r = 0; r < REP; ++r) for (int i = 0; i < 8; ++i) acc[r][i] = make_float4(i, r, 0, threadIdx.x);
  float x[REP];
  for (int r = 0; r < REP; ++r) x[r] = threadIdx.x * 1e-7f + 0.5f + r;
  int idx = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 a = lt[(idx + i) & 511];
#pragma unroll
      for (int r = 0; r < REP; ++r) {
        acc[r][i].x = fmaf(x[r], a.x, acc[r][i].x); acc[r][i].y = fmaf(x[r], a.y, acc[r][i].y);
        acc[r][i].z = fmaf(x[r], a.z, acc[r][i].z); acc[r][i].w = fmaf(x[r], a.w, acc[r][i].w);
      }
    }
    idx += 8;
  }
  float s = 0; for (int r = 0; r < REP; ++r) for (int i = 0; i < 8; ++i) s += acc[r][i].x + acc[r][i].y + acc[r][i].z + acc[r][i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// F: fp32 MFMA 32x32x2
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float seed) {
  v16f acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = i + j;
  float a = seed + threadIdx.x * 1e-6f, b = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0; for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F launch, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float *out, *tab;
  CK(hipMalloc(&out, 64 << 20));
  CK(hipMalloc(&tab, 1 << 20));
  std::vector<float> h(1 << 18); for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3f * (i % 97);
  CK(hipMemcpy(tab, h.data(), 1 << 20, hipMemcpyHostToDevice));
  const int iters = 2000;
  for (int wgs_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * wgs_per_cu;  // one 4-wave workgroup per SIMD set
    const double fl = 2.0 * NACC * (double)iters * 256.0 * grid;
    float t;
    t = timeit([&] { hipLaunchKernelGGL(k_fma_vgpr, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
    printf("wg/cu %d  fma vgpr          : %7.3f ms  %6.1f TF\n", wgs_per_cu, t, fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fma_sgpr, dim3(grid), dim3(256), 0, 0, out, iters, tab); });
    printf("wg/cu %d  fma sgpr (hoisted): %7.3f ms  %6.1f TF\n", wgs_per_cu, t, fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fma_sload<1>, dim3(grid), dim3(256), 0, 0, out, iters, tab, 0); });
    printf("wg/cu %d  fma sload rep1 same: %7.3f ms  %6.1f TF\n", wgs_per_cu, t, fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fma_sload<1>, dim3(grid), dim3(256), 0, 0, out, iters, tab, 2); });
    printf("wg/cu %d  fma sload rep1 walk: %7.3f ms  %6.1f TF\n", wgs_per_cu, t, fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fma_sload<2>, dim3(grid), dim3(256), 0, 0, out, iters, tab, 2); });
    printf("wg/cu %d  fma sload rep2 walk: %7.3f ms  %6.1f TF\n", wgs_per_cu, t, 2 * fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fma_sload<4>, dim3(grid), dim3(256), 0, 0, out, iters, tab, 2); });
    printf("wg/cu %d  fma sload rep4 walk: %7.3f ms  %6.1f TF\n", wgs_per_cu, t, 4 * fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_pkfma_vgpr, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
    printf("wg/cu %d  pk_fma vgpr       : %7.3f ms  %6.1f TF\n", wgs_per_cu, t, fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fma_lds<1>, dim3(grid), dim3(256), 0, 0, out, iters, tab); });
    printf("wg/cu %d  fma lds-bcast rep1: %7.3f ms  %6.1f TF\n", wgs_per_cu, t, fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fma_lds<2>, dim3(grid), dim3(256), 0, 0, out, iters, tab); });
    printf("wg/cu %d  fma lds-bcast rep2: %7.3f ms  %6.1f TF\n", wgs_per_cu, t, 2 * fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fma_lds<4>, dim3(grid), dim3(256), 0, 0, out, iters, tab); });
    printf("wg/cu %d  fma lds-bcast rep4: %7.3f ms  %6.1f TF\n", wgs_per_cu, t, 4 * fl / t / 1e9);
    const double mf = 4.0 * 2 * 32 * 32 * 2 * (double)iters * 4 * grid;
    t = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
    printf("wg/cu %d  mfma f32 32x32x2  : %7.3f ms  %6.1f TF\n", wgs_per_cu, t, mf / t / 1e9);
  }
  return 0;
}
