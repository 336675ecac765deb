// Do MFMA and VALU instructions overlap on a gfx950 SIMD (a) inside one wave, (b) across two waves of the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o tools/coissue_bench tools/coissue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));

#define MFMA4(A, B)                                                      \
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, c0, 0, 0, 0);          \
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, c1, 0, 0, 0);          \
  c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, c2, 0, 0, 0);          \
  c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, c3, 0, 0, 0);
#define FMA7(k)                                                                                      \
  f0 = fmaf(f0, k, 1.f); f1 = fmaf(f1, k, 1.f); f2 = fmaf(f2, k, 1.f); f3 = fmaf(f3, k, 1.f);       \
  f4 = fmaf(f4, k, 1.f); f5 = fmaf(f5, k, 1.f); f6 = fmaf(f6, k, 1.f);

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int n, float a, float b) {
  v4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6;
  const int slot = threadIdx.x >> 8;     // waves 0-3 -> slot 0, waves 4-7 -> slot 1 (round-robin SIMD assignment)
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && slot == 0);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && slot == 1);
  if (MODE == 2) {
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); FMA7(b)
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0); FMA7(b)
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); FMA7(b)
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0); FMA7(b)
    }
  } else if (do_mfma) {
    for (int i = 0; i < n; ++i) { MFMA4(a, b) }
  } else if (do_valu) {
    for (int i = 0; i < n; ++i) { FMA7(b) FMA7(b) FMA7(b) FMA7(b) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6;
}

template <int MODE>
float run(float* out, int block, int n) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, out, n, 1.0f, 0.5f);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, out, n, 1.0f, 0.5f);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  const int n = 20000;     // 4 MFMA (+ 28 FMA) per iteration
  printf("per iteration: 4 x v_mfma_f32_16x16x4_f32 (4 x 32 clk) and/or 28 x v_fma_f32 (28 x 4 clk), n=%d\n", n);
  printf("1 wave/SIMD  MFMA only          %8.1f us\n", run<0>(out, 256, n));
  printf("1 wave/SIMD  VALU only          %8.1f us\n", run<1>(out, 256, n));
  printf("1 wave/SIMD  interleaved        %8.1f us\n", run<2>(out, 256, n));
  printf("2 waves/SIMD MFMA only (both)   %8.1f us\n", run<0>(out, 512, n));
  printf("2 waves/SIMD VALU only (both)   %8.1f us\n", run<1>(out, 512, n));
  printf("2 waves/SIMD one MFMA, one VALU %8.1f us\n", run<3>(out, 512, n));
  printf("2 waves/SIMD interleaved (both) %8.1f us\n", run<2>(out, 512, n));
  return 0;
}
