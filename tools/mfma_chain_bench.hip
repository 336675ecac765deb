// Issue cost of DEPENDENT fp32 MFMA chains on gfx950 (D of one is C of the next) vs independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_chain_bench tools/mfma_chain_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(uint64_t* out, float* sink, int n, float a, float b) {
  v4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) {          // 8 MFMA 16x16x4, one chain
#pragma unroll
      for (int r = 0; r < 8; ++r) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    } else if (MODE == 1) {   // 8 MFMA 16x16x4, two chains interleaved
#pragma unroll
      for (int r = 0; r < 4; ++r) { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0); }
    } else if (MODE == 2) {   // 8 MFMA 4x4x1, one chain
#pragma unroll
      for (int r = 0; r < 8; ++r) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    } else if (MODE == 3) {   // 8 MFMA 4x4x1, two chains
#pragma unroll
      for (int r = 0; r < 4; ++r) { c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0); }
    } else if (MODE == 4) {   // 8 MFMA 4x4x1, four chains
#pragma unroll
      for (int r = 0; r < 2; ++r) { c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0); }
    } else if (MODE == 5) {   // chains of 5 (like one stage-2 coordinate tile), fresh accumulator each time
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v4 c = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 5; ++m) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + m, c, 0, 0, 0);
        c1 += c;
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0];
}

template <int MODE>
void run(const char* what, int per_iter, uint64_t* out, float* sink) {
  const int n = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256 * wps), 0, 0, out, sink, n, 1.f, 0.5f);
    uint64_t h[32];
    (void)hipMemcpy(h, out, 8 * 2 * 4 * wps, hipMemcpyDeviceToHost);
    uint64_t lo = ~0ull, hi = 0;
    for (int q = 0; q < 4 * wps; ++q) { if (h[2 * q] < lo) lo = h[2 * q]; if (h[2 * q + 1] > hi) hi = h[2 * q + 1]; }
    printf("%-44s %d wave/SIMD: %6.2f clk per MFMA\n", what, wps, (double)(hi - lo) / ((double)n * per_iter * wps));
  }
}

int main() {
  uint64_t* out; float* sink;
  (void)hipMalloc(&out, 8 * 64); (void)hipMalloc(&sink, 4 * 1024);
  run<0>("16x16x4 one dependent chain", 8, out, sink);
  run<1>("16x16x4 two chains", 8, out, sink);
  run<5>("16x16x4 chains of 5 + v_pk_add", 40, out, sink);
  run<2>("4x4x1 one dependent chain", 8, out, sink);
  run<3>("4x4x1 two chains", 8, out, sink);
  run<4>("4x4x1 four chains", 8, out, sink);
  return 0;
}
