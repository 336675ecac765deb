// Issue cost (clk per wave-instruction on one SIMD) of the integer / packed ops the gather address path can be built from,
// and of the small MFMA shapes.  One wave per SIMD, long dependent-free streams, s_memtime around them.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate_bench tools/valu_rate_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define BODY(NAME, ASM8)                                                              \
  __global__ void NAME(uint64_t* out, int n) {                                        \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7; \
    v2 p0 = {1.f, 2.f}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0; \
    v4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;                                  \
    float f = 1.0f;                                                                   \
    uint64_t t0 = __builtin_readcyclecounter();                                       \
    for (int i = 0; i < n; ++i) { ASM8 }                                              \
    uint64_t t1 = __builtin_readcyclecounter();                                       \
    if ((threadIdx.x & 63) == 0) { out[100 + 2 * (threadIdx.x >> 6)] = t0; out[101 + 2 * (threadIdx.x >> 6)] = t1; } \
    out[1 + (threadIdx.x & 63)] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + (uint64_t)(p0.x + p1.x + p2.y + p3.x + p4.x + p5.x + p6.x + p7.x + c0[0] + c1[0] + c2[0] + c3[0]); \
  }

BODY(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(n));)
BODY(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %8, %8\n v_mad_u32_u24 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mad_u32_u24 %3, %3, %8, %8\n v_mad_u32_u24 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mad_u32_u24 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(n));)
BODY(k_mad_u64, asm volatile("v_mad_u64_u32 %0, vcc, %8, %8, %0\n v_mad_u64_u32 %1, vcc, %8, %8, %1\n v_mad_u64_u32 %2, vcc, %8, %8, %2\n v_mad_u64_u32 %3, vcc, %8, %8, %3\n v_mad_u64_u32 %4, vcc, %8, %8, %4\n v_mad_u64_u32 %5, vcc, %8, %8, %5\n v_mad_u64_u32 %6, vcc, %8, %8, %6\n v_mad_u64_u32 %7, vcc, %8, %8, %7" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(n) : "vcc");)
BODY(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 4, %0\n v_lshl_add_u64 %1, %1, 4, %1\n v_lshl_add_u64 %2, %2, 4, %2\n v_lshl_add_u64 %3, %3, 4, %3\n v_lshl_add_u64 %4, %4, 4, %4\n v_lshl_add_u64 %5, %5, 4, %5\n v_lshl_add_u64 %6, %6, 4, %6\n v_lshl_add_u64 %7, %7, 4, %7" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));)
BODY(k_add_u32, asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(n));)
BODY(k_pk_add, asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n v_pk_add_f32 %4, %4, %4\n v_pk_add_f32 %5, %5, %5\n v_pk_add_f32 %6, %6, %6\n v_pk_add_f32 %7, %7, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
BODY(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
BODY(k_mfma_16x16x4, c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, f, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, f, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, f, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, f, c3, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, f, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, f, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, f, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, f, c3, 0, 0, 0);)
BODY(k_mfma_4x4x1, c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, f, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, f, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, f, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, f, c3, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, f, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, f, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, f, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, f, c3, 0, 0, 0);)

#define RUN(NAME)                                                                                 \
  {                                                                                               \
    printf("%-16s", #NAME);                                                                        \
    for (int wps = 1; wps <= 4; wps *= 2) {                                                        \
      hipLaunchKernelGGL(NAME, dim3(1), dim3(256 * wps), 0, 0, out, n);                            \
      hipLaunchKernelGGL(NAME, dim3(1), dim3(256 * wps), 0, 0, out, n);                            \
      (void)hipMemcpy(hh, out + 100, 8 * 2 * 4 * wps, hipMemcpyDeviceToHost);                       \
      { uint64_t lo = ~0ull, hi = 0; for (int q = 0; q < 4 * wps; ++q) { if (hh[2 * q] < lo) lo = hh[2 * q]; if (hh[2 * q + 1] > hi) hi = hh[2 * q + 1]; } h = hi - lo; } \
      printf("  %d wave/SIMD: %6.2f clk per SIMD instruction", wps, (double)h / (8.0 * n * wps));   \
    }                                                                                              \
    printf("\n");                                                                                  \
  }

int main() {
  uint64_t* out; uint64_t h = 0; uint64_t hh[64];
  (void)hipMalloc(&out, 8 * 200);
  const int n = 100000;
  RUN(k_add_u32) RUN(k_mul_lo) RUN(k_mad_u24) RUN(k_mad_u64) RUN(k_lshl_add_u64) RUN(k_pk_add) RUN(k_pk_fma) RUN(k_mfma_16x16x4) RUN(k_mfma_4x4x1)
  return 0;
}
