// How far is the Gumbel transform with the hardware logarithm (v_log_f32) from the one with logf, over EVERY uniform the generators can
// draw (the 2^24 lattice of u01)?  Decides the margin under which the hard gate's arg-max may be taken from the fast transform
// (ktup_eval.hip: gumbel_argmax) and must be recomputed with logf otherwise.
//   hipcc --offload-arch=gfx950 -O2 -o tools/gumbel_log_check tools/gumbel_log_check.hip && tools/gumbel_log_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__device__ float g_exact(float u) { return -logf(-logf(u + 1e-20f) + 1e-20f); }
__device__ float g_fast(float u) {
  const float ln2 = 0.69314718055994530942f;
  const float t = -ln2 * __builtin_amdgcn_logf(u + 1e-20f) + 1e-20f;
  return -ln2 * __builtin_amdgcn_logf(t);
}
__global__ void k(float* maxabs, unsigned* where, float* maxrel_inner) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  const float u = (float)i * (1.0f / 16777216.0f);
  const float a = g_exact(u), b = g_fast(u);
  float d = fabsf(a - b);
  if (!(d == d)) d = (a == b || (a != a && b != b)) ? 0.f : 1e30f;       // inf / nan on both sides agree
  if (isinf(a) && isinf(b) && a == b) d = 0.f;
  const float ti = -logf(u + 1e-20f), tf = -0.69314718055994530942f * __builtin_amdgcn_logf(u + 1e-20f);
  const float r = ti != 0.f ? fabsf(ti - tf) / fabsf(ti) : (tf == 0.f ? 0.f : 1e30f);
  // block max via atomics on the bit patterns (non-negative floats order like ints)
  atomicMax(reinterpret_cast<unsigned*>(maxabs), __float_as_uint(d));
  if (d > 1e-4f) atomicMin(where, i);
  atomicMax(reinterpret_cast<unsigned*>(maxrel_inner), __float_as_uint(r));
}
int main() {
  float *d0, *d2; unsigned* d1;
  hipMalloc(&d0, 4); hipMalloc(&d1, 4); hipMalloc(&d2, 4);
  hipMemset(d0, 0, 4); hipMemset(d1, 0xff, 4); hipMemset(d2, 0, 4);
  hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, d0, d1, d2);
  float m, r; unsigned w;
  hipMemcpy(&m, d0, 4, hipMemcpyDeviceToHost); hipMemcpy(&w, d1, 4, hipMemcpyDeviceToHost); hipMemcpy(&r, d2, 4, hipMemcpyDeviceToHost);
  printf("max |g_fast - g_logf| over the 2^24 uniforms: %.3e   (first u index with a difference > 1e-4: %u)\n", m, w);
  printf("max relative difference of the inner -log(u): %.3e\n", r);
  return 0;
}
