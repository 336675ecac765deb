// Row-gather microbenchmark for gfx950: what bounds the scattered 400-byte row gathers the scoring kernels live on?
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_bench tools/gather_bench.hip ; tools/gather_bench
// Each wave gathers tiles of R rows (d floats each, row pitch `pitch` floats) by random ids and writes one float per lane
// per tile (a checksum, so that the loads stay live).  Sweeps: vector width, loads in flight, waves per CU, table size,
// pitch alignment, nontemporal hint.  Reports GB/s of useful row bytes and B/clk/CU at 2.4 GHz over 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <cstdint>

template <int VEC> struct V;
template <> struct V<1> { typedef float T; };
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
template <> struct V<2> { typedef vf2 T; };
template <> struct V<4> { typedef vf4 T; };
__device__ inline float hsum(float a) { return a; }
__device__ inline float hsum(vf2 a) { return a.x + a.y; }
__device__ inline float hsum(vf4 a) { return a.x + a.y + a.z + a.w; }

// J loads in flight per lane; a wave's tile = R rows with R * cpr <= J * 64 (cpr = chunks per row).  Like the scoring
// kernels: ids of the tile sit in LDS (the next tile's ids are fetched a tile ahead), 32-bit incremental (row, chunk).
template <int VEC, int J, bool NT>
__global__ void gather_kernel(const float* __restrict__ table, int64_t pitch, int d, const int32_t* __restrict__ ids,
                              int64_t nrows, float* __restrict__ out) {
  typedef typename V<VEC>::T T;
  __shared__ int32_t sid_all[16][64];
  const int cpr = d / VEC;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int32_t* sid = sid_all[w];
  const int R = J * 64 / cpr > 64 ? 64 : J * 64 / cpr;          // rows per tile
  const int total = R * cpr;
  const int qstep = 64 / cpr, rstep = 64 - qstep * cpr;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t ntiles = nrows / R;
  float acc = 0.f;
  int32_t nx = (wave < ntiles && lane < R) ? ids[wave * R + lane] : 0;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    if (lane < R) sid[lane] = nx;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int v = lane, row = lane / cpr, c = lane - (lane / cpr) * cpr;
    asm volatile("" : "+v"(v), "+v"(row), "+v"(c));
    T x[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      if (v < total) {
        const T* p = reinterpret_cast<const T*>(table + (int64_t)sid[row] * pitch) + c;
        if (NT) x[j] = __builtin_nontemporal_load(p); else x[j] = *p;
      } else {
        x[j] = T(0.f);
      }
      v += 64; row += qstep; c += rstep;
      if (c >= cpr) { c -= cpr; ++row; }
    }
    if (tile + nwaves < ntiles && lane < R) nx = ids[(tile + nwaves) * R + lane];
#pragma unroll
    for (int j = 0; j < J; ++j) acc += hsum(x[j]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int VEC, int J, bool NT>
float run(const float* table, int64_t pitch, int d, const int32_t* ids, int64_t nrows, float* out, int wg, int wpc) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 256 * wg;
  const int block = 64 * wpc / wg;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gather_kernel<VEC, J, NT>), dim3(grid), dim3(block), 0, 0, table, pitch, d, ids, nrows, out);
  hipEventRecord(a);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather_kernel<VEC, J, NT>), dim3(grid), dim3(block), 0, 0, table, pitch, d, ids, nrows, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

// `footprint`: ONE 9.7 GB (d = 100) / 24.8 GB (d = 256) allocation, the same 2,150,400 random rows drawn from a leading window of it
// of growing size.  Every row is its own DRAM page miss at all of these footprints, so what separates them is (a) the 256 MB
// Infinity Cache (gone by ~1 GB) and (b) address translation reach; a rate that keeps falling from 2.4 GB to the full table is
// translation, a flat one is the HBM's random-row rate itself.
template <int D>
void footprint_sweep() {
  const int64_t nrows = 2150400, trows = 24248000;
  float* table; int32_t* ids; float* out;
  if (hipMalloc(&table, trows * D * 4) != hipSuccess) { printf("d=%d: allocation of %.1f GB failed\n", D, trows * D * 4 / 1e9); return; }
  hipMemset(table, 0, trows * D * 4);
  hipMalloc(&ids, nrows * 4);
  hipMalloc(&out, 256 * 2048 * 4);
  std::vector<int32_t> h(nrows);
  for (double frac : {0.0423, 0.125, 0.25, 0.5, 1.0}) {
    const int64_t window = (int64_t)(trows * frac);
    uint64_t s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (int32_t)(s % (uint64_t)window); }
    hipMemcpy(ids, h.data(), nrows * 4, hipMemcpyHostToDevice);
    const double bytes = (double)nrows * D * 4;
    for (int nt = 0; nt < 2; ++nt) {
      const float ms = D == 100 ? (nt ? run<4, 21, true>(table, D, D, ids, nrows, out, 1, 16) : run<4, 21, false>(table, D, D, ids, nrows, out, 1, 16))
                                : (nt ? run<4, 16, true>(table, D, D, ids, nrows, out, 1, 16) : run<4, 16, false>(table, D, D, ids, nrows, out, 1, 16));
      printf("d=%3d rows of %4d B  window %6.2f GB of %5.1f GB  %-12s %8.1f us  %7.1f GB/s of row bytes  (%4.1f %% of 8 TB/s)\n", D, D * 4,
             window * (double)D * 4 / 1e9, trows * (double)D * 4 / 1e9, nt ? "nontemporal" : "default", ms * 1e3, bytes / ms / 1e6,
             bytes / ms / 1e6 / 80.0);
    }
  }
  hipFree(table); hipFree(ids); hipFree(out);
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "footprint") {
    footprint_sweep<100>();
    footprint_sweep<256>();
    return 0;
  }
  const bool big = argc > 1 && std::string(argv[1]) == "big";   // working sets beyond L2 + Infinity Cache only
  const int d = 100;
  const int64_t nrows = 2150400;   // 3 x 716800 row reads, like one KTUP launch
  struct Cfg { const char* name; int64_t trows; int64_t pitch; };
  const Cfg cfgs[] = {{"1.3MB p100", 3240, 100}, {"9.7MB p100", 24248, 100}, {"9.7MB p128", 24248, 128}, {"410MB p100", 1024000, 100},
                      {"9.7GB p100", 24248000, 100}, {"12.4GB p128", 24248000, 128}};
  for (const Cfg& c : cfgs) {
    if (big ? c.trows < 1024000 : c.trows > 1024000) continue;
    float* table; int32_t* ids; float* out;
    hipMalloc(&table, c.trows * c.pitch * 4);
    hipMemset(table, 0, c.trows * c.pitch * 4);
    std::vector<int32_t> h(nrows);
    uint64_t s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (int32_t)(s % (uint64_t)c.trows); }
    hipMalloc(&ids, nrows * 4);
    hipMemcpy(ids, h.data(), nrows * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 2048 * 4);
    const double bytes = (double)nrows * d * 4;
    auto rep = [&](const char* what, int wpc, float ms) {
      printf("%-12s %-22s waves/CU %2d  %8.1f us  %7.1f GB/s  %5.1f B/clk/CU\n", c.name, what, wpc, ms * 1e3, bytes / ms / 1e6,
             bytes / (ms * 1e-3) / 2.4e9 / 256);
    };
    for (int wpc : {8, 16, 32}) {
      const int wg = wpc >= 16 ? wpc / 16 : 1;    // workgroups per CU (<= 1024 threads each)
      rep("x4 J=7", wpc, run<4, 7, false>(table, c.pitch, d, ids, nrows, out, wg, wpc));
      rep("x4 J=14", wpc, run<4, 14, false>(table, c.pitch, d, ids, nrows, out, wg, wpc));
      rep("x4 J=21", wpc, run<4, 21, false>(table, c.pitch, d, ids, nrows, out, wg, wpc));
      rep("x4 J=21 nontemporal", wpc, run<4, 21, true>(table, c.pitch, d, ids, nrows, out, wg, wpc));
      rep("x2 J=28", wpc, run<2, 28, false>(table, c.pitch, d, ids, nrows, out, wg, wpc));
      rep("x1 J=32", wpc, run<1, 32, false>(table, c.pitch, d, ids, nrows, out, wg, wpc));
    }
    hipFree(table); hipFree(ids); hipFree(out);
  }
  return 0;
}
