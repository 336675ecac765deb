// Repro of the fault DESIGN.md §8 records: a memset NODE captured in a HIP graph stopped clearing its buffer after the first replay
// (seen in round 2 when the sampler's scratch clear was a hipMemsetAsync inside the step's captured graph; no graph of the library
// has a memset node since).  Isolated from the library and from torch:
//   graph = [ memset(A, 0) -> kernel A[i] += 1 ]   replayed R times; A[i] must be 1 after every replay.
// Result (profiles/r03_graph_memset_repro.txt): with the HIP runtime that torch 2.10+rocm7.0 BUNDLES (torch/lib/libamdhip64.so, runtime
// version 70051831) every variant whose destination lies at a non-zero offset INSIDE an allocation -- every tensor of a caching
// allocator -- fails from the SECOND replay on: the node then writes the low 32 bits of the allocation's base address instead of the
// value; capture mode, instantiate flags, the replay stream and eager memsets between replays do not matter, and a destination that
// is the start of its own allocation always works.  With the image's own ROCm 7.2 runtime (/opt/rocm/lib) no variant fails.
//   hipcc --offload-arch=gfx950 -O2 -o tools/graph_memset_repro tools/graph_memset_repro.hip
//   tools/graph_memset_repro                                                   # the runtime the binary links: /opt/rocm
//   LD_PRELOAD=<site-packages>/torch/lib/libamdhip64.so tools/graph_memset_repro   # the runtime every torch process uses
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

__global__ void inc_kernel(int* a, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += 1;
}

enum Eager { NONE, SAME_BUF_SAME_STREAM, OTHER_BUF_SAME_STREAM, OTHER_BUF_OTHER_STREAM, OTHER_BUF_NULL_STREAM, SYNC_MEMSET_OTHER_BUF };
static const char* eager_name[] = {"no eager memset", "eager memset of the SAME buffer, same stream", "eager memset of another buffer, same stream",
                                   "eager memset of another buffer, other stream", "eager memset of another buffer, null stream",
                                   "synchronous hipMemset of another buffer"};

static int run(size_t n, Eager eager, bool d32, bool stream_capture, int replays, bool destroy_early = false, bool autofree = false,
               bool global_mode = false, int replay_on = 0, size_t sub_offset = 0) {
  int *A, *B, *A0;
  CK(hipMalloc(&A0, (n + sub_offset) * sizeof(int)));
  A = A0 + sub_offset;                            // the node's destination lies INSIDE an allocation (a caching allocator's block)
  CK(hipMalloc(&B, n * sizeof(int)));
  CK(hipMemset(A, 0x7f, n * sizeof(int)));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipGraph_t g;
  hipGraphExec_t ge;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (stream_capture) {
    CK(hipStreamBeginCapture(s, global_mode ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
    if (d32) CK(hipMemsetD32Async((hipDeviceptr_t)A, 0, n, s)); else CK(hipMemsetAsync(A, 0, n * sizeof(int), s));
    hipLaunchKernelGGL(inc_kernel, dim3(blocks), dim3(256), 0, s, A, n);
    CK(hipStreamEndCapture(s, &g));
  } else {
    CK(hipGraphCreate(&g, 0));
    hipMemsetParams mp{};
    mp.dst = A; mp.value = 0; mp.pitch = 0; mp.elementSize = d32 ? 4 : 1; mp.width = d32 ? n : n * sizeof(int); mp.height = 1;
    hipGraphNode_t mn, kn;
    CK(hipGraphAddMemsetNode(&mn, g, nullptr, 0, &mp));
    hipKernelNodeParams kp{};
    void* args[] = {&A, &n};
    kp.func = (void*)inc_kernel; kp.gridDim = dim3(blocks); kp.blockDim = dim3(256); kp.kernelParams = args;
    CK(hipGraphAddKernelNode(&kn, g, &mn, 1, &kp));
  }
  if (autofree) CK(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch));
  else CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  if (destroy_early) CK(hipGraphDestroy(g));     // what torch.cuda.CUDAGraph.capture_end does: only the executable graph is kept
  std::vector<int> host(n);
  int bad_replay = -1, bad_value = 0;
  for (int r = 0; r < replays && bad_replay < 0; ++r) {
    hipStream_t rs = replay_on == 0 ? s : (replay_on == 1 ? s2 : (hipStream_t)0);   // the capture stream, another one, the null stream
    CK(hipGraphLaunch(ge, rs));
    switch (eager) {
      case SAME_BUF_SAME_STREAM: break;   // issued after the check below (it would hide the result)
      case OTHER_BUF_SAME_STREAM: CK(hipMemsetAsync(B, 0, n * sizeof(int), s)); break;
      case OTHER_BUF_OTHER_STREAM: CK(hipMemsetAsync(B, 0, n * sizeof(int), s2)); break;
      case OTHER_BUF_NULL_STREAM: CK(hipMemsetAsync(B, 0, n * sizeof(int), 0)); break;
      case SYNC_MEMSET_OTHER_BUF: CK(hipMemset(B, 0, n * sizeof(int))); break;
      default: break;
    }
    CK(hipStreamSynchronize(rs));
    CK(hipMemcpy(host.data(), A, n * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i)
      if (host[i] != 1) { bad_replay = r; bad_value = host[i]; break; }
    if (eager == SAME_BUF_SAME_STREAM) {  // leave A dirty through an EAGER memset to a non-zero pattern + increments: the node must clear it
      CK(hipMemsetAsync(A, 0x01, n * sizeof(int), s));
      hipLaunchKernelGGL(inc_kernel, dim3(blocks), dim3(256), 0, s, A, n);
    }
    CK(hipDeviceSynchronize());
  }
  CK(hipGraphExecDestroy(ge)); if (!destroy_early) CK(hipGraphDestroy(g));
  CK(hipStreamDestroy(s)); CK(hipStreamDestroy(s2));
  CK(hipFree(A0)); CK(hipFree(B));
  if (bad_replay >= 0) { printf("FAIL (replay %d: A[i] = %d)", bad_replay, bad_value); return 1; }
  printf("OK");
  return 0;
}

int main() {
  int fails = 0, rv = 0, dv = 0;
  CK(hipRuntimeGetVersion(&rv)); CK(hipDriverGetVersion(&dv));
  printf("HIP runtime version %d, driver version %d\n", rv, dv);
  const size_t sizes[] = {1, 1024, 262144, 16u << 20};
  for (int cap = 1; cap >= 0; --cap)
    for (int d32 = 0; d32 < 2; ++d32)
      for (size_t n : sizes)
        for (int e = NONE; e <= SYNC_MEMSET_OTHER_BUF; ++e) {
          printf("%-16s %-4s n=%-9zu %-48s : ", cap ? "stream capture" : "explicit node", d32 ? "D32" : "D8", n, eager_name[e]);
          fails += run(n, (Eager)e, d32 != 0, cap != 0, 50);
          printf("\n");
        }
  // torch.cuda.graph's way: global capture mode, AutoFreeOnLaunch, the hipGraph_t destroyed right after instantiation
  for (int de = 0; de < 2; ++de)
    for (int af = 0; af < 2; ++af)
      for (int gm = 0; gm < 2; ++gm)
        for (size_t n : {(size_t)1024, (size_t)(1 << 20)}) {
          printf("stream capture D8 n=%-8zu graph destroyed after instantiate=%d autofree=%d global capture mode=%d : ", n, de, af, gm);
          fails += run(n, NONE, false, true, 20, de != 0, af != 0, gm != 0);
          printf("\n");
        }
  for (int ro = 0; ro < 3; ++ro)
    for (int e = NONE; e <= SYNC_MEMSET_OTHER_BUF; ++e) {
      printf("stream capture D8 n=1024 replayed on %-18s %-48s : ", ro == 0 ? "the capture stream" : (ro == 1 ? "another stream" : "the null stream"), eager_name[e]);
      fails += run(1024, (Eager)e, false, true, 20, true, true, true, ro);
      printf("\n");
    }
  for (size_t off : {(size_t)1024, (size_t)(512 * 1024)})
    for (int e = NONE; e <= SYNC_MEMSET_OTHER_BUF; ++e) {
      printf("stream capture D8 n=1024 at offset %zu ints inside its allocation, null stream, %-48s : ", off, eager_name[e]);
      fails += run(1024, (Eager)e, false, true, 20, true, true, true, 2, off);
      printf("\n");
    }
  // which of the other circumstances matter once the destination is a sub-buffer: none of them
  for (int cap = 1; cap >= 0; --cap)
    for (int d32 = 0; d32 < 2; ++d32)
      for (int ro = 0; ro < 3; ++ro) {
        printf("%-16s %-4s n=1024 at offset 1024 ints, default instantiate, graph kept, replayed on %-18s : ", cap ? "stream capture" : "explicit node",
               d32 ? "D32" : "D8", ro == 0 ? "the capture stream" : (ro == 1 ? "another stream" : "the null stream"));
        fails += run(1024, NONE, d32 != 0, cap != 0, 20, false, false, false, ro, 1024);
        printf("\n");
      }
  printf("%d failing variant(s)\n", fails);
  return 0;
}
