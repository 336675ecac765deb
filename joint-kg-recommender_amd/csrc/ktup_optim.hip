// K20: global-norm gradient clip + dense optimizer step over a LIST of tables: two launches
// (ktup_optim_gradnorm, ktup_optim_step), or ONE with a grid-wide barrier between the norm and the update (ktup_optim_clip_step).
//
// Reference: every driver ends its step with  clip_grad_norm([all params], clipping_max_value); optimizer.step()
// (item_recommendation.py:189-192, knowledge_representation.py:209-211, knowledgable_recommendation.py:399-401) on one of
// torch.optim.{Adagrad, Adam, SGD(momentum), RMSprop(momentum)} with weight_decay = l2_lambda (utils/trainer.py:63-77).
// torch runs that as ~5 multi-tensor passes per optimizer plus 3 for the clip, each re-reading every table.  Here:
//   launch 1  ktup_optim_gradnorm : sum of squares of all gradients -> one device double
//   launch 2  ktup_optim_step     : g <- g * min(1, max_norm / (||g|| + 1e-6))  [written back, like clip_grad_norm_],
//                                   weight decay, state update, parameter update; one read and one write per array.
// Arithmetic follows torch.optim's single-tensor formulas operation by operation (fp32), so results agree to rounding.
// All tables are flattened into one index space; a workgroup-sized chunk never straddles two tables.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ktup_common.h"

namespace {

using namespace ktup;

constexpr int MAXT = KTUP_OPTIM_MAX_TENSORS;
constexpr int CHUNK = 256 * 4 * 4;   // floats per workgroup iteration: 256 threads x 4 float4

struct OptTensors {
  float* p[MAXT];
  float* g[MAXT];
  float* s1[MAXT];
  float* s2[MAXT];
  int64_t chunk0[MAXT + 1];   // first chunk of each tensor in the flattened chunk space
  int64_t n[MAXT];
  float bc1[MAXT], bc2s[MAXT];   // Adam: 1 - beta1^t, sqrt(1 - beta2^t) of each tensor's own step count
  int first[MAXT];               // SGD / RMSprop momentum: buffer not initialised yet (torch clones the gradient)
  int count;
};

KTUP_DEV int find_tensor(const OptTensors& T, int64_t chunk) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < MAXT; ++i) k += (i < T.count && chunk >= T.chunk0[i]) ? 1 : 0;
  return k;
}

__global__ __launch_bounds__(256) void gradnorm_kernel(OptTensors T, double* __restrict__ sumsq, int slots) {
  // work unit = a quarter chunk (256 float4): four loads in flight per thread
  const int64_t nunits = T.chunk0[T.count] * 4;
  float acc = 0.f;
  for (int64_t u0 = blockIdx.x; u0 < nunits; u0 += (int64_t)gridDim.x * 4) {
    float4 v[4];
    int64_t rest_i[4], rest_n[4];
    const float* rest_g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t unit = u0 + (int64_t)q * gridDim.x;
      v[q] = f4zero(); rest_g[q] = nullptr; rest_i[q] = 0; rest_n[q] = 0;
      if (unit < nunits) {
        const int64_t chunk = unit >> 2;
        const int k = find_tensor(T, chunk);
        const float* g = T.g[k];
        const int64_t n = T.n[k];
        const int64_t i = (chunk - T.chunk0[k]) * CHUNK + ((unit & 3) * 256 + threadIdx.x) * 4;
        if (i + 3 < n) v[q] = *reinterpret_cast<const float4*>(g + i);
        else { rest_g[q] = g; rest_i[q] = i; rest_n[q] = n; }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc += dot4(v[q], v[q]);
      for (int64_t e = rest_i[q]; e < rest_n[q]; ++e) acc = fmaf(rest_g[q][e], rest_g[q][e], acc);
    }
  }
  acc = group_sum<64>(acc);                    // <= 64 x (a few chunks x 16) fp32 terms per wave; the cross-workgroup sum is fp64
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  // one double atomic per workgroup; callers with many workgroups spread them over `slots` words (~20 ns each on ONE address)
  if (threadIdx.x == 0) atomicAdd(sumsq + (slots > 1 ? blockIdx.x % slots : 0), ((double)red[0] + (double)red[1]) + ((double)red[2] + (double)red[3]));
}

struct Hyper {
  float lr, wd, momentum, beta1, beta2, eps, alpha;
  float max_norm;   // <= 0: no clipping
  int zero_grads;   // write 0 to the gradients instead of their clipped values (the next step starts from zero-filled grads)
};

template <int KIND>
KTUP_DEV void update1(float& p, float& g, float& s1, float& s2, const Hyper& h, float coef, float bc1, float bc2s, bool first) {
  g *= coef;                                   // clip_grad_norm_ scales .grad in place (also by 1.0)
  float d = fmaf(h.wd, p, g);                  // grad.add(param, alpha=weight_decay)
  if (h.wd == 0.f) d = g;
  if (KIND == KTUP_OPT_SGD) {                  // torch/optim/sgd.py _single_tensor_sgd (dampening 0, no nesterov)
    if (h.momentum != 0.f) {
      s1 = first ? d : fmaf(h.momentum, s1, d);
      d = s1;
    }
    p = fmaf(-h.lr, d, p);
  } else if (KIND == KTUP_OPT_ADAGRAD) {       // adagrad.py: lr_decay 0 -> clr = lr; eps 1e-10
    s1 = fmaf(d, d, s1);
    p = p - h.lr * (d / (sqrtf(s1) + h.eps));
  } else if (KIND == KTUP_OPT_ADAM) {          // adam.py _single_tensor_adam (amsgrad off)
    s1 = s1 + (d - s1) * (1.f - h.beta1);      // exp_avg.lerp_(grad, 1 - beta1)
    s2 = fmaf(1.f - h.beta2, d * d, h.beta2 * s2);
    const float denom = sqrtf(s2) / bc2s + h.eps;
    p = p - (h.lr / bc1) * (s1 / denom);
  } else {                                     // rmsprop.py (centered off)
    s1 = fmaf(1.f - h.alpha, d * d, h.alpha * s1);
    const float avg = sqrtf(s1) + h.eps;
    if (h.momentum > 0.f) {
      s2 = first ? d / avg : fmaf(h.momentum, s2, d / avg);   // buf.mul_(momentum).addcdiv_(grad, avg); buf starts at 0
      p = fmaf(-h.lr, s2, p);
    } else {
      p = p - h.lr * (d / avg);
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void step_kernel(OptTensors T, Hyper h, const double* __restrict__ sumsq,
                                                   const int64_t* __restrict__ steps_dev) {
  // Adam with device-resident step counts (graph replay): the bias corrections are evaluated here, the way the host does
  __shared__ float dev_bc1[MAXT], dev_bc2s[MAXT];
  const bool dev_bc = KIND == KTUP_OPT_ADAM && steps_dev != nullptr;
  if (dev_bc) {
    if ((int)threadIdx.x < T.count) {
      const double t = (double)steps_dev[threadIdx.x];
      dev_bc1[threadIdx.x] = (float)(1.0 - pow((double)h.beta1, t));
      dev_bc2s[threadIdx.x] = (float)sqrt(1.0 - pow((double)h.beta2, t));
    }
    __syncthreads();
  }
  float coef = 1.f;
  if (h.max_norm > 0.f) {
    const float c = h.max_norm / ((float)sqrt(*sumsq) + 1e-6f);
    coef = c < 1.f ? c : 1.f;
  }
  const int64_t nchunks = T.chunk0[T.count];
  for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int k = find_tensor(T, chunk);
    const int64_t base = (chunk - T.chunk0[k]) * CHUNK;
    float* p = T.p[k];
    float* g = T.g[k];
    float* s1 = T.s1[k];
    float* s2 = T.s2[k];
    const int64_t n = T.n[k];
    const float bc1 = dev_bc ? dev_bc1[k] : T.bc1[k], bc2s = dev_bc ? dev_bc2s[k] : T.bc2s[k];
    const bool first = T.first[k] != 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = base + ((int64_t)r * 256 + threadIdx.x) * 4;
      const int m = i + 3 < n ? 4 : (i < n ? (int)(n - i) : 0);
      if (m == 4) {
        float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<float4*>(g + i);
        float4 a = s1 ? *reinterpret_cast<float4*>(s1 + i) : f4zero(), b = s2 ? *reinterpret_cast<float4*>(s2 + i) : f4zero();
        update1<KIND>(pv.x, gv.x, a.x, b.x, h, coef, bc1, bc2s, first);
        update1<KIND>(pv.y, gv.y, a.y, b.y, h, coef, bc1, bc2s, first);
        update1<KIND>(pv.z, gv.z, a.z, b.z, h, coef, bc1, bc2s, first);
        update1<KIND>(pv.w, gv.w, a.w, b.w, h, coef, bc1, bc2s, first);
        *reinterpret_cast<float4*>(p + i) = pv;
        if (h.zero_grads) *reinterpret_cast<float4*>(g + i) = f4zero();
        else if (h.max_norm > 0.f) *reinterpret_cast<float4*>(g + i) = gv;
        if (s1) *reinterpret_cast<float4*>(s1 + i) = a;
        if (s2) *reinterpret_cast<float4*>(s2 + i) = b;
      } else {
        for (int e = 0; e < m; ++e) {
          float pv = p[i + e], gv = g[i + e], a = s1 ? s1[i + e] : 0.f, b = s2 ? s2[i + e] : 0.f;
          update1<KIND>(pv, gv, a, b, h, coef, bc1, bc2s, first);
          p[i + e] = pv;
          if (h.zero_grads) g[i + e] = 0.f;
          else if (h.max_norm > 0.f) g[i + e] = gv;
          if (s1) s1[i + e] = a;
          if (s2) s2[i + e] = b;
        }
      }
    }
  }
}

// ktup_optim_clip_step: norm + clip + update in ONE launch.  Two launches cost a second pass over the gradients plus the gap
// between two dependent kernels -- at B = 512 that pair was 25 us of a 49 us step whose tables (9.7 MB) fit in registers chip-wide.
//   phase 1  a thread loads its CS_UPT float4 of gradient (and, while those are in flight, of parameter and state) and the
//            workgroup's sum of squares joins a hierarchical fp64 sum: workgroup b belongs to slot b % GN_SLOTS, adds to the
//            slot's accumulator and takes a ticket of the SLOT (one atomic per workgroup on ONE address serialises at ~30 ns each;
//            32 short chains on separate 64-byte lines do not); the workgroup completing a slot takes a global ticket, and
//            the one completing the last slot folds the slot sums, clears the scratch for the next launch and publishes the
//            total next to each slot's epoch word, which it then increments;
//   barrier  thread 0 of every workgroup read its slot's epoch BEFORE arriving (the epoch cannot move until every workgroup of
//            this launch has arrived) and polls it with agent-scope loads until it changes;
//   phase 2  the update runs on the registers of phase 1: no second read of anything.
// All workgroups must be resident at once: the grid is capped at CS_MAX_WG = 512 workgroups of 256 threads (two per CU -- a
// second process sharing the GPU, as in the 2-rank tests, still fits).  Tables too large for CS_UPT registers per thread take
// the same kernel with phase 2 re-reading (resident == false).  The poll is bounded: on a timeout ws[GN_ERR] is set and the
// update proceeds unclipped -- wrong, but never a hung GPU.
// `slots` (optional): the step's loss terms, accumulated by the fused step kernel that ran before this launch.  Thread 0 of
// workgroup 0 publishes  *loss_out = loss_scale * sum(slots)  (adding it to *loss_acc, a running sum, if given) and zeroes the
// slots for the next step.
constexpr int GN_SLOTS = 32, GN_STRIDE = 8;                       // doubles / u64 words; one 64-byte line per slot
constexpr int GN_SUM = 1, GN_TICK = GN_SUM + GN_SLOTS * GN_STRIDE, GN_GLOBAL = GN_TICK + GN_SLOTS * GN_STRIDE;
constexpr int GN_EPOCH = GN_GLOBAL + GN_STRIDE;                   // [slot]: {epoch, total} on one line
constexpr int GN_ERR = KTUP_OPTIM_WS_DOUBLES - 1;
static_assert(GN_EPOCH + GN_SLOTS * GN_STRIDE <= GN_ERR, "workspace");
constexpr int CS_UPT = 5, CS_MAX_WG = 512, CS_SPIN_LIMIT = 1 << 21;

template <int KIND>
__global__ __launch_bounds__(256, 4) void clip_step_kernel(OptTensors T, Hyper h, double* __restrict__ ws, double* __restrict__ gn, float* __restrict__ slots,
                                                        int n_slots, float loss_scale, float* __restrict__ loss_out,
                                                        float* __restrict__ loss_acc, const int64_t* __restrict__ steps_dev) {
  __shared__ float dev_bc1[MAXT], dev_bc2s[MAXT];
  __shared__ float red[4];
  __shared__ float coef_s;
  __shared__ int last_wg;
  unsigned long long* words = reinterpret_cast<unsigned long long*>(ws);
  const int slot = blockIdx.x % GN_SLOTS;
  const bool clip = h.max_norm > 0.f;
  unsigned long long e0 = 0;
  if (threadIdx.x == 0) {
    if (clip && !gn) e0 = __hip_atomic_load(words + GN_EPOCH + slot * GN_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (slots && blockIdx.x == 0) {
      float s = 0.f;
      for (int i = 0; i < n_slots; ++i) { s += slots[i]; slots[i] = 0.f; }
      *loss_out = loss_scale * s;
      if (loss_acc) *loss_acc += loss_scale * s;
    }
  }
  const bool dev_bc = KIND == KTUP_OPT_ADAM && steps_dev != nullptr;
  if (dev_bc && (int)threadIdx.x < T.count) {
    const double t = (double)steps_dev[threadIdx.x];
    dev_bc1[threadIdx.x] = (float)(1.0 - pow((double)h.beta1, t));
    dev_bc2s[threadIdx.x] = (float)sqrt(1.0 - pow((double)h.beta2, t));
  }
  const int64_t nunits = T.chunk0[T.count] * 4;                   // unit = a quarter chunk: 256 float4
  const int64_t stride = (int64_t)gridDim.x * CS_UPT;
  const bool resident = nunits <= stride;                         // everything this thread touches stays in its registers
  int kk[CS_UPT], mm[CS_UPT];
  uint32_t ii[CS_UPT];                                            // element offsets (tensors of 2^32 floats and more: see the host side)
  float4 gv[CS_UPT], pv[CS_UPT], av[CS_UPT];
  auto locate = [&](int64_t unit, int q) {
    kk[q] = 0; ii[q] = 0; mm[q] = 0;
    if (unit < nunits) {
      const int64_t chunk = unit >> 2;
      const int k = find_tensor(T, chunk);
      const int64_t i = (chunk - T.chunk0[k]) * CHUNK + ((unit & 3) * 256 + threadIdx.x) * 4, n = T.n[k];
      kk[q] = k; ii[q] = (uint32_t)i;
      mm[q] = i + 3 < n ? 4 : (i < n ? (int)(n - i) : 0);
    }
  };
  float coef = 1.f;
  if (clip && gn) {
    // the squared norm came with the gradients (ktup_common.h gnorm_*: the kernels that built them tracked it): every workgroup sums
    // the 16 slots of the set in use for itself -- no norm pass, no barrier; workgroup 0 also clears the idle set and counts the step
    if (threadIdx.x < 64) {
      unsigned long long* gw = reinterpret_cast<unsigned long long*>(gn);
      const int set = (int)(gw[1] & 1ull);
      double v = threadIdx.x < ktup::GNORM_SLOTS ? gn[ktup::GNORM_SET0 + ktup::GNORM_SLOTS * set + threadIdx.x] : 0.0;
#pragma unroll
      for (int m = ktup::GNORM_SLOTS / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      v = v > 0.0 ? v : 0.0;                                       // (rounding of the tracked terms can leave -1e-12 for a zero gradient)
      if (threadIdx.x == 0) {
        float c = h.max_norm / ((float)sqrt(v) + 1e-6f);
        coef_s = c < 1.f ? c : 1.f;
        if (blockIdx.x == 0) { ws[0] = v; gn[2] = v; gw[0] += 1ull; }
      }
      if (blockIdx.x == 0 && threadIdx.x < ktup::GNORM_SLOTS) gn[ktup::GNORM_SET0 + ktup::GNORM_SLOTS * (1 - set) + threadIdx.x] = 0.0;
    }
    __syncthreads();
    coef = coef_s;
  } else if (clip) {
    float acc = 0.f;
    for (int64_t base = blockIdx.x; base < nunits; base += stride) {
#pragma unroll
      for (int q = 0; q < CS_UPT; ++q) {
        locate(base + (int64_t)q * gridDim.x, q);
        gv[q] = mm[q] == 4 ? *reinterpret_cast<const float4*>(T.g[kk[q]] + ii[q]) : f4zero();
      }
      if (resident) {
#pragma unroll
        for (int q = 0; q < CS_UPT; ++q) {
          pv[q] = mm[q] == 4 ? *reinterpret_cast<const float4*>(T.p[kk[q]] + ii[q]) : f4zero();
          av[q] = (mm[q] == 4 && T.s1[kk[q]]) ? *reinterpret_cast<const float4*>(T.s1[kk[q]] + ii[q]) : f4zero();
        }
      }
#pragma unroll
      for (int q = 0; q < CS_UPT; ++q) {
        acc += dot4(gv[q], gv[q]);
        if (mm[q] > 0 && mm[q] < 4)
          for (int e = 0; e < mm[q]; ++e) acc = fmaf(T.g[kk[q]][ii[q] + e], T.g[kk[q]][ii[q] + e], acc);
      }
    }
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double blk = ((double)red[0] + (double)red[1]) + ((double)red[2] + (double)red[3]);
      const unsigned in_slot = (gridDim.x - slot + GN_SLOTS - 1) / GN_SLOTS;          // workgroups b with b % GN_SLOTS == slot
      const unsigned used = gridDim.x < GN_SLOTS ? gridDim.x : GN_SLOTS;             // slots that have a workgroup at all
      int last = 0;
      // no __threadfence() anywhere in this protocol: at agent scope it writes back / invalidates the XCD's whole L2 (the first
      // version spent 50 us in fences).  Everything the workgroups exchange travels in agent-scope atomics, which are performed
      // at the memory side whatever the XCD; order between two of them = wait for the first one's RETURN value.
      const double prev = atomicAdd(ws + GN_SUM + slot * GN_STRIDE, blk);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(prev) : "memory");
      if (atomicAdd(words + GN_TICK + slot * GN_STRIDE, 1ull) == in_slot - 1) {      // this slot is complete
        if (atomicAdd(words + GN_GLOBAL, 1ull) == used - 1) last = 1;                // ... and so is every other slot
      }
      last_wg = last;
    }
    __syncthreads();
    if (last_wg && threadIdx.x < 64) {         // fold the slots: one lane per slot, all exchanges in flight at once
      double v = 0.0;
      unsigned long long t0 = 0;
      if (threadIdx.x < GN_SLOTS) {
        v = __longlong_as_double((long long)atomicExch(words + GN_SUM + threadIdx.x * GN_STRIDE, 0ull));
        t0 = atomicExch(words + GN_TICK + threadIdx.x * GN_STRIDE, 0ull);
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      unsigned long long g0 = 0;
      if (threadIdx.x == 0) {
        ws[0] = v;                             // for the host (total_norm) and later launches
        g0 = atomicExch(words + GN_GLOBAL, 0ull);
      }
      if (threadIdx.x < GN_SLOTS) {            // the total, then (once that exchange has returned) the epoch that releases the waiters
        const unsigned long long was =
            atomicExch(words + GN_EPOCH + threadIdx.x * GN_STRIDE + 1, (unsigned long long)__double_as_longlong(v));
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(was), "v"(t0), "v"(g0) : "memory");
        atomicAdd(words + GN_EPOCH + threadIdx.x * GN_STRIDE, 1ull);
      }
    }
    if (threadIdx.x == 0) {
      int spins = 0;
      bool ok = true;
      while (__hip_atomic_load(words + GN_EPOCH + slot * GN_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > CS_SPIN_LIMIT) { ok = false; break; }
      }
      float c = 1.f;
      if (ok) {
        const double total = __longlong_as_double((long long)__hip_atomic_load(words + GN_EPOCH + slot * GN_STRIDE + 1, __ATOMIC_RELAXED,
                                                                               __HIP_MEMORY_SCOPE_AGENT));
        c = h.max_norm / ((float)sqrt(total) + 1e-6f);
        c = c < 1.f ? c : 1.f;
      } else {
        atomicExch(words + GN_ERR, 1ull);
      }
      coef_s = c;
    }
    __syncthreads();
    coef = coef_s;
  } else {
    __syncthreads();                                              // dev_bc tables
  }
  const bool in_regs = clip && resident && !gn;
  for (int64_t base = blockIdx.x; base < nunits; base += stride) {
#pragma unroll
    for (int q = 0; q < CS_UPT; ++q) {
      if (!in_regs) {
        locate(base + (int64_t)q * gridDim.x, q);
        if (mm[q] == 4) {
          gv[q] = *reinterpret_cast<const float4*>(T.g[kk[q]] + ii[q]);
          pv[q] = *reinterpret_cast<const float4*>(T.p[kk[q]] + ii[q]);
          av[q] = T.s1[kk[q]] ? *reinterpret_cast<const float4*>(T.s1[kk[q]] + ii[q]) : f4zero();
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CS_UPT; ++q) {
      if (mm[q] == 0) continue;
      const int k = kk[q];
      const size_t i = ii[q];
      float *p = T.p[k], *g = T.g[k], *s1 = T.s1[k], *s2 = T.s2[k];
      const float bc1 = dev_bc ? dev_bc1[k] : T.bc1[k], bc2s = dev_bc ? dev_bc2s[k] : T.bc2s[k];
      const bool first = T.first[k] != 0;
      if (mm[q] == 4) {
        float4 b = s2 ? *reinterpret_cast<float4*>(s2 + i) : f4zero();
        update1<KIND>(pv[q].x, gv[q].x, av[q].x, b.x, h, coef, bc1, bc2s, first);
        update1<KIND>(pv[q].y, gv[q].y, av[q].y, b.y, h, coef, bc1, bc2s, first);
        update1<KIND>(pv[q].z, gv[q].z, av[q].z, b.z, h, coef, bc1, bc2s, first);
        update1<KIND>(pv[q].w, gv[q].w, av[q].w, b.w, h, coef, bc1, bc2s, first);
        *reinterpret_cast<float4*>(p + i) = pv[q];
        if (h.zero_grads) *reinterpret_cast<float4*>(g + i) = f4zero();
        else if (clip) *reinterpret_cast<float4*>(g + i) = gv[q];
        if (s1) *reinterpret_cast<float4*>(s1 + i) = av[q];
        if (s2) *reinterpret_cast<float4*>(s2 + i) = b;
      } else {
        for (int e = 0; e < mm[q]; ++e) {
          float pe = p[i + e], ge = g[i + e], a = s1 ? s1[i + e] : 0.f, b = s2 ? s2[i + e] : 0.f;
          update1<KIND>(pe, ge, a, b, h, coef, bc1, bc2s, first);
          p[i + e] = pe;
          if (h.zero_grads) g[i + e] = 0.f;
          else if (clip) g[i + e] = ge;
          if (s1) s1[i + e] = a;
          if (s2) s2[i + e] = b;
        }
      }
    }
  }
}

int fill(const char* name, OptTensors& T, int n_tensors, float* const* params, float* const* grads, float* const* state1,
         float* const* state2, const int64_t* sizes) {
  KTUP_REQUIRE(n_tensors >= 0 && n_tensors <= MAXT, "%s: %d tensors (max %d per call)", name, n_tensors, MAXT);
  T.count = n_tensors;
  int64_t c = 0;
  for (int i = 0; i < n_tensors; ++i) {
    KTUP_REQUIRE(grads[i] && sizes[i] >= 0, "%s: tensor %d: null gradient or negative size", name, i);
    KTUP_REQUIRE(aligned16(grads[i]) && (!params || aligned16(params[i])) && (!state1 || !state1[i] || aligned16(state1[i])) &&
                     (!state2 || !state2[i] || aligned16(state2[i])),
                 "%s: tensor %d: arrays must be 16-byte aligned", name, i);
    T.p[i] = params ? params[i] : nullptr;
    T.g[i] = grads[i];
    T.s1[i] = state1 ? state1[i] : nullptr;
    T.s2[i] = state2 ? state2[i] : nullptr;
    T.n[i] = sizes[i];
    T.chunk0[i] = c;
    c += (sizes[i] + CHUNK - 1) / CHUNK;
    T.bc1[i] = 1.f; T.bc2s[i] = 1.f; T.first[i] = 0;
  }
  for (int i = n_tensors; i <= MAXT; ++i) T.chunk0[i] = c;
  T.chunk0[n_tensors] = c;
  return KTUP_OK;
}

}  // namespace

extern "C" int ktup_optim_gradnorm(int n_tensors, float* const* grads, const int64_t* sizes, double* sumsq, void* stream) {
  OptTensors T{};
  KTUP_REQUIRE(grads && sizes && sumsq, "ktup_optim_gradnorm: null pointer argument");
  if (int e = fill("ktup_optim_gradnorm", T, n_tensors, nullptr, grads, nullptr, nullptr, sizes)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sumsq, 0, sizeof(double), st) != hipSuccess) return check_launch("ktup_optim_gradnorm");
  const int64_t nchunks = T.chunk0[T.count];
  if (nchunks == 0) return KTUP_OK;
  hipLaunchKernelGGL(gradnorm_kernel, dim3(grid_for((nchunks * 4 + 3) / 4, 256)), dim3(256), 0, st, T, sumsq, 1);
  return check_launch("ktup_optim_gradnorm");
}

// The same sum ADDED to sumsq[0 .. n_slots) (workgroup b adds to slot b mod n_slots; the sum of the slots is the result).  No
// clearing memset: for callers inside a HIP graph that keep their accumulators clean themselves.
extern "C" int ktup_optim_gradnorm_acc(int n_tensors, float* const* grads, const int64_t* sizes, double* sumsq, int n_slots, void* stream) {
  OptTensors T{};
  KTUP_REQUIRE(grads && sizes && sumsq && n_slots >= 1, "ktup_optim_gradnorm_acc: null pointer argument or no slot");
  if (int e = fill("ktup_optim_gradnorm_acc", T, n_tensors, nullptr, grads, nullptr, nullptr, sizes)) return e;
  const int64_t nchunks = T.chunk0[T.count];
  if (nchunks == 0) return KTUP_OK;
  hipLaunchKernelGGL(gradnorm_kernel, dim3(grid_for((nchunks * 4 + 3) / 4, 1024)), dim3(256), 0, (hipStream_t)stream, T, sumsq, n_slots);
  return check_launch("ktup_optim_gradnorm_acc");
}

namespace {

int prep_step(const char* name, OptTensors& T, int kind, int n_tensors, float* const* params, float* const* grads, float* const* state1,
              float* const* state2, const int64_t* sizes, const int64_t* steps, const int64_t* steps_dev, const int32_t* first,
              float momentum, float beta1, float beta2) {
  KTUP_REQUIRE(kind >= KTUP_OPT_SGD && kind <= KTUP_OPT_RMSPROP, "%s: unknown optimizer kind %d", name, kind);
  KTUP_REQUIRE(params && grads && sizes, "%s: null pointer argument", name);
  if (int e = fill(name, T, n_tensors, params, grads, state1, state2, sizes)) return e;
  for (int i = 0; i < n_tensors; ++i) {
    KTUP_REQUIRE(params[i], "%s: tensor %d: null parameter", name, i);
    const bool need1 = kind == KTUP_OPT_ADAGRAD || kind == KTUP_OPT_ADAM || kind == KTUP_OPT_RMSPROP ||
                       (kind == KTUP_OPT_SGD && momentum != 0.f);
    const bool need2 = kind == KTUP_OPT_ADAM || (kind == KTUP_OPT_RMSPROP && momentum > 0.f);
    KTUP_REQUIRE((!need1 || T.s1[i]) && (!need2 || T.s2[i]), "%s: tensor %d: optimizer state missing", name, i);
    if (kind == KTUP_OPT_ADAM && !steps_dev) {
      KTUP_REQUIRE(steps && steps[i] >= 1, "%s: Adam needs the (already incremented) step count of tensor %d", name, i);
      const double t = (double)steps[i];
      T.bc1[i] = (float)(1.0 - pow((double)beta1, t));
      T.bc2s[i] = (float)sqrt(1.0 - pow((double)beta2, t));
    }
    T.first[i] = first ? first[i] : 0;
  }
  return KTUP_OK;
}

}  // namespace

extern "C" int ktup_optim_step(int kind, int n_tensors, float* const* params, float* const* grads, float* const* state1,
                               float* const* state2, const int64_t* sizes, const int64_t* steps, const int64_t* steps_dev,
                               const int32_t* first, float lr,
                               float weight_decay, float momentum, float beta1, float beta2, float eps, float alpha,
                               const double* sumsq, float max_norm, int zero_grads, void* stream) {
  KTUP_REQUIRE(max_norm <= 0.f || sumsq, "ktup_optim_step: clipping needs the gradnorm result");
  OptTensors T{};
  if (int e = prep_step("ktup_optim_step", T, kind, n_tensors, params, grads, state1, state2, sizes, steps, steps_dev, first, momentum,
                        beta1, beta2))
    return e;
  const int64_t nchunks = T.chunk0[T.count];
  if (nchunks == 0) return KTUP_OK;
  const Hyper h{lr, weight_decay, momentum, beta1, beta2, eps, alpha, max_norm, zero_grads};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for(nchunks, 256 * 8)), block(256);
  switch (kind) {
    case KTUP_OPT_SGD: hipLaunchKernelGGL(step_kernel<KTUP_OPT_SGD>, grid, block, 0, st, T, h, sumsq, steps_dev); break;
    case KTUP_OPT_ADAGRAD: hipLaunchKernelGGL(step_kernel<KTUP_OPT_ADAGRAD>, grid, block, 0, st, T, h, sumsq, steps_dev); break;
    case KTUP_OPT_ADAM: hipLaunchKernelGGL(step_kernel<KTUP_OPT_ADAM>, grid, block, 0, st, T, h, sumsq, steps_dev); break;
    default: hipLaunchKernelGGL(step_kernel<KTUP_OPT_RMSPROP>, grid, block, 0, st, T, h, sumsq, steps_dev); break;
  }
  return check_launch("ktup_optim_step");
}

namespace {

// How many clip_step workgroups can be RESIDENT at once on the current device: the hand-rolled grid barrier needs them all.
// occupancy query x compute units, one workgroup per CU short of it (MI355X_MICROARCH.md: the API answer can be one block per
// CU too high when the kernel's SGPR count is 81-112, and a surplus block would strand the barrier), never more than
// CS_MAX_WG.  Cached per (device, optimizer kind).  0 = the query failed: the caller keeps the two-launch route.
template <int KIND>
int cs_capacity() {
  static int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (dev == cached_dev) return cached;
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)clip_step_kernel<KIND>, 256, 0) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (per_cu > 8) per_cu = 8;
  if (per_cu > 1) per_cu -= 1;
  int cap = per_cu * cus;
  cached_dev = dev;
  cached = cap < CS_MAX_WG ? cap : CS_MAX_WG;
  return cached;
}

int cs_capacity_of(int kind) {
  switch (kind) {
    case KTUP_OPT_SGD: return cs_capacity<KTUP_OPT_SGD>();
    case KTUP_OPT_ADAGRAD: return cs_capacity<KTUP_OPT_ADAGRAD>();
    case KTUP_OPT_ADAM: return cs_capacity<KTUP_OPT_ADAM>();
    default: return cs_capacity<KTUP_OPT_RMSPROP>();
  }
}

}  // namespace

// The largest grid ktup_optim_clip_step will launch on the current device (all of it resident at once), 0 if unknown.
extern "C" int ktup_optim_clip_step_capacity(int kind) {
  if (kind < KTUP_OPT_SGD || kind > KTUP_OPT_RMSPROP) return 0;
  return cs_capacity_of(kind);
}

// ktup_optim_gradnorm + ktup_optim_step as one launch (see clip_step_kernel); `ws`: KTUP_OPTIM_WS_DOUBLES doubles, zero-filled
// ONCE by the caller and left consistent by every launch ([0] = the squared norm of the last clipped step).
extern "C" int ktup_optim_clip_step(int kind, int n_tensors, float* const* params, float* const* grads, float* const* state1,
                                    float* const* state2, const int64_t* sizes, const int64_t* steps, const int64_t* steps_dev,
                                    const int32_t* first, float lr, float weight_decay, float momentum, float beta1, float beta2,
                                    float eps, float alpha, double* ws, double* gnorm, float max_norm, int zero_grads, float* loss_slots, int n_slots,
                                    float loss_scale, float* loss_out, float* loss_acc, void* stream) {
  const char* name = "ktup_optim_clip_step";
  KTUP_REQUIRE(ws, "%s: null workspace", name);
  KTUP_REQUIRE(!loss_slots || (n_slots > 0 && loss_out), "%s: loss slots need a count and an output", name);
  KTUP_REQUIRE(!gnorm || max_norm > 0.f, "%s: a tracked gradient norm without clipping", name);
  OptTensors T{};
  if (int e = prep_step(name, T, kind, n_tensors, params, grads, state1, state2, sizes, steps, steps_dev, first, momentum, beta1, beta2))
    return e;
  for (int i = 0; i < n_tensors; ++i) KTUP_REQUIRE(sizes[i] < (1ll << 32), "%s: tensor %d: 2^32 elements or more", name, i);
  const int64_t nunits = T.chunk0[T.count] * 4;
  if (nunits == 0 && !loss_slots) return KTUP_OK;
  const Hyper h{lr, weight_decay, momentum, beta1, beta2, eps, alpha, max_norm, zero_grads};
  hipStream_t st = (hipStream_t)stream;
  const int cap = cs_capacity_of(kind);
  if (cap <= 0)
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: cannot size a resident grid on this device (occupancy query failed): use ktup_optim_gradnorm + ktup_optim_step", name);
  const dim3 grid((unsigned)(nunits < 1 ? 1 : nunits < cap ? nunits : cap)), block(256);
  switch (kind) {
    case KTUP_OPT_SGD:
      hipLaunchKernelGGL(clip_step_kernel<KTUP_OPT_SGD>, grid, block, 0, st, T, h, ws, gnorm, loss_slots, n_slots, loss_scale, loss_out, loss_acc, steps_dev);
      break;
    case KTUP_OPT_ADAGRAD:
      hipLaunchKernelGGL(clip_step_kernel<KTUP_OPT_ADAGRAD>, grid, block, 0, st, T, h, ws, gnorm, loss_slots, n_slots, loss_scale, loss_out, loss_acc, steps_dev);
      break;
    case KTUP_OPT_ADAM:
      hipLaunchKernelGGL(clip_step_kernel<KTUP_OPT_ADAM>, grid, block, 0, st, T, h, ws, gnorm, loss_slots, n_slots, loss_scale, loss_out, loss_acc, steps_dev);
      break;
    default:
      hipLaunchKernelGGL(clip_step_kernel<KTUP_OPT_RMSPROP>, grid, block, 0, st, T, h, ws, gnorm, loss_slots, n_slots, loss_scale, loss_out, loss_acc, steps_dev);
      break;
  }
  return check_launch(name);
}
