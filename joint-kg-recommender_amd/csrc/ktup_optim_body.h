// Shared pieces of K20 (ktup_optim.hip) for kernels elsewhere that carry the optimizer step as part of their launch
// (ktup_sample.hip: the optimizer step with the NEXT batch's feed riding in one extra workgroup).
#pragma once
#include "ktup_common.h"

namespace ktup {
namespace optb {

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

// one chunk (CHUNK floats of one tensor) by 256 threads; t256: the thread's index among them
template <int KIND>
KTUP_DEV void step_chunk(const OptTensors& T, const Hyper& h, float coef, int64_t chunk, int t256, bool dev_bc, const float* dev_bc1,
                         const float* dev_bc2s) {
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
      const int64_t i = base + ((int64_t)r * 256 + t256) * 4;
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

}  // namespace optb

// ktup_optim.hip: validate the caller's arrays and fill the by-value tensor table of a step launch (the checks of ktup_optim_step)
int optim_prepare(const char* name, optb::OptTensors& T, int kind, int n_tensors, float* const* params, float* const* grads,
                  float* const* state1, float* const* state2, const int64_t* sizes, const int64_t* steps, const int64_t* steps_dev,
                  const int32_t* first, float momentum, float beta1, float beta2);

}  // namespace ktup
