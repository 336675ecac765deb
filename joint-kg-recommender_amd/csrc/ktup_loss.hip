// K8-K10: BPR / margin losses and the norm / orthogonality regularisers, forward and backward.
//   reference: jTransUP/utils/loss.py:8-31
// The reference builds each of these from 4-6 elementwise torch kernels and (for the regularisers) re-gathers
// the batch's embedding rows a second time; here the gather is fused (ids may be NULL = every row of the table)
// and every loss is one launch: per-thread partials -> block reduction -> one atomic per workgroup.
// `gloss` (upstream gradient of the scalar) is a DEVICE scalar so that no host sync is needed.
#include "ktup_rows.h"

using namespace ktup;

namespace {

// -log(sigmoid(x)) = softplus(-x), evaluated the way torch's logsigmoid does: max(-x,0) + log1p(exp(-|x|))
KTUP_DEV float neg_logsigmoid(float x) { return fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))); }
KTUP_DEV float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

template <bool MARGIN>
__global__ __launch_bounds__(256) void pair_loss_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                            int64_t n, float param, float scale, float* __restrict__ loss) {
  float part = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float diff = pos[i] - neg[i];
    part += MARGIN ? fmaxf(diff + param, 0.f) : neg_logsigmoid(param * diff);
  }
  part = block_sum_256(part);
  if (threadIdx.x == 0) atomicAdd(loss, part * scale);
}

template <bool MARGIN>
__global__ __launch_bounds__(256) void pair_loss_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                            int64_t n, float param, float scale, const float* __restrict__ gloss,
                                                            float* __restrict__ gpos, float* __restrict__ gneg) {
  const float g = gloss[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float diff = pos[i] - neg[i];
    // d softplus(-x)/dx = -sigmoid(-x), x = target*diff ;  d max(diff+m,0)/ddiff = [diff+m > 0]
    const float gd = MARGIN ? (diff + param > 0.f ? g : 0.f) : -g * param * sigmoidf(-param * diff);
    gpos[i] = gd;
    gneg[i] = -gd;
  }
}

// value + gradient in one pass (the GPU-resident training step): the loss is ADDED to *loss_acc, which the caller zeroes once
// per step for all its loss terms
template <bool MARGIN>
__global__ __launch_bounds__(256) void pair_loss_fused_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                              int64_t n, float param, float scale, const float* __restrict__ gloss,
                                                              float* __restrict__ loss_acc, float* __restrict__ gpos,
                                                              float* __restrict__ gneg) {
  const float g = gloss[0] * scale;
  float part = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float diff = pos[i] - neg[i];
    part += MARGIN ? fmaxf(diff + param, 0.f) : neg_logsigmoid(param * diff);
    const float gd = MARGIN ? (diff + param > 0.f ? g : 0.f) : -g * param * sigmoidf(-param * diff);
    gpos[i] = gd;
    gneg[i] = -gd;
  }
  part = block_sum_256(part);
  if (threadIdx.x == 0) atomicAdd(loss_acc, part * scale);
}

struct NormFwd {  // loss.py:21-23  sum_rows max(|x|^2 - 1, 0)
  const float* T; int64_t ld; const int64_t* ids;
  template <typename V, int G, int CPL>
  KTUP_DEV float run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    V x[CPL];
    cx.load(x, T + (ids ? ids[row] : row) * ld);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s += vdot(x[j], x[j]);
    return fmaxf(group_sum<G>(s) - 1.f, 0.f);
  }
};
struct NormBwd {
  const float* T; int64_t ld; const int64_t* ids; const float* gloss; float* gT;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t r = ids ? ids[row] : row;
    V x[CPL];
    cx.load(x, T + r * ld);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s += vdot(x[j], x[j]);
    s = group_sum<G>(s);
    if (s - 1.f > 0.f) {
      const float g = 2.f * gloss[0];
#pragma unroll
      for (int j = 0; j < CPL; ++j) x[j] = vscale(g, x[j]);
      cx.scatter_add(gT + r * ld, x);
    }
  }
};

struct OrthFwd {  // loss.py:18-19  sum_rows (w.r)^2 / |r|^2
  const float *R, *W; int64_t ldr, ldw; const int64_t* ids;
  template <typename V, int G, int CPL>
  KTUP_DEV float run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t r = ids ? ids[row] : row;
    V a[CPL], w[CPL];
    cx.load(a, R + r * ldr);
    cx.load(w, W + r * ldw);
    float dot = 0.f, nr = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { dot += vdot(a[j], w[j]); nr += vdot(a[j], a[j]); }
    dot = group_sum<G>(dot);
    nr = group_sum<G>(nr);
    return dot * dot / nr;
  }
};
struct OrthBwd {  // d/dw = (2a/b) r ;  d/dr = (2a/b) w - (2a^2/b^2) r   with a = w.r, b = r.r
  const float *R, *W; int64_t ldr, ldw; const int64_t* ids; const float* gloss; float *gR, *gW;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t r = ids ? ids[row] : row;
    V a[CPL], w[CPL];
    cx.load(a, R + r * ldr);
    cx.load(w, W + r * ldw);
    float dot = 0.f, nr = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { dot += vdot(a[j], w[j]); nr += vdot(a[j], a[j]); }
    dot = group_sum<G>(dot);
    nr = group_sum<G>(nr);
    const float g = gloss[0], c1 = g * 2.f * dot / nr, c2 = g * 2.f * dot * dot / (nr * nr);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const V gw = vscale(c1, a[j]);
      a[j] = vfma(-c2, a[j], vscale(c1, w[j]));
      w[j] = gw;
    }
    cx.scatter_add(gR + r * ldr, a);
    cx.scatter_add(gW + r * ldw, w);
  }
};

struct NormFused {   // NormFwd value + NormBwd scatter
  const float* T; int64_t ld; const int64_t* ids; const float* gloss; float* gT;
  template <typename V, int G, int CPL>
  KTUP_DEV float run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t r = ids ? ids[row] : row;
    V x[CPL];
    cx.load(x, T + r * ld);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s += vdot(x[j], x[j]);
    s = group_sum<G>(s);
    if (s - 1.f > 0.f) {
      const float g = 2.f * gloss[0];
#pragma unroll
      for (int j = 0; j < CPL; ++j) x[j] = vscale(g, x[j]);
      cx.scatter_add(gT + r * ld, x);
    }
    return fmaxf(s - 1.f, 0.f);
  }
};

struct OrthFused {   // OrthFwd value + OrthBwd scatter
  const float *R, *W; int64_t ldr, ldw; const int64_t* ids; const float* gloss; float *gR, *gW;
  template <typename V, int G, int CPL>
  KTUP_DEV float run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    const int64_t r = ids ? ids[row] : row;
    V a[CPL], w[CPL];
    cx.load(a, R + r * ldr);
    cx.load(w, W + r * ldw);
    float dot = 0.f, nr = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { dot += vdot(a[j], w[j]); nr += vdot(a[j], a[j]); }
    dot = group_sum<G>(dot);
    nr = group_sum<G>(nr);
    const float g = gloss[0], c1 = g * 2.f * dot / nr, c2 = g * 2.f * dot * dot / (nr * nr);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const V gw = vscale(c1, a[j]);
      a[j] = vfma(-c2, a[j], vscale(c1, w[j]));
      w[j] = gw;
    }
    cx.scatter_add(gR + r * ldr, a);
    cx.scatter_add(gW + r * ldw, w);
    return dot * dot / nr;
  }
};

template <bool MARGIN>
int pair_loss(bool bwd, const char* name, const float* pos, const float* neg, int64_t n, float param, bool mean,
              float* loss, const float* gloss, float* gpos, float* gneg, void* stream) {
  KTUP_REQUIRE(n >= 0, "%s: negative length", name);
  hipStream_t st = (hipStream_t)stream;
  const float scale = mean ? (n > 0 ? 1.f / (float)n : 0.f) : 1.f;
  if (!bwd) {
    KTUP_REQUIRE(loss, "%s: null loss pointer", name);
    if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) return check_launch(name);
    if (n == 0) return KTUP_OK;   // (torch's mean of an empty tensor is nan; an empty batch never reaches here)
    KTUP_REQUIRE(pos && neg, "%s: null score pointer", name);
    hipLaunchKernelGGL(pair_loss_fwd_kernel<MARGIN>, dim3(grid_for((n + 1023) / 1024, 64)), dim3(256), 0, st, pos, neg, n,
                       param, scale, loss);
  } else {
    if (n == 0) return KTUP_OK;
    KTUP_REQUIRE(pos && neg && gloss && gpos && gneg, "%s: null pointer argument", name);
    hipLaunchKernelGGL(pair_loss_bwd_kernel<MARGIN>, dim3(grid_for((n + 255) / 256, 1024)), dim3(256), 0, st, pos, neg, n,
                       param, scale, gloss, gpos, gneg);
  }
  return check_launch(name);
}

}  // namespace

extern "C" int ktup_loss_bpr_fwd(const float* pos, const float* neg, int64_t n, float target, float* loss, void* stream) {
  return pair_loss<false>(false, "ktup_loss_bpr_fwd", pos, neg, n, target, true, loss, nullptr, nullptr, nullptr, stream);
}
extern "C" int ktup_loss_bpr_bwd(const float* pos, const float* neg, int64_t n, float target, const float* gloss, float* gpos,
                                 float* gneg, void* stream) {
  return pair_loss<false>(true, "ktup_loss_bpr_bwd", pos, neg, n, target, true, nullptr, gloss, gpos, gneg, stream);
}
extern "C" int ktup_loss_margin_fwd(const float* pos, const float* neg, int64_t n, float margin, float* loss, void* stream) {
  return pair_loss<true>(false, "ktup_loss_margin_fwd", pos, neg, n, margin, false, loss, nullptr, nullptr, nullptr, stream);
}
extern "C" int ktup_loss_margin_bwd(const float* pos, const float* neg, int64_t n, float margin, const float* gloss,
                                    float* gpos, float* gneg, void* stream) {
  return pair_loss<true>(true, "ktup_loss_margin_bwd", pos, neg, n, margin, false, nullptr, gloss, gpos, gneg, stream);
}

extern "C" int ktup_reg_norm_fwd(const float* T, int64_t ld, int d, const int64_t* ids, int64_t n, float* loss, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && loss && (T || n == 0), "ktup_reg_norm_fwd: bad argument");
  NormFwd op{T, ld, ids};
  return launch_rows_reduce(op, d, can_vec4(d, {T}, {ld}), n, loss, (hipStream_t)stream, "ktup_reg_norm_fwd");
}
extern "C" int ktup_reg_norm_bwd(const float* T, int64_t ld, int d, const int64_t* ids, int64_t n, const float* gloss,
                                 float* gT, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && ((T && gloss && gT) || n == 0), "ktup_reg_norm_bwd: bad argument");
  NormBwd op{T, ld, ids, gloss, gT};
  return launch_rows(op, d, can_vec4(d, {T, gT}, {ld}), n, (hipStream_t)stream, "ktup_reg_norm_bwd");
}
extern "C" int ktup_reg_orth_fwd(const float* Rel, int64_t ldr, const float* Nrm, int64_t ldn, int d, const int64_t* ids,
                                 int64_t n, float* loss, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && loss && ((Rel && Nrm) || n == 0), "ktup_reg_orth_fwd: bad argument");
  OrthFwd op{Rel, Nrm, ldr, ldn, ids};
  return launch_rows_reduce(op, d, can_vec4(d, {Rel, Nrm}, {ldr, ldn}), n, loss, (hipStream_t)stream, "ktup_reg_orth_fwd");
}
extern "C" int ktup_reg_orth_bwd(const float* Rel, int64_t ldr, const float* Nrm, int64_t ldn, int d, const int64_t* ids,
                                 int64_t n, const float* gloss, float* gRel, float* gNrm, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && ((Rel && Nrm && gloss && gRel && gNrm) || n == 0), "ktup_reg_orth_bwd: bad argument");
  OrthBwd op{Rel, Nrm, ldr, ldn, ids, gloss, gRel, gNrm};
  return launch_rows(op, d, can_vec4(d, {Rel, Nrm, gRel, gNrm}, {ldr, ldn}), n, (hipStream_t)stream, "ktup_reg_orth_bwd");
}

// ---- value + gradient in one launch each; `loss_acc` is accumulated into (zero it once per step)
template <bool MARGIN>
static int pair_loss_fused(const char* name, const float* pos, const float* neg, int64_t n, float param, bool mean,
                           const float* gloss, float* loss_acc, float* gpos, float* gneg, void* stream) {
  KTUP_REQUIRE(n >= 0, "%s: negative length", name);
  if (n == 0) return KTUP_OK;
  KTUP_REQUIRE(pos && neg && gloss && loss_acc && gpos && gneg, "%s: null pointer argument", name);
  const float scale = mean ? 1.f / (float)n : 1.f;
  hipLaunchKernelGGL(pair_loss_fused_kernel<MARGIN>, dim3(grid_for((n + 1023) / 1024, 64)), dim3(256), 0, (hipStream_t)stream, pos,
                     neg, n, param, scale, gloss, loss_acc, gpos, gneg);
  return check_launch(name);
}
extern "C" int ktup_loss_bpr_fused(const float* pos, const float* neg, int64_t n, float target, const float* gloss, float* loss_acc,
                                   float* gpos, float* gneg, void* stream) {
  return pair_loss_fused<false>("ktup_loss_bpr_fused", pos, neg, n, target, true, gloss, loss_acc, gpos, gneg, stream);
}
extern "C" int ktup_loss_margin_fused(const float* pos, const float* neg, int64_t n, float margin, const float* gloss,
                                      float* loss_acc, float* gpos, float* gneg, void* stream) {
  return pair_loss_fused<true>("ktup_loss_margin_fused", pos, neg, n, margin, false, gloss, loss_acc, gpos, gneg, stream);
}
extern "C" int ktup_reg_norm_fused(const float* T, int64_t ld, int d, const int64_t* ids, int64_t n, const float* gloss,
                                   float* loss_acc, float* gT, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && ((T && gloss && loss_acc && gT) || n == 0), "ktup_reg_norm_fused: bad argument");
  NormFused op{T, ld, ids, gloss, gT};
  return launch_rows_reduce(op, d, can_vec4(d, {T, gT}, {ld}), n, loss_acc, (hipStream_t)stream, "ktup_reg_norm_fused", false);
}
extern "C" int ktup_reg_orth_fused(const float* Rel, int64_t ldr, const float* Nrm, int64_t ldn, int d, const int64_t* ids,
                                   int64_t n, const float* gloss, float* loss_acc, float* gRel, float* gNrm, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && ((Rel && Nrm && gloss && loss_acc && gRel && gNrm) || n == 0), "ktup_reg_orth_fused: bad argument");
  OrthFused op{Rel, Nrm, ldr, ldn, ids, gloss, gRel, gNrm};
  return launch_rows_reduce(op, d, can_vec4(d, {Rel, Nrm, gRel, gNrm}, {ldr, ldn}), n, loss_acc, (hipStream_t)stream,
                            "ktup_reg_orth_fused", false);
}
