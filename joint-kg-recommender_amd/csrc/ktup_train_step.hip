// The B = 512 training step of the drivers as TWO launches: (1) one fused kernel per step kind that scores the positive and
// negative rows, forms the pairwise loss and every regulariser, and scatters all gradients; (2) the K20 global-norm clip + dense
// optimizer step in one launch (ktup_optim_clip_step, ktup_optim.hip).  Round 1 issued the same arithmetic as ~12 small launches per step (prepare, forward,
// loss, backward, regularisers, gradient fan-out, zero-fills): 0.085 ms per step with the device idle most of the time.
//
//   rec step (knowledgable_recommendation.py:335-344, item_recommendation.py:160-182):
//       bprLoss(pos, neg, target) + orthogonalLoss(pref, pref_norm)            -> STEP kernels of ktup_score_pref_bwd_wide.hip
//   kg step  (knowledgable_recommendation.py:345-382, knowledge_representation.py:176-204), TransH / TransE:
//       marginLoss(pos, neg, margin) + orthogonalLoss(rel, norm)[rel ids] + normLoss(ent)[h, t of pos and neg] + normLoss(rel)[rel ids]
//                                                                              -> kg_step_kernel below
// A lane group owns pair k = the positive triple k and its corrupted twin: both scores, the margin term, both backward passes
// and the regulariser terms of the rows it has in registers anyway (the reference re-gathers them through second nn.Embedding
// lookups); each gathered row receives ONE float4 atomic per chunk carrying the sum of all its contributions.
#include "ktup_rows.h"
#include "ktup_pref_geom.h"

using namespace ktup;

namespace {

struct KgStepArgs {
  const float *E, *R, *Nm; int64_t lde, ldr, ldn;
  const int64_t *h, *t, *r;      // [pos ; neg] rows: k and k + B
  int64_t B; int nch; bool l1;
  float margin, gscale;          // upstream gradient of every term of the step (kg_lambda, or 1)
  int regs;                      // bit 0 orthogonalLoss(rel, norm) rows, bit 1 normLoss(entity rows), bit 2 normLoss(relation rows)
  float* loss;                   // [4]: margin sum, orth, normE, normR  (accumulated)
  float *gE, *gR, *gN;
  double* gnorm;                 // (may be null) the gradient-norm workspace of ktup_common.h: every add tracks the squared norm it builds
  int serial;                    // option `deterministic`: ONE lane group of one workgroup walks every pair, so that each gradient cell receives its
                                 // adds in program order (the sums of float atomics from several waves depend on the order they land in)
};

template <int GL, bool TRANSH>
__global__ __launch_bounds__(256) void kg_step_kernel(KgStepArgs a) {
  constexpr int GPB = 256 / GL;
  const int lane = threadIdx.x % GL;
  const bool on = lane < a.nch;
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  const float g1 = a.gscale;
  const bool track = a.gnorm != nullptr;
  const int gset = track ? gnorm_set(a.gnorm) : 0;
  float ssq = 0.f;
  const int64_t k0 = a.serial ? (threadIdx.x < GL ? 0 : a.B) : (int64_t)blockIdx.x * GPB + threadIdx.x / GL;
  const int64_t kstep = a.serial ? 1 : (int64_t)gridDim.x * GPB;
  for (int64_t k = k0; k < a.B; k += kstep) {
    const int64_t id[4] = {a.h[k], a.t[k], a.h[k + a.B], a.t[k + a.B]};      // ph, pt, nh, nt
    const int64_t rid[2] = {a.r[k], a.r[k + a.B]};
    float4 e[4], rr[2], ww[2];
#pragma unroll
    for (int x = 0; x < 4; ++x) e[x] = on ? reinterpret_cast<const float4*>(a.E + id[x] * a.lde)[lane] : f4zero();
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      rr[x] = on ? reinterpret_cast<const float4*>(a.R + rid[x] * a.ldr)[lane] : f4zero();
      ww[x] = (TRANSH && on) ? reinterpret_cast<const float4*>(a.Nm + rid[x] * a.ldn)[lane] : f4zero();
    }
    // ---- forward of both triples (transH.py:58-71 / transE.py:51-63)
    float dh[2] = {0.f, 0.f}, dt[2] = {0.f, 0.f}, sc[2];
    float4 z[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const float4 hh = e[2 * x], tt = e[2 * x + 1];
      if (TRANSH) {
        dh[x] = group_sum<GL>(dot4(hh, ww[x]));
        dt[x] = group_sum<GL>(dot4(tt, ww[x]));
        const float4 ph = fma4(-dh[x], ww[x], hh), pt = fma4(-dt[x], ww[x], tt);
        z[x] = (ph + rr[x]) - pt;
      } else {
        z[x] = (hh + rr[x]) - tt;
      }
      sc[x] = group_sum<GL>(dist4(z[x], a.l1));
    }
    // ---- marginLoss (utils/loss.py:8-16): sum_k max(pos - neg + margin, 0)
    const float diff = sc[0] - sc[1];
    const bool act = diff + a.margin > 0.f;
    if (lane == 0) part[0] += fmaxf(diff + a.margin, 0.f);
    const float gs[2] = {act ? g1 : 0.f, act ? -g1 : 0.f};
    // ---- backward of both triples + the regularisers of the rows in registers
    float4 ge[4], gr[2], gw[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const float4 gz = gs[x] * ddist4(z[x], a.l1);
      if (TRANSH) {
        const float aw = group_sum<GL>(dot4(gz, ww[x]));
        const float sq = dh[x] - dt[x];
        const float4 q = e[2 * x] - e[2 * x + 1];
        ge[2 * x] = fma4(-aw, ww[x], gz);                          // gh
        ge[2 * x + 1] = -1.f * ge[2 * x];                          // gt
        gw[x] = fma4(-aw, q, (-sq) * gz);                          // gw
      } else {
        ge[2 * x] = gz;
        ge[2 * x + 1] = -1.f * gz;
        gw[x] = f4zero();
      }
      gr[x] = gz;
    }
    if (a.regs & 2) {                 // normLoss(ent rows): sum max(|x|^2 - 1, 0)  (utils/loss.py:21-23)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float s = group_sum<GL>(dot4(e[x], e[x]));
        if (s - 1.f > 0.f) ge[x] = fma4(2.f * g1, e[x], ge[x]);
        if (lane == 0) part[2] += fmaxf(s - 1.f, 0.f);
      }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      if (TRANSH && (a.regs & 1)) {   // orthogonalLoss(rel rows, norm rows): sum (w.r)^2 / |r|^2  (utils/loss.py:18-19)
        const float dot = group_sum<GL>(dot4(rr[x], ww[x])), nr = group_sum<GL>(dot4(rr[x], rr[x]));
        const float c1 = g1 * 2.f * dot / nr, c2 = g1 * 2.f * dot * dot / (nr * nr);
        gr[x] = gr[x] + fma4(-c2, rr[x], c1 * ww[x]);
        gw[x] = fma4(c1, rr[x], gw[x]);
        if (lane == 0) part[1] += dot * dot / nr;
      }
      if (a.regs & 4) {               // normLoss(rel rows)
        const float s = group_sum<GL>(dot4(rr[x], rr[x]));
        if (s - 1.f > 0.f) gr[x] = fma4(2.f * g1, rr[x], gr[x]);
        if (lane == 0) part[3] += fmaxf(s - 1.f, 0.f);
      }
    }
    if (on && track) {                // every add issued before the first returned value is used (ktup_common.h)
      float4 oe[4], orr[2], ow[2];
#pragma unroll
      for (int x = 0; x < 4; ++x) oe[x] = atomic_add4_old(a.gE + id[x] * a.lde + 4 * lane, ge[x]);
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        orr[x] = atomic_add4_old(a.gR + rid[x] * a.ldr + 4 * lane, gr[x]);
        ow[x] = TRANSH ? atomic_add4_old(a.gN + rid[x] * a.ldn + 4 * lane, gw[x]) : f4zero();
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) ssq += sq_gain4(oe[x], ge[x]);
#pragma unroll
      for (int x = 0; x < 2; ++x) ssq += sq_gain4(orr[x], gr[x]) + (TRANSH ? sq_gain4(ow[x], gw[x]) : 0.f);
    } else if (on) {
#pragma unroll
      for (int x = 0; x < 4; ++x) atomic_add4(a.gE + id[x] * a.lde + 4 * lane, ge[x]);
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        atomic_add4(a.gR + rid[x] * a.ldr + 4 * lane, gr[x]);
        if (TRANSH) atomic_add4(a.gN + rid[x] * a.ldn + 4 * lane, gw[x]);
      }
    }
  }
  // ---- the four loss values: wave sums -> one atomic per workgroup and slot
  __shared__ float red[4][5];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float v = group_sum<64>(part[s]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][s] = v;
  }
  if (track) {
    ssq = group_sum<64>(ssq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][4] = ssq;
  }
  __syncthreads();
  if (track && threadIdx.x == 64) gnorm_add(a.gnorm, gset, ((double)red[0][4] + (double)red[1][4]) + ((double)red[2][4] + (double)red[3][4]));
  if (threadIdx.x < 4) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (v != 0.f) atomicAdd(a.loss + threadIdx.x, v);
  }
}

template <bool TRANSH>
int launch_kg(const KgStepArgs& a, hipStream_t st, const char* name) {
#define KTUP_KG(GL)                                                                                            \
  {                                                                                                            \
    const int grid = a.serial ? 1 : grid_for((a.B + (256 / GL) - 1) / (256 / GL), 1024);                       \
    hipLaunchKernelGGL((kg_step_kernel<GL, TRANSH>), dim3(grid), dim3(256), 0, st, a);                         \
    return check_launch(name);                                                                                 \
  }
  if (a.nch <= 16) KTUP_KG(16)
  if (a.nch <= 32) KTUP_KG(32)
  KTUP_KG(64)
#undef KTUP_KG
}

// ---- the row regularisers of TUP's rec step (item_recommendation.py:177-180) for a step whose row gradients are STORED per pair
// (ktup_train_rec_step_rows): normLoss(user rows of the B examples) + normLoss(item rows of the 2B pairs) + normLoss(pref), with
// normLoss(x) = sum_rows max(|x|^2 - 1, 0) (utils/loss.py:21-23): a row with |x|^2 > 1 adds 2 x to its gradient.  Row k of GU is example
// k's user, row k of GV pair k's item, so the terms go to the stored rows by plain read-modify-write (a lane group per row), after the step
// kernel and before the reduction reads them.
struct RegRowsArgs {
  const float *U, *I; int64_t ldu, ldi; int nch;
  const int64_t *u_ids, *i_ids; int64_t B;
  float *GU, *GV; int64_t ldg;
  const float* pref; int n_pref; float* gP;
  float scale_rows, scale_pref;
  float* loss;                     // [2]: normLoss(user rows) + normLoss(item rows), normLoss(pref)  (accumulated)
};

template <int GL>
__global__ __launch_bounds__(256) void reg_rows_kernel(RegRowsArgs a) {
  constexpr int GPB = 256 / GL;
  const int lane = threadIdx.x % GL;
  const bool on = lane < a.nch;
  float part[2] = {0.f, 0.f};
  const int64_t total = 3 * a.B + a.n_pref;
  for (int64_t r = (int64_t)blockIdx.x * GPB + threadIdx.x / GL; r < total; r += (int64_t)gridDim.x * GPB) {
    const float* src;
    float* dst;
    float scale = a.scale_rows;
    int which = 0;
    if (r < a.B) { src = a.U + a.u_ids[r] * a.ldu; dst = a.GU + r * a.ldg; }
    else if (r < 3 * a.B) { const int64_t k = r - a.B; src = a.I + a.i_ids[k] * a.ldi; dst = a.GV + k * a.ldg; }
    else { const int64_t p = r - 3 * a.B; src = a.pref + p * 4 * a.nch; dst = a.gP + p * 4 * a.nch; scale = a.scale_pref; which = 1; }
    const float4 x = on ? reinterpret_cast<const float4*>(src)[lane] : f4zero();
    const float n2 = group_sum<GL>(dot4(x, x));
    if (n2 > 1.f) {
      if (on) {
        float4* g = reinterpret_cast<float4*>(dst) + lane;
        float4 v = *g;
        const float c = 2.f * scale;
        v.x = fmaf(c, x.x, v.x); v.y = fmaf(c, x.y, v.y); v.z = fmaf(c, x.z, v.z); v.w = fmaf(c, x.w, v.w);
        *g = v;
      }
      if (lane == 0) part[which] += scale * (n2 - 1.f);
    }
  }
  __shared__ float red[2][256 / 16];
  if (lane == 0) { red[0][threadIdx.x / GL] = part[0]; red[1][threadIdx.x / GL] = part[1]; }
  __syncthreads();
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int g = 0; g < GPB; ++g) t += red[threadIdx.x][g];
    if (t != 0.f) atomicAdd(a.loss + threadIdx.x, t);
  }
}

}  // namespace

extern "C" int ktup_train_rec_reg_rows(const float* U, int64_t ldu, const float* I, int64_t ldi, int d, const int64_t* u_ids,
                                       const int64_t* i_ids, int64_t B, float* GU, float* GV, const float* pref, int n_pref, float* gP,
                                       float scale_rows, float scale_pref, float* loss, void* stream) {
  const char* name = "ktup_train_rec_reg_rows";
  KTUP_REQUIRE(B >= 0 && n_pref >= 0, "%s: negative sizes", name);
  if (B == 0 && n_pref == 0) return KTUP_OK;
  KTUP_REQUIRE(U && I && u_ids && i_ids && GU && GV && loss && (n_pref == 0 || (pref && gP)), "%s: null pointer argument", name);
  if (d % 4 || d > 256 || (ldu | ldi) % 4 || !aligned16(U) || !aligned16(I) || !aligned16(GU) || !aligned16(GV) || !aligned16(pref) || !aligned16(gP))
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: needs d %% 4 == 0 (<= 256) and 16-byte aligned rows", name);
  RegRowsArgs a{U, I, ldu, ldi, d / 4, u_ids, i_ids, B, GU, GV, (int64_t)d, pref, n_pref, gP, scale_rows, scale_pref, loss};
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = 3 * B + n_pref;
#define KTUP_RR(GL)                                                                                                  \
  {                                                                                                                  \
    hipLaunchKernelGGL((reg_rows_kernel<GL>), dim3(grid_for((total + (256 / GL) - 1) / (256 / GL), 2048)), dim3(256), 0, st, a); \
    return check_launch(name);                                                                                       \
  }
  if (a.nch <= 16) KTUP_RR(16)
  if (a.nch <= 32) KTUP_RR(32)
  KTUP_RR(64)
#undef KTUP_RR
}

// 1 = this (step kind, d, n_pref) has a fused kernel; 0 = keep the multi-launch step.  kind: 0 rec (TUP / KTUP), 1 kg TransH, 2 kg TransE
extern "C" int ktup_train_step_supported(int kind, int d, int n_pref) {
  if (kind == 0) return n_pref > 0 && ((d == 256 && n_pref <= 20) || ((d == 64 || d == 100 || d == 128) && n_pref <= 32)) && opt_pref_mc();
  if (kind == 1 || kind == 2) return d > 0 && d % 4 == 0 && d <= 256;
  return 0;
}

extern "C" int ktup_train_rec_step(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                   const int32_t* item2ent, int64_t ent_pad, const float* pref, const float* pref_norm,
                                   const float* rel, const float* norm, int64_t ldp, int n_pref, int d, const int64_t* u_ids,
                                   const int64_t* i_ids, int64_t B, int l1, int gumbel_mode, const float* uniform, uint64_t seed,
                                   uint64_t offset, float target, float gscale, int orth, float* loss, float* gU, float* gI, float* gE,
                                   float* gP, float* gPn, float* gR, float* gRn, double* gnorm, void* stream) {
  const char* name = "ktup_train_rec_step";
  KTUP_REQUIRE(B >= 0 && (B > 0 || !gnorm), "%s: negative batch (or an empty one with a tracked norm)", name);
  if (B == 0) return KTUP_OK;
  KTUP_REQUIRE(U && I && pref && pref_norm && u_ids && i_ids && loss && gU && gI && gP && gPn, "%s: null pointer argument", name);
  KTUP_REQUIRE((E == nullptr) == (item2ent == nullptr) && (!E || gE), "%s: E, item2ent and gE go together", name);
  KTUP_REQUIRE((rel == nullptr) == (norm == nullptr) && (!rel || (gR && gRn)), "%s: rel, norm and their gradients go together", name);
  KTUP_REQUIRE(ldp == d, "%s: preference-side tables and their gradients must be contiguous (pitch d)", name);
  KTUP_REQUIRE(aligned16(U) && aligned16(I) && aligned16(E) && aligned16(pref) && aligned16(pref_norm) && aligned16(rel) && aligned16(norm) &&
                   aligned16(gU) && aligned16(gI) && aligned16(gE) && aligned16(gP) && aligned16(gPn) && aligned16(gR) && aligned16(gRn),
               "%s: tables and gradients must be 16-byte aligned", name);
  KTUP_REQUIRE(gumbel_mode >= KTUP_GUMBEL_OFF && gumbel_mode <= KTUP_GUMBEL_PHILOX_DEV, "%s: bad gumbel_mode %d", name, gumbel_mode);
  KTUP_REQUIRE((gumbel_mode != KTUP_GUMBEL_INPUT && gumbel_mode != KTUP_GUMBEL_PHILOX_DEV) || uniform,
               "%s: KTUP_GUMBEL_INPUT / KTUP_GUMBEL_PHILOX_DEV need the `uniform` pointer", name);
  const int rc = pref_step_mc(U, ldu, I, ldi, E, lde, item2ent, ent_pad, pref, pref_norm, rel, norm, ldp, n_pref, d, u_ids, i_ids, B, l1,
                              gumbel_mode, uniform, seed, offset, target, gscale, orth, loss, gU, gI, gE, gP, gPn, gR, gRn,
                              (hipStream_t)stream, name, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 1, gnorm);
  if (rc == 1) return set_error(KTUP_ERR_UNSUPPORTED, "%s: no fused kernel for d=%d, n_pref=%d (see ktup_train_step_supported)", name, d, n_pref);
  return rc;
}

// The same step with the row gradients of pair k written to rows k of GU / GV (plain stores, 2B x d each; GV is the gradient of
// the item row AND of its entity row) instead of float atomics into table-shaped buffers: for batches whose rows are then reduced
// by sorted segments (config 5: ktup_shard_reduce_rows).  gR / gRn may be NULL with rel / norm given: the caller then applies
// gP / gPn to both summands of the mixed tables itself (they are the same numbers).
extern "C" size_t ktup_train_rec_step_rows_ws_bytes(int64_t B, int n_pref, int d) { return pref_step_small_ws_bytes(B, n_pref, d); }

extern "C" int ktup_train_rec_step_rows(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                        const int32_t* item2ent, int64_t ent_pad, const float* pref, const float* pref_norm,
                                        const float* rel, const float* norm, int64_t ldp, int n_pref, int d, const int64_t* u_ids,
                                        const int64_t* i_ids, int64_t B, int l1, float target, float gscale, int orth, float* loss,
                                        float* GU, float* GV, float* gP, float* gPn, float* gR, float* gRn, double* sumsq, int n_slots,
                                        const int64_t* neg_ids, const int64_t* cursor, int64_t n_batches, int gumbel_mode,
                                        const void* gumbel, void* stream) {
  return ktup_train_rec_step_rows_ws(U, ldu, I, ldi, E, lde, item2ent, ent_pad, pref, pref_norm, rel, norm, ldp, n_pref, d, u_ids, i_ids, B, l1,
                                     target, gscale, orth, loss, GU, GV, gP, gPn, gR, gRn, sumsq, n_slots, neg_ids, cursor, n_batches, gumbel_mode,
                                     gumbel, nullptr, 0, stream);
}

extern "C" int ktup_train_rec_step_rows_ws(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* E, int64_t lde,
                                           const int32_t* item2ent, int64_t ent_pad, const float* pref, const float* pref_norm,
                                           const float* rel, const float* norm, int64_t ldp, int n_pref, int d, const int64_t* u_ids,
                                           const int64_t* i_ids, int64_t B, int l1, float target, float gscale, int orth, float* loss,
                                           float* GU, float* GV, float* gP, float* gPn, float* gR, float* gRn, double* sumsq, int n_slots,
                                           const int64_t* neg_ids, const int64_t* cursor, int64_t n_batches, int gumbel_mode,
                                           const void* gumbel, void* small_ws, size_t small_ws_bytes, void* stream) {
  const char* name = "ktup_train_rec_step_rows";
  KTUP_REQUIRE(gumbel_mode == KTUP_GUMBEL_OFF || ((gumbel_mode == KTUP_GUMBEL_INPUT || gumbel_mode == KTUP_GUMBEL_PHILOX_DEV) && gumbel),
               "%s: the ST-Gumbel gate takes its uniforms (KTUP_GUMBEL_INPUT: 2B x n_pref floats, positives then negatives) or a device-resident "
               "Philox position (KTUP_GUMBEL_PHILOX_DEV: uint64 {seed, offset})", name);
  KTUP_REQUIRE(B >= 0, "%s: negative batch", name);
  if (B == 0) return KTUP_OK;
  KTUP_REQUIRE(U && I && pref && pref_norm && u_ids && i_ids && loss && GU && GV && gP && gPn, "%s: null pointer argument", name);
  KTUP_REQUIRE((E == nullptr) == (item2ent == nullptr), "%s: E and item2ent go together", name);
  KTUP_REQUIRE((rel == nullptr) == (norm == nullptr) && (gR == nullptr) == (gRn == nullptr) && (rel || !gR), "%s: rel / norm (and gR / gRn) go together", name);
  KTUP_REQUIRE(!orth || !rel || gR, "%s: orthogonalLoss(pref, pref_norm) needs separate gradients for pref and rel", name);
  KTUP_REQUIRE(ldp == d, "%s: preference-side tables and their gradients must be contiguous (pitch d)", name);
  KTUP_REQUIRE(!sumsq || n_slots >= 1, "%s: the sum of squares needs at least one slot", name);
  KTUP_REQUIRE(!neg_ids || n_batches > 0, "%s: id columns need n_batches > 0", name);
  KTUP_REQUIRE(aligned16(U) && aligned16(I) && aligned16(E) && aligned16(pref) && aligned16(pref_norm) && aligned16(rel) && aligned16(norm) &&
                   aligned16(GU) && aligned16(GV) && aligned16(gP) && aligned16(gPn) && aligned16(gR) && aligned16(gRn),
               "%s: tables and gradients must be 16-byte aligned", name);
  const int rc = pref_step_mc(U, ldu, I, ldi, E, lde, item2ent, ent_pad, pref, pref_norm, rel, norm, ldp, n_pref, d, u_ids, i_ids, B, l1,
                              gumbel_mode, reinterpret_cast<const float*>(gumbel), 0, 0, target, gscale, orth, loss, nullptr, nullptr, nullptr, gP, gPn, gR, gRn,
                              (hipStream_t)stream, name, GU, GV, sumsq, n_slots, neg_ids, cursor, n_batches, nullptr, small_ws, small_ws_bytes);
  if (rc == 1) return set_error(KTUP_ERR_UNSUPPORTED, "%s: no fused kernel for d=%d, n_pref=%d (see ktup_train_step_supported)", name, d, n_pref);
  return rc;
}

extern "C" int ktup_train_kg_step(int transh, const float* E, int64_t lde, const float* R, int64_t ldr, const float* Nrm, int64_t ldn,
                                  int d, const int64_t* h, const int64_t* t, const int64_t* r, int64_t B, int l1, float margin,
                                  float gscale, int regs, float* loss, float* gE, float* gR, float* gN, double* gnorm, void* stream) {
  const char* name = "ktup_train_kg_step";
  KTUP_REQUIRE(B >= 0 && (B > 0 || !gnorm), "%s: negative batch (or an empty one with a tracked norm)", name);
  if (B == 0) return KTUP_OK;
  KTUP_REQUIRE(E && R && h && t && r && loss && gE && gR && (!transh || (Nrm && gN)), "%s: null pointer argument", name);
  if (d <= 0 || d % 4 || d > 256 || (lde | ldr | (transh ? ldn : 0)) % 4 || !aligned16(E) || !aligned16(R) || !aligned16(gE) || !aligned16(gR) ||
      (transh && (!aligned16(Nrm) || !aligned16(gN))))
    return set_error(KTUP_ERR_UNSUPPORTED, "%s: needs d %% 4 == 0 (<= 256) and 16-byte aligned rows", name);
  KgStepArgs a{E, R, Nrm, lde, ldr, ldn, h, t, r, B, d / 4, l1 != 0, margin, gscale, regs, loss, gE, gR, gN, gnorm, opt_deterministic() != 0};
  return transh ? launch_kg<true>(a, (hipStream_t)stream, name) : launch_kg<false>(a, (hipStream_t)stream, name);
}
