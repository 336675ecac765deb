// Shared host/device helpers for libktup_hip.so (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ktup_hip.h"

namespace ktup {

// ---- host side -------------------------------------------------------------------------------
int set_error(int code, const char* fmt, ...);  // records a thread-local message, returns `code`
int check_launch(const char* what);             // hipGetLastError() -> KTUP_OK / KTUP_ERR_LAUNCH
// process-wide options (ktup_runtime.hip): read from the environment once at load, changed by ktup_set_option
int opt_pref_mc();
int opt_eval_mc();
int opt_rank_chunk();
int opt_seg_bwd_min();
int opt_bwd_wide_max();
int opt_dbg_noflush();
int opt_shard_chunk();
int opt_nt_gather();
int opt_kg_wtab();
int opt_eval_nsplit();
int opt_dbg_eval();
int opt_wide_waves();
int opt_kg_exact();
int opt_deterministic();
int opt_fwd_wide();
hipStream_t fork_side(hipStream_t st);              // ktup_runtime.hip: second stream for input-only work (nullptr: stay on st)
void join_side(hipStream_t st, hipStream_t side);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Memory-bound launches: enough workgroups to fill 256 CUs x several resident blocks, grid-stride the rest.
inline int grid_for(int64_t work_blocks, int max_blocks = 256 * 8) {
  if (work_blocks < 1) work_blocks = 1;
  return (int)(work_blocks < max_blocks ? work_blocks : max_blocks);
}

#define KTUP_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return ::ktup::set_error(KTUP_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

// ---- device side -----------------------------------------------------------------------------
#define KTUP_DEV __device__ __forceinline__

// Sum across G consecutive lanes (G = 16/32/64); every lane of the group gets the total.
template <int G>
KTUP_DEV float group_sum(float v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

KTUP_DEV float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
KTUP_DEV float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
KTUP_DEV float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
KTUP_DEV float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
KTUP_DEV float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
KTUP_DEV float4 fma4(float s, float4 a, float4 c) {  // c + s*a
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

// The reference's two dissimilarities: L1 = sum|z|, "L2" = sum z^2 (SQUARED, no sqrt; transE.py:57-61).
KTUP_DEV float dist1(float z, bool l1) { return l1 ? fabsf(z) : z * z; }
KTUP_DEV float dist4(float4 z, bool l1) { return dist1(z.x, l1) + dist1(z.y, l1) + dist1(z.z, l1) + dist1(z.w, l1); }
// d dist / dz : torch's abs backward is sign(z) with sign(0) = 0; pow(2) backward is 2z.
KTUP_DEV float ddist1(float z, bool l1) { return l1 ? (z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f)) : 2.f * z; }
KTUP_DEV float4 ddist4(float4 z, bool l1) { return make_float4(ddist1(z.x, l1), ddist1(z.y, l1), ddist1(z.z, l1), ddist1(z.w, l1)); }

KTUP_DEV void atomic_add4(float* p, float4 v) {
  atomicAdd(p + 0, v.x);
  atomicAdd(p + 1, v.y);
  atomicAdd(p + 2, v.z);
  atomicAdd(p + 3, v.w);
}

// ---- the gradient norm without a pass over the gradients (the B = 512 fused step).  A kernel that BUILDS zero-filled gradient buffers
// with atomics can track the squared norm of what it builds: adding v onto a cell that held `old` raises the buffer's squared norm by
// (2 old + v) v, and the returning form of the atomic hands `old` back.  Exact (up to fp32 rounding of the terms) as long as EVERY add
// into the buffers goes through these.  The per-workgroup totals go to a small workspace the optimizer launch reads (gnorm_* below,
// ktup_optim.hip clip_step_kernel): no grid barrier around a norm pass any more.
// Issue first, use later: a wave that consumes each returned value right after its atomic pays one memory round trip (~0.35 us) per
// atomic -- the rec step's ~40 dependent ones cost 13 us.  The callers collect the old values of a whole group of adds in registers
// and fold them into their sum once, after the last add of the group was issued.
KTUP_DEV float sq_gain(float old, float v) { return fmaf(2.f, old, v) * v; }
KTUP_DEV float4 atomic_add4_old(float* p, float4 v) {
  float4 o;
  o.x = atomicAdd(p + 0, v.x); o.y = atomicAdd(p + 1, v.y); o.z = atomicAdd(p + 2, v.z); o.w = atomicAdd(p + 3, v.w);
  return o;
}
KTUP_DEV float sq_gain4(float4 o, float4 v) { return (sq_gain(o.x, v.x) + sq_gain(o.y, v.y)) + (sq_gain(o.z, v.z) + sq_gain(o.w, v.w)); }
// Workspace (KTUP_GNORM_WS_DOUBLES doubles, zero-filled once by the caller): word 0 = steps taken (its low bit picks the slot set the
// NEXT building kernel adds to), word 1 = the set the last building kernel used, double 2 = the last squared norm (for the host),
// doubles 8 .. 8 + 2 x 16 = two sets of 16 slots.  The building kernel adds to set (steps & 1); the optimizer launch sums that set,
// clears the OTHER one (idle until the next building kernel) and bumps `steps` -- no set is ever cleared while someone adds to it.
constexpr int GNORM_SLOTS = 16, GNORM_SET0 = 8;
KTUP_DEV int gnorm_set(const double* ws) { return (int)(reinterpret_cast<const unsigned long long*>(ws)[0] & 1ull); }
KTUP_DEV void gnorm_add(double* ws, int set, double wg_total) {      // one thread per workgroup
  if (wg_total != 0.0) atomicAdd(ws + GNORM_SET0 + GNORM_SLOTS * set + (blockIdx.x % GNORM_SLOTS), wg_total);
  if (blockIdx.x == 0) reinterpret_cast<unsigned long long*>(ws)[1] = (unsigned long long)set;
}

// ST-Gumbel noise, transUP.py:159-162 : g = -log(-log(u + 1e-20) + 1e-20)
KTUP_DEV float gumbel_from_uniform(float u) { return -logf(-logf(u + 1e-20f) + 1e-20f); }

// Philox4x32-10 (counter-based; production Gumbel / negative-sampling draws).
// KTUP_GUMBEL_PHILOX_DEV: the stream position lives in device memory (graph replay); resolve it into the by-value
// argument block at kernel entry (uniform address -> scalar loads) and continue as KTUP_GUMBEL_PHILOX.
#define KTUP_RESOLVE_GUMBEL(a)                                              \
  do {                                                                      \
    if ((a).gumbel == KTUP_GUMBEL_PHILOX_DEV) {                             \
      const uint64_t* ktup_gs_ = reinterpret_cast<const uint64_t*>((a).uniform); \
      (a).seed = ktup_gs_[0];                                               \
      (a).offset = ktup_gs_[1];                                             \
      (a).gumbel = KTUP_GUMBEL_PHILOX;                                      \
    }                                                                       \
  } while (0)

struct Philox {
  uint32_t k0, k1;
  KTUP_DEV Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  KTUP_DEV uint4 operator()(uint64_t ctr_lo, uint64_t ctr_hi) const {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low one
      const uint64_t m0 = (uint64_t)0xD2511F53u * (uint64_t)c0, m1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
      const uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0, hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
      const uint32_t n0 = hi1 ^ c1 ^ a, n1 = lo1, n2 = hi0 ^ c3 ^ b, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
  }
};
// 24-bit uniform in [0, 1), the same lattice torch's uniform_() draws from.
KTUP_DEV float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }


// ---- the hard (ST-Gumbel) gate's choice: arg max_p logit(p) + Gumbel(u_p), first maximum on ties (the one-hot of
// transUP.py:45-60 st_gumbel_softmax).  u_p: Philox draw at stream position base + p (+ offset), or uniform[base + p] (parity mode).
// With Philox draws the transform first runs on the hardware logarithm (v_log_f32): over EVERY uniform of the 24-bit lattice it is within
// 1.94e-6 of the logf form (tools/gumbel_log_check.hip, exhaustive), so a winner that leads the runner-up by more than GATE_MARGIN is
// the logf form's winner too; lanes with a closer call redo the choice with logf.  Input uniforms always take the logf form.
KTUP_DEV float fast_gumbel_from_uniform(float u) {
  constexpr float ln2 = 0.69314718055994530942f;
  return -ln2 * __builtin_amdgcn_logf(-ln2 * __builtin_amdgcn_logf(u + 1e-20f) + 1e-20f);
}
// MAXP > 0: the loop over the preferences is unrolled to MAXP steps (logit(p) may then index registers)
struct GatePick { int ps; float best, second; };     // (returned by value: reference outputs kept best / second in scratch memory,
                                                     //  a memory round trip per preference -- the waves waited 57 % of the time)
template <bool FAST, int MAXP, typename LogitFn>
KTUP_DEV GatePick gate_argmax_pass(int P, uint64_t base, bool input, const float* uniform, uint64_t seed, uint64_t offset, LogitFn logit) {
  int ps = 0;
  float best = -__builtin_inff(), second = best;
  auto take = [&](int p, float u) {
    const float v = logit(p) + (FAST ? fast_gumbel_from_uniform(u) : gumbel_from_uniform(u));
    if (v > best) { second = best; best = v; ps = p; }
    else if (v > second) second = v;
  };
  if (!input && (P & 3) == 0 && MAXP == 0) {
    // a pair's P draws start at stream position base + offset with base a multiple of P, hence of 4: every lane sits at the same place
    // `sh` inside its first Philox block, and the walk goes block by block -- one 64-bit add per pair instead of one (plus a shift, a
    // compare and a component select) per preference
    const uint64_t first = base + offset;
    const int sh = __builtin_amdgcn_readfirstlane((int)((uint32_t)offset & 3u));
    uint64_t blk = first >> 2;
    for (int p0 = -sh; p0 < P; p0 += 4, ++blk) {
      const uint4 r = Philox(seed)(blk, 0x4b545550ull);
      if (p0 >= 0) take(p0, u01(r.x));
      if (p0 + 1 >= 0 && p0 + 1 < P) take(p0 + 1, u01(r.y));
      if (p0 + 2 >= 0 && p0 + 2 < P) take(p0 + 2, u01(r.z));
      if (p0 + 3 < P) take(p0 + 3, u01(r.w));
    }
    return GatePick{ps, best, second};
  }
  uint64_t blk = ~0ull;              // the Philox block (4 draws) in hand: consecutive preferences share it
  uint4 r = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int p = 0; p < (MAXP > 0 ? MAXP : P); ++p) {
    if (MAXP == 0 || p < P) {
      float u;
      if (input) {
        u = uniform[base + p];
      } else {
        const uint64_t idx = base + p + offset;
        if ((idx >> 2) != blk) {
          blk = idx >> 2;
          r = Philox(seed)(blk, 0x4b545550ull);
        }
        u = u01((idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w);
      }
      take(p, u);
    }
  }
  return GatePick{ps, best, second};
}
template <int MAXP = 0, typename LogitFn>
KTUP_DEV int gate_argmax(int P, uint64_t base, bool input, const float* uniform, uint64_t seed, uint64_t offset, LogitFn logit) {
  if (input) return gate_argmax_pass<false, MAXP>(P, base, true, uniform, seed, offset, logit).ps;
  const GatePick f = gate_argmax_pass<true, MAXP>(P, base, false, uniform, seed, offset, logit);
  int ps = f.ps;
  // two sums off by <= 1.94e-6 + an ulp each: 4e-6 + 5e-7 |best| bounds the error of the lead; four times that decides
  const bool close = !(f.best - f.second > 2e-5f + 2e-6f * fabsf(f.best));       // (NaNs land here too)
  if (__builtin_amdgcn_ballot_w64(close)) {
    if (close) ps = gate_argmax_pass<false, MAXP>(P, base, false, uniform, seed, offset, logit).ps;
  }
  return ps;
}

}  // namespace ktup
