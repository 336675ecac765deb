// Error reporting and process-wide options of libktup_hip.so.
//
// Errors: thread-local message, nothing shared.  Options: three integer knobs for tests and A/B runs, seeded ONCE from the
// environment when the library is loaded (never re-read on the launch path) and changeable through ktup_set_option; they
// select between kernels that compute the same results, so no caller depends on them for correctness.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ktup_common.h"

namespace ktup {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(KTUP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return KTUP_OK;
}

namespace {
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
struct Option { const char* name; const char* env; std::atomic<int> value; };
Option g_opts[] = {
    {"pref_mc", "KTUP_PREF_MC", {env_int("KTUP_PREF_MC", 1)}},          // 0: generic K5-K7 kernels even where the matrix-core ones apply
    {"eval_mc", "KTUP_EVAL_MC", {env_int("KTUP_EVAL_MC", 1)}},          // 0: VALU evaluation kernels instead of the matrix-core ones
    {"rank_chunk", "KTUP_RANK_CHUNK", {env_int("KTUP_RANK_CHUNK", 0)}}, // > 0: force the chunked ranking kernels with this chunk size
    {"seg_bwd_min", "KTUP_SEG_BWD_MIN", {env_int("KTUP_SEG_BWD_MIN", 8192)}},   // rows from which the backward kernels reduce row gradients by segments (0: never)
    {"bwd_wide_max", "KTUP_BWD_WIDE_MAX", {env_int("KTUP_BWD_WIDE_MAX", 4096)}},   // K5-K7 backward: pairs up to which four waves share a 16-pair tile (d <= 128)
    {"side_sort", "KTUP_SIDE_SORT", {env_int("KTUP_SIDE_SORT", 1)}},    // 0: the id sorts of the segment reductions stay on the caller's stream
    {"shard_chunk", "KTUP_SHARD_CHUNK", {env_int("KTUP_SHARD_CHUNK", 0)}},   // > 0: sorted entries per lane group in the sharded step's two reduction walks (0: by batch size)
    {"nt_gather", "KTUP_NT_GATHER", {env_int("KTUP_NT_GATHER", 0)}},         // 1: K5-K7 forward gathers its rows with the nontemporal hint (tables >> Infinity Cache)
    // the two MEASUREMENT-ONLY knobs (wrong results) are never seeded from the environment: only ktup_set_option turns them on, and it says so on stderr
    {"dbg_noflush", "", {0}},                                              // the fused step kernels skip the small-table gradient flush
    {"eval_nsplit", "KTUP_EVAL_NSPLIT", {env_int("KTUP_EVAL_NSPLIT", 0)}},  // > 0: catalogue splits of the one-sweep rec evaluation (0: by occupancy, at most 8)
    {"dbg_eval", "", {0}},                                                  // bits switch phases of the rec evaluation sweep off
    {"kg_wtab", "KTUP_KG_WTAB", {env_int("KTUP_KG_WTAB", 1)}},              // 0: the fused TransH link-prediction pass computes w.e in the sweep instead of once per (relation, candidate)
    {"kg_exact", "KTUP_KG_EXACT", {env_int("KTUP_KG_EXACT", 1)}},            // 0: the fused squared-L2 link-prediction pass ranks by its own fp32 scores alone (no fp64 referee near the golds)
    {"wide_waves", "KTUP_WIDE_WAVES", {env_int("KTUP_WIDE_WAVES", 4)}},     // d = 256 coordinate-sliced K5-K7 backward / fused step: waves that share a 16-pair tile: 4, or 8 (two per SIMD: 1 us faster alone,
                                                                            // but its 2 x 240 registers per SIMD leave no room for the route's kernels beside it: 0.149 against 0.139 ms per config-5 step)
    {"fwd_wide", "KTUP_FWD_WIDE", {env_int("KTUP_FWD_WIDE", 1)}},          // 0: K5-K7 forward at d = 256 keeps one wave per 16-pair tile for every batch size
    {"deterministic", "KTUP_DETERMINISTIC", {env_int("KTUP_DETERMINISTIC", 0)}},   // 1: ktup_train_rec_step / ktup_train_kg_step issue every gradient add from ONE workgroup in program order (parity runs: the same
                                                                            // bits on every run; ~1 ms per B = 512 step instead of ~0.02)
};
Option* find(const char* name) {
  for (auto& o : g_opts)
    if (name && strcmp(name, o.name) == 0) return &o;
  return nullptr;
}
}  // namespace

int opt_pref_mc() { return g_opts[0].value.load(std::memory_order_relaxed); }
int opt_eval_mc() { return g_opts[1].value.load(std::memory_order_relaxed); }
int opt_rank_chunk() { return g_opts[2].value.load(std::memory_order_relaxed); }
int opt_seg_bwd_min() { return g_opts[3].value.load(std::memory_order_relaxed); }
int opt_bwd_wide_max() { return g_opts[4].value.load(std::memory_order_relaxed); }
int opt_shard_chunk() { return g_opts[6].value.load(std::memory_order_relaxed); }
int opt_nt_gather() { return g_opts[7].value.load(std::memory_order_relaxed); }
int opt_dbg_noflush() { return g_opts[8].value.load(std::memory_order_relaxed); }
int opt_eval_nsplit() { return g_opts[9].value.load(std::memory_order_relaxed); }
int opt_dbg_eval() { return g_opts[10].value.load(std::memory_order_relaxed); }
int opt_kg_wtab() { return g_opts[11].value.load(std::memory_order_relaxed); }
int opt_kg_exact() { return g_opts[12].value.load(std::memory_order_relaxed); }
int opt_wide_waves() { return g_opts[13].value.load(std::memory_order_relaxed); }
int opt_fwd_wide() { return g_opts[14].value.load(std::memory_order_relaxed); }
int opt_deterministic() { return g_opts[15].value.load(std::memory_order_relaxed); }

// A library-owned second stream for work that depends only on a call's INPUTS (the counting sorts of the segment reductions)
// while the caller's stream runs the kernel that produces the data: fork_side makes it wait for everything enqueued on `st` so far,
// join_side makes `st` wait for it.  Returns nullptr (the caller stays on `st`) when switched off, while `st` is being captured
// into a graph, or when the stream / events cannot be made.
namespace {
struct Side {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool tried = false;
} g_side;
}  // namespace

hipStream_t fork_side(hipStream_t st) {
  if (!g_opts[5].value.load(std::memory_order_relaxed)) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (!g_side.tried) {
    g_side.tried = true;
    if (hipStreamCreateWithFlags(&g_side.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      g_side.stream = nullptr;
    }
  }
  if (!g_side.stream) return nullptr;
  if (hipEventRecord(g_side.fork, st) != hipSuccess || hipStreamWaitEvent(g_side.stream, g_side.fork, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return g_side.stream;
}

void join_side(hipStream_t st, hipStream_t side) {
  if (!side) return;
  (void)hipEventRecord(g_side.join, side);
  (void)hipStreamWaitEvent(st, g_side.join, 0);
}

}  // namespace ktup

extern "C" int ktup_version(void) { return 2; }
extern "C" const char* ktup_last_error(void) { return ktup::g_err; }

extern "C" int ktup_set_option(const char* name, int value) {
  ktup::Option* o = ktup::find(name);
  if (!o) return ktup::set_error(KTUP_ERR_INVALID_ARG, "ktup_set_option: unknown option '%s'", name ? name : "(null)");
  if (value != 0 && strncmp(o->name, "dbg_", 4) == 0)
    fprintf(stderr, "libktup_hip: option %s = %d is a MEASUREMENT-ONLY knob: kernels now skip work and produce WRONG results until it is set back to 0\n", o->name, value);
  o->value.store(value, std::memory_order_relaxed);
  return KTUP_OK;
}

extern "C" int ktup_get_option(const char* name, int* value) {
  ktup::Option* o = ktup::find(name);
  if (!o || !value) return ktup::set_error(KTUP_ERR_INVALID_ARG, "ktup_get_option: unknown option '%s'", name ? name : "(null)");
  *value = o->value.load(std::memory_order_relaxed);
  return KTUP_OK;
}
