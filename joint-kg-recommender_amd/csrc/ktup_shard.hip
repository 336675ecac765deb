// Row pack / unpack for row-sharded embedding tables (config 5: tables partitioned by `row % world` across the GPUs of
// one node, lookups exchanged with RCCL all-to-all over xGMI).  The reference has no distributed code; these are the
// two device-side halves of the exchange:
//   pack  : out[k, :]  = table[ids[k], :]        (owner side: rows requested by a peer -> contiguous send buffer)
//   unpack: gtable[ids[k], :] += rows[k, :]      (owner side: row gradients coming back -> shard gradient, atomics)
// Same lane-group-per-row mapping as the scorers (contiguous 16-B pieces per row).
#include "ktup_rows.h"

using namespace ktup;

namespace {

struct PackRows {
  const float* T; int64_t ldt; const int64_t* ids; float* out; int64_t ldo;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    V x[CPL];
    cx.load(x, T + ids[row] * ldt);
    V* o = reinterpret_cast<V*>(out + row * ldo);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = cx.lane + j * G;
      if (c < cx.nch) o[c] = x[j];
    }
  }
};

struct UnpackAdd {
  const float* rows; int64_t ldr; const int64_t* ids; float* gT; int64_t ldg;
  template <typename V, int G, int CPL>
  KTUP_DEV void run(const RowCtx<V, G, CPL>& cx, int64_t row) const {
    V x[CPL];
    cx.load(x, rows + row * ldr);
    cx.scatter_add(gT + ids[row] * ldg, x);
  }
};

}  // namespace

extern "C" int ktup_shard_pack_rows(const float* table, int64_t ldt, int d, const int64_t* ids, int64_t n, float* out,
                                    int64_t ldo, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && ldo >= d, "ktup_shard_pack_rows: bad sizes");
  KTUP_REQUIRE(n == 0 || (table && ids && out), "ktup_shard_pack_rows: null pointer argument");
  PackRows op{table, ldt, ids, out, ldo};
  return launch_rows(op, d, can_vec4(d, {table, out}, {ldt, ldo}), n, (hipStream_t)stream, "ktup_shard_pack_rows");
}

extern "C" int ktup_shard_unpack_rows_add(const float* rows, int64_t ldr, int d, const int64_t* ids, int64_t n, float* gtable,
                                          int64_t ldg, void* stream) {
  KTUP_REQUIRE(d > 0 && n >= 0 && ldr >= d, "ktup_shard_unpack_rows_add: bad sizes");
  KTUP_REQUIRE(n == 0 || (rows && ids && gtable), "ktup_shard_unpack_rows_add: null pointer argument");
  UnpackAdd op{rows, ldr, ids, gtable, ldg};
  return launch_rows(op, d, can_vec4(d, {rows, gtable}, {ldr, ldg}), n, (hipStream_t)stream, "ktup_shard_unpack_rows_add");
}
