// Geometry of the prepared preference tables (ktup_pref_prepare workspace), shared by the training-path
// kernels (ktup_score_pref.hip) and the all-candidate evaluation kernels (ktup_eval.hip).
#pragma once
#include "ktup_common.h"

namespace ktup {

constexpr int TR = 64;  // pairs per tile (= lanes per wave)
constexpr int PB = 5;   // preferences handled per stage-1 pass by one wave

// Row pitch (in floats) of the prepared tables, and rows of the zero-padded logit table.
struct PrefGeom {
  int CH, NW;   // float4 chunks per lane in stage 2, waves per workgroup
  int dp;       // padded row length in floats = 4 * NW * CH
  int ppad;     // logit-table rows: multiple of NW * PB
  bool ok;
};
inline PrefGeom pref_geom(int d, int n_pref) {
  PrefGeom g{0, 0, 0, 0, false};
  if (d <= 0 || d % 4 || n_pref <= 0) return g;
  const int nch = d / 4;
  if (nch <= 16) { g.CH = 4; g.NW = 4; }
  else if (nch <= 28) { g.CH = 7; g.NW = 4; }
  else if (nch <= 32) { g.CH = 8; g.NW = 4; }
  else if (nch <= 64) { g.CH = 8; g.NW = 8; }
  else return g;
  g.dp = 4 * g.NW * g.CH;
  const int step = g.NW * PB;
  g.ppad = ((n_pref + step - 1) / step) * step;
  g.ok = true;
  return g;
}
// workspace layout (floats): Alog[ppad][dp] | Ar[P][dp] | Cn[P][dp]
inline size_t ws_floats(const PrefGeom& g, int n_pref) { return (size_t)(g.ppad + 2 * n_pref) * g.dp; }


// Constant-address-space view of a read-only table: a wave-uniform index selects s_load_dwordx4 and the
// value lives in SGPRs (one scalar operand per v_fmac), not in VGPRs / LDS.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f* sptr4;
KTUP_DEV sptr4 as_scalar(const float4* p) { return (sptr4)(uintptr_t)p; }
KTUP_DEV sptr4 as_scalar(const float* p) { return (sptr4)(uintptr_t)p; }
KTUP_DEV float4 sld(sptr4 p, int idx) { const v4f v = p[idx]; return make_float4(v.x, v.y, v.z, v.w); }

}  // namespace ktup
