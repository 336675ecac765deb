// Lane-swap reductions over the 4 "k-quarter" lanes l, l^16, l^32, l^48 of the 16x16x4 MFMA layouts (gfx950).
#pragma once
#include "ktup_common.h"

namespace ktup {

typedef float v4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// gfx950 lane swaps (VALU, no LDS): permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its
// second; permlane16_swap exchanges the odd 16-lane rows of the first with the even rows of the second.
KTUP_DEV float swap32_sum(float a, float b) {   // lanes 0-31: a[l] + a[l+32]; lanes 32-63: b[l-32] + b[l]
  const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
KTUP_DEV float swap16_sum(float a, float b) {   // even rows: a[row] + a[row+1]; odd rows: b[row-1] + b[row]
  const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
KTUP_DEV float allsum_kq(float v) {             // sum over the 4 lanes l, l^16, l^32, l^48, in all of them
  v = swap32_sum(v, v);
  return swap16_sum(v, v);
}
// reduce-scatter over the same 4 lanes: lane with kq = l >> 4 ends up with the 4-lane sum of component kq
KTUP_DEV float scatter_kq(const v4& a) {
  const float v0 = swap32_sum(a[0], a[2]);      // kq 0,1: component 0; kq 2,3: component 2   (summed over l, l^32)
  const float v1 = swap32_sum(a[1], a[3]);      // kq 0,1: component 1; kq 2,3: component 3
  return swap16_sum(v0, v1);                    // row kq: component kq
}

}  // namespace ktup
