// lins_solve6.h — the 6 x 6 solve of the IESKF step, (sigma^2 I + A P_SS) w = g + A d_S (the
// push-through form of SE:542-549), as a Gauss-Jordan elimination.  This scalar form is the definition: the
// wave-spread version the kernels run (ieskf_rowsum.h wave_gj_solve6) performs exactly these operations on every
// element, in this order, and is tested to return the same bits; this one compiles for the host as well, where
// tests/test_fastmath.py holds it against numpy.
//
// Why Gauss-Jordan and not elimination + back-substitution: on the device the system sits one element per lane and
// one wave walks through the solve while the rest of the workgroup waits, so what counts is the length of the
// dependent chain.  Back-substitution is six divisions and fifteen multiply-subtracts one after the other;
// normalising the pivot row and clearing the column above AND below the pivot costs no extra step per column (the
// lanes of the upper rows were idle anyway) and leaves the solution in the last column — no second phase.
//   pivot of column k: the row r >= k with the largest |a_rk|, compared on the HIGH WORDS of the doubles (sign
//   cleared; first maximum wins): one scalar integer compare per candidate instead of an f64 compare through the
//   vector unit.  Magnitudes that differ only below bit 32 are equally good pivots.  A NaN compares above every
//   number, is chosen, and poisons the solution — which is what the divergence test of SE:552-563 must see.
#pragma once

#include <stdint.h>
#include <string.h>

#include "lins_math.h"

namespace lins {

LINS_HD uint32_t abs_hi_word(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__double2hiint(v) & 0x7FFFFFFFu;
#else
  uint64_t b;
  memcpy(&b, &v, 8);
  return (uint32_t)(b >> 32) & 0x7FFFFFFFu;
#endif
}

LINS_HD void gj_solve6(double (&a)[6][7], double (&x)[6]) {
  for (int k = 0; k < 6; ++k) {
    int p = k;
    uint32_t best = abs_hi_word(a[k][k]);
    for (int r = k + 1; r < 6; ++r) {
      const uint32_t h = abs_hi_word(a[r][k]);
      if (h > best) best = h, p = r;
    }
    if (p != k)
      for (int j = 0; j < 7; ++j) {
        const double t = a[k][j];
        a[k][j] = a[p][j], a[p][j] = t;
      }
    const double inv = 1.0 / a[k][k];
    double nk[7];
    for (int j = k + 1; j < 7; ++j) nk[j] = a[k][j] * inv;  // the normalised pivot row
    for (int i = 0; i < 6; ++i)
      for (int j = k + 1; j < 7; ++j) a[i][j] = i == k ? nk[j] : a[i][j] - a[i][k] * nk[j];
  }
  for (int r = 0; r < 6; ++r) x[r] = a[r][6];
}

}  // namespace lins
