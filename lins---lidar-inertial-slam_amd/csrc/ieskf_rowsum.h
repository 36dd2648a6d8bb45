// ieskf_rowsum.h — register-level building blocks shared by the persistent LDS kernel (ieskf_lds_impl.h) and
// (until round 3 also the list kernel of the split path): the 6x6 solve, the next iteration's constants and the
// wave-level reduction of the H rows to the 28 sums.  Pure functions of their arguments (no LDS, no globals).
#pragma once

#include <hip/hip_runtime.h>

#include "ieskf_device.h"
#include "lins_solve6.h"

namespace lins {

// ---------------------------------------------------------------------------
// 6x6 pivoted elimination, every lane of the wave redundantly in registers (the wave is
// one instruction stream anyway): no shuffles, no LDS traffic, no barriers.  Fully
// unrolled; row exchanges are value selects so nothing is dynamically indexed.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void reg_solve6(double (&a)[6][7], double (&x)[6]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double best = fabs(a[k][k]);
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      double v = fabs(a[i][k]);
      if (v > best) best = v, p = i;
    }
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const bool sw = (p == i);
#pragma unroll
      for (int j = k; j < 7; ++j) {
        double u = a[k][j], w = a[i][j];
        a[k][j] = sw ? w : u;
        a[i][j] = sw ? u : w;
      }
    }
    const double inv = 1.0 / a[k][k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const double f = a[i][k] * inv;
#pragma unroll
      for (int j = k + 1; j < 7; ++j) a[i][j] -= f * a[k][j];
    }
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double sacc = a[i][6];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) sacc -= a[i][k] * x[k];
    x[i] = sacc / a[i][i];
  }
}

// ---------------------------------------------------------------------------
// The same elimination spread over the lanes of ONE wave: lane l < 42 holds element (l / 7, l % 7) of the 6 x 7
// system [N | z].  Every element goes through exactly the operations reg_solve6 applies to it, in the same order
// (first-maximum pivot, row exchange, f = a_ik * (1 / a_kk), a_ij -= f * a_kj as a multiply and a subtract), so
// the solution is bit-identical — in ~1/3 of the instructions and a fraction of the dependent latency of the
// one-lane version (which runs 6 x 7 doubles through a single lane's registers).  Call with all 64 lanes active;
// returns x[0..5] in every lane.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double shfl_f64(double v, int src) {
  const int lo = __shfl(__double2loint(v), src), hi = __shfl(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// (a wave-uniform source lane: v_readlane, no LDS crossbar round trip)
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_solve6(double a, int lane, double (&x)[6]) {
  const int i = lane < 42 ? lane / 7 : 7, j = lane < 42 ? lane % 7 : 0;  // (lanes >= 42 take part in no update)
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double best = fabs(readlane_f64(a, k * 7 + k));
#pragma unroll
    for (int r = k + 1; r < 6; ++r) {
      const double v = fabs(readlane_f64(a, r * 7 + k));
      if (v > best) best = v, p = r;
    }
    if (p != k) {  // (wave-uniform) exchange rows k and p
      const int src = i == k ? p * 7 + j : (i == p ? k * 7 + j : lane);
      a = shfl_f64(a, src);
    }
    const double inv = 1.0 / readlane_f64(a, k * 7 + k);
    const double aik = shfl_f64(a, (i < 6 ? i : 0) * 7 + k), akj = shfl_f64(a, k * 7 + j);
    const double f = aik * inv;
    if (i > k && i < 6 && j > k) a -= f * akj;
  }
#pragma unroll
  for (int r = 5; r >= 0; --r) {
    double sacc = readlane_f64(a, r * 7 + 6);
#pragma unroll
    for (int k = r + 1; k < 6; ++k) sacc -= readlane_f64(a, r * 7 + k) * x[k];
    x[r] = sacc / readlane_f64(a, r * 7 + r);
  }
}

// ---------------------------------------------------------------------------
// The solve the kernels run since round 2: Gauss-Jordan (lins_solve6.h gj_solve6 — the definition, with the why)
// spread over the wave like wave_solve6 above.  Per column: the pivot by scalar integer compares on the high words,
// ONE division, two lane gathers, and every row — above and below the pivot — cleared in the same step; the solution
// is the last column, no back-substitution.  About a quarter of wave_solve6's dependent chain (which stays, with
// reg_solve6, as the reference the device test compares against within rounding).  Bit-identical to gj_solve6.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_gj_solve6(double a, int lane, double (&x)[6]) {
  const int i = lane < 42 ? lane / 7 : 7, j = lane < 42 ? lane % 7 : 0;  // (lanes >= 42 take part in no update)
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int p = k;
    unsigned best = (unsigned)__builtin_amdgcn_readlane(__double2hiint(a), k * 7 + k) & 0x7FFFFFFFu;
#pragma unroll
    for (int r = k + 1; r < 6; ++r) {
      const unsigned h = (unsigned)__builtin_amdgcn_readlane(__double2hiint(a), r * 7 + k) & 0x7FFFFFFFu;
      if (h > best) best = h, p = r;
    }
    if (p != k) {  // (wave-uniform) exchange rows k and p
      const int src = i == k ? p * 7 + j : (i == p ? k * 7 + j : lane);
      a = shfl_f64(a, src);
    }
    const double inv = 1.0 / readlane_f64(a, k * 7 + k);
    const double aik = shfl_f64(a, (i < 6 ? i : 0) * 7 + k), akj = shfl_f64(a, k * 7 + j);
    const double nkj = akj * inv;  // the normalised pivot row
    if (i < 6 && j > k) a = i == k ? nkj : a - aik * nkj;
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) x[r] = readlane_f64(a, r * 7 + 6);
}

// The same Gauss-Jordan for three right-hand sides at once: [N | B] is 6 x 9, element (i, j) in lane i * 9 + j
// (54 lanes); after the six pivot steps column 6 + r of row i holds (N^-1 B)(i, r).  Used twice (two waves, B = the left
// and the right half of the identity) for N^-1 in the Joseph epilogue.  Same pivoting rule as wave_gj_solve6.
__device__ __forceinline__ double wave_gj_solve6x3(double a, int lane) {
  const int i = lane < 54 ? lane / 9 : 7, j = lane < 54 ? lane % 9 : 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int p = k;
    unsigned best = (unsigned)__builtin_amdgcn_readlane(__double2hiint(a), k * 9 + k) & 0x7FFFFFFFu;
#pragma unroll
    for (int r = k + 1; r < 6; ++r) {
      const unsigned h = (unsigned)__builtin_amdgcn_readlane(__double2hiint(a), r * 9 + k) & 0x7FFFFFFFu;
      if (h > best) best = h, p = r;
    }
    if (p != k) {  // (wave-uniform) exchange rows k and p
      const int src = i == k ? p * 9 + j : (i == p ? k * 9 + j : lane);
      a = shfl_f64(a, src);
    }
    const double inv = 1.0 / readlane_f64(a, k * 9 + k);
    const double aik = shfl_f64(a, (i < 6 ? i : 0) * 9 + k), akj = shfl_f64(a, k * 9 + j);
    const double nkj = akj * inv;
    if (i < 6 && j > k) a = i == k ? nkj : a - aik * nkj;
  }
  return a;  // lanes i * 9 + 6 + r: (N^-1 B)(i, r)
}

// Rinvleft(-phi)^T and phi from a unit quaternion without libm sin/cos.  Small rotations (w > 0, tan(|phi|/2) <= 1/8:
// every scan-to-scan motion a lidar sees) take the short-series form phi_and_gt_small of lins_math.h — no square
// root, one division, two fma chains.  Otherwise: with h = |phi|/2 the half angle, cos h = |w| / |q| and
// sin h = |v| / |q| exactly, so s = h cot h needs only the atan2 that Quat2axis performs anyway (math_utils.h:75-88,
// 304-321; differs from the sin/cos route in the last ulp only).
__device__ __forceinline__ void phi_and_Gt_general(const Q4& q, V3& phi, M3& Gt) {
  const double mag = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  phi = V3{q.x, q.y, q.z};
  Gt = M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  if (!(mag >= 1e-10)) return;  // Quat2axis leaves v unscaled; |phi| < 1e-10 => Rinvleft = I
  const double ang = wrap_pi(2.0 * atan2(mag, q.w));
  const V3 u = V3{q.x, q.y, q.z} / mag;
  phi = ang * u;
  const double theta = norm(phi);
  if (theta < 1e-10) return;
  const double h = theta / 2.0;
  const double n = sqrt(q.w * q.w + mag * mag);
  const double s = h * ((fabs(q.w) / n) / (mag / n));
  const V3 a = V3{-phi.x, -phi.y, -phi.z} / theta;  // axis of -phi
  const M3 k = skew(a);
  const double av[3] = {a.x, a.y, a.z};
  // Rinvleft(-phi) = s I + (1 - s) a a^T - h [a]x ; store the transpose
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
      Gt.m[jj * 3 + i] = (s * (i == jj ? 1.0 : 0.0) + (1.0 - s) * av[i] * av[jj]) - h * k.m[i * 3 + jj];
}
__device__ __forceinline__ void phi_and_Gt(const Q4& q, V3& phi, M3& Gt) {
  if (!phi_and_gt_small(q, phi, Gt)) phi_and_Gt_general(q, phi, Gt);
}

// The constants of an iteration from its linearisation state, by the routes the serial tail between two iterations
// takes (phi_and_Gt, quat2axis_fast) — so that an update's first iteration, the list kernel's first iteration after
// the hand-over and every later iteration compute them alike.  (make_iter_const of ieskf_device.h, the libm route,
// stays with the any-size kernels.)
__device__ __forceinline__ void make_iter_const_tail(const double* filt, IterConst& ic) {
  const Q4 q{ic.lin[6], ic.lin[7], ic.lin[8], ic.lin[9]};
  ic.Rt = mtrans(qmat(q));
  phi_and_Gt(q, ic.phi, ic.Gt);
  const Q4 qf{filt[6], filt[7], filt[8], filt[9]};
  const V3 da = quat2axis_fast(qmul(qinverse(q), qf));  // boxMinus(filter, lin), KF:84-94
  for (int k = 0; k < 3; ++k) {
    ic.d[0 + k] = filt[0 + k] - ic.lin[0 + k];
    ic.d[3 + k] = filt[3 + k] - ic.lin[3 + k];
    ic.d[9 + k] = filt[10 + k] - ic.lin[10 + k];
    ic.d[12 + k] = filt[13 + k] - ic.lin[13 + k];
    ic.d[15 + k] = filt[16 + k] - ic.lin[16 + k];
  }
  ic.d[6] = da.x, ic.d[7] = da.y, ic.d[8] = da.z;
}

// ---------------------------------------------------------------------------
// rows -> 28 sums inside a wave without LDS: every lane forms the 28 products of its own row
// (21 of H^T H, 6 of H^T r, r^T r; all zero for an unused lane), then five halving butterflies
// (xor 32, 16, 8, 4, 2) in which a lane pair splits the sums it still carries — one keeps the lower
// half, the other the upper half, each adding the partner's copy — and a final xor-1 add.  29
// shuffles instead of 28 x 6; the tree is fixed, so the sums are bit-reproducible from run to run.
// Afterwards lane l holds sum number reduce_sum_index(l) of the whole wave.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  const int lo = __shfl_xor(__double2loint(v), mask), hi = __shfl_xor(__double2hiint(v), mask);
  return __hiloint2double(hi, lo);
}
// The same exchange (lane <-> lane ^ MASK, MASK a power of two, every lane of the wave active) on the VALU's own
// cross-lane paths instead of the LDS crossbar (ds_bpermute: an address register and ~100 clocks per dword):
// quad permutes for 1 and 2, two bank-masked row shifts for 4, a row rotation for 8, the gfx950 lane-swap
// instructions for 16 and 32.
template <int MASK>
__device__ __forceinline__ int xor_lane_i32(int v, int lane) {
  if constexpr (MASK == 1) {
    return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
  } else if constexpr (MASK == 2) {
    return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);  // quad_perm [2, 3, 0, 1]
  } else if constexpr (MASK == 4) {
    const int r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);  // row_shl:4 into lanes 0-3, 8-11 of a row
    return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);         // row_shr:4 into lanes 4-7, 12-15
  } else if constexpr (MASK == 8) {
    return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);  // row_ror:8
  } else if constexpr (MASK == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane & 16) ? r[0] : r[1]);
  } else {
    static_assert(MASK == 32, "power of two below the wave size");
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane & 32) ? r[0] : r[1]);
  }
}
template <int MASK>
__device__ __forceinline__ double xor_lane_f64(double v, int lane) {
  return __hiloint2double(xor_lane_i32<MASK>(__double2hiint(v), lane), xor_lane_i32<MASK>(__double2loint(v), lane));
}
__device__ __forceinline__ int reduce_sum_index(int lane) {
  const int local = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
  return (local < 7 && !(lane & 1)) ? ((lane >> 5) & 1) * 14 + ((lane >> 4) & 1) * 7 + local : -1;
}
__device__ __forceinline__ double wave_reduce_rows(const double (&row)[7], int lane) {
  constexpr int A[28] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6};
  constexpr int B[28] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6};
  double v[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) v[k] = row[A[k]] * row[B[k]];
  {
    const bool up = (lane & 32) != 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const double lo = v[i], hi = v[i + 14];
      v[i] = (up ? hi : lo) + xor_lane_f64<32>(up ? lo : hi, lane);
    }
  }
  {
    const bool up = (lane & 16) != 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const double lo = v[i], hi = v[i + 7];
      v[i] = (up ? hi : lo) + xor_lane_f64<16>(up ? lo : hi, lane);
    }
  }
  v[7] = 0.0;
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double lo = v[i], hi = v[i + 4];
      v[i] = (up ? hi : lo) + xor_lane_f64<8>(up ? lo : hi, lane);
    }
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double lo = v[i], hi = v[i + 2];
      v[i] = (up ? hi : lo) + xor_lane_f64<4>(up ? lo : hi, lane);
    }
  }
  {
    const bool up = (lane & 2) != 0;
    const double lo = v[0], hi = v[1];
    v[0] = (up ? hi : lo) + xor_lane_f64<2>(up ? lo : hi, lane);
  }
  return v[0] + xor_lane_f64<1>(v[0], lane);
}

}  // namespace lins
