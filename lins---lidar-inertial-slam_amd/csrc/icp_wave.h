// icp_wave.h — the Gauss-Newton tail of the ICP fallback (icp_math.h: qr_solve6, eig_sym6, icp_gn_solve) spread over
// one wave: COLUMN j of a 6 x 6 matrix lives in lane j (six doubles in registers), the right-hand side in lane 6.
//
// icp_math.h is the definition (it also compiles for the host: the oracle-side shim runs it); this file performs the
// SAME floating-point operations in the SAME order per element — every sum runs over its index in the order the scalar
// loops run, pivots and rotation angles are formed once from values read across lanes — so the results are the
// scalar version's bits (tests/test_gpu_math.py holds the two against each other on random, rank-deficient and
// degenerate systems).  What changes is where the operands live: the scalar version, run by one lane, keeps its arrays
// in LDS (registers would be indexed at run time) and pays an LDS round trip per operand, ~45 us per Gauss-Newton round;
// here a column's Householder update or Jacobi rotation is local to its lane and only pivots, the Householder vector
// and the two columns of a rotation cross lanes (v_readlane).
#ifndef LINS_ICP_WAVE_H_
#define LINS_ICP_WAVE_H_

#include "icp_math.h"
#include "ieskf_rowsum.h"

namespace lins {

// x = A^-1 b as qr_solve6 computes it.  A row-major (any address space), every lane gets x.
__device__ __forceinline__ void wave_qr_solve6(const double* __restrict__ A, const double* __restrict__ b_in, int lane, double (&x)[6]) {
  double a[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) a[i] = lane < 6 ? A[i * 6 + lane] : (lane == 6 ? b_in[i] : 0.0);
  int perm = lane;
  double diag[6], maxpiv = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double s = 0;  // squared norm of this lane's column below row k, rows in ascending order
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i >= k) s += a[i] * a[i];
    int p = k;
    double best = -1;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j >= k) {
        const double sj = readlane_f64(s, j);
        if (sj > best) best = sj, p = j;
      }
    if (p != k) {  // (wave-uniform) exchange columns k and p
      const int src = lane == k ? p : (lane == p ? k : lane);
#pragma unroll
      for (int i = 0; i < 6; ++i) a[i] = shfl_f64(a[i], src);
      perm = __shfl(perm, src);
    }
    const double nrm = sqrt(best > 0 ? best : 0);
    if (nrm > 0) {
      double v[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = i >= k ? readlane_f64(a[i], k) : 0.0;
      const double alpha = v[k] >= 0 ? -nrm : nrm;
      v[k] -= alpha;
      double vv = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (i >= k) vv += v[i] * v[i];
      if (vv > 0 && lane >= k && lane <= 6) {  // the columns k .. 5 and the right-hand side
        double s2 = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
          if (i >= k) s2 += v[i] * a[i];
        s2 = 2 * s2 / vv;
#pragma unroll
        for (int i = 0; i < 6; ++i)
          if (i >= k) a[i] -= s2 * v[i];
      }
    }
    diag[k] = readlane_f64(a[k], k);
    maxpiv = maxpiv > fabs(diag[k]) ? maxpiv : fabs(diag[k]);
  }
  int rank = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (fabs(diag[k]) > maxpiv * 2.220446049250313e-16 * 6) ++rank;
  double y[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 5; i >= 0; --i)
    if (i < rank) {
      double s = readlane_f64(a[i], 6);
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j > i && j < rank) s -= readlane_f64(a[i], j) * y[j];
      y[i] = s / readlane_f64(a[i], i);
    }
#pragma unroll
  for (int r = 0; r < 6; ++r) x[r] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int pi = __builtin_amdgcn_readlane(perm, i);
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (pi == r) x[r] = y[i];
  }
}

// eig_sym6 over the wave: ascending eigenvalues w (every lane), eigenvector j = column j of V in lane j.
__device__ __forceinline__ void wave_eig_sym6(const double* __restrict__ A, int lane, double (&w)[6], double (&V)[6]) {
  double a[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) a[i] = lane < 6 ? A[i * 6 + lane] : 0.0, V[i] = i == lane ? 1.0 : 0.0;
#pragma unroll 1
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i + 1; j < 6; ++j) {
        const double e = readlane_f64(a[i], j);
        off += e * e;
      }
    if (off < 1e-300) break;
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int q = p + 1; q < 6; ++q) {
        const double apq = readlane_f64(a[p], q);
        if (apq == 0) continue;
        const double th = (readlane_f64(a[q], q) - readlane_f64(a[p], p)) / (2 * apq);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
        const double c = 1 / sqrt(t * t + 1), s = t * c;
#pragma unroll
        for (int k = 0; k < 6; ++k) {  // columns p and q of a
          const double xk = readlane_f64(a[k], p), yk = readlane_f64(a[k], q);
          if (lane == p) a[k] = c * xk - s * yk;
          if (lane == q) a[k] = s * xk + c * yk;
        }
        {  // rows p and q: local to every column
          const double xk = a[p], yk = a[q];
          a[p] = c * xk - s * yk, a[q] = s * xk + c * yk;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {  // columns p and q of V
          const double xk = readlane_f64(V[k], p), yk = readlane_f64(V[k], q);
          if (lane == p) V[k] = c * xk - s * yk;
          if (lane == q) V[k] = s * xk + c * yk;
        }
      }
  }
  // insertion sort by eigenvalue, ascending (stable) — the scalar loop with its indices made static
  double dv[6];
  int ord[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) dv[i] = readlane_f64(a[i], i), ord[i] = i;
#pragma unroll
  for (int i = 1; i < 6; ++i) {
    bool moving = true;
#pragma unroll
    for (int j = i; j > 0; --j) {
      moving = moving && dv[j] < dv[j - 1];
      if (moving) {
        const double td = dv[j];
        dv[j] = dv[j - 1], dv[j - 1] = td;
        const int to = ord[j];
        ord[j] = ord[j - 1], ord[j - 1] = to;
      }
    }
  }
  double nv[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    w[j] = dv[j];
    double col[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i] = readlane_f64(V[i], ord[j]);
    double bigv = col[0];
#pragma unroll
    for (int i = 1; i < 6; ++i)
      if (fabs(col[i]) > fabs(bigv)) bigv = col[i];
    const double sg = bigv < 0 ? -1.0 : 1.0;
    if (lane == j) {
#pragma unroll
      for (int i = 0; i < 6; ++i) nv[i] = sg * col[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) V[i] = nv[i];
}

// icp_gn_solve over the wave.  ws: kIcpWorkspace doubles in LDS, touched only when the first round finds a degenerate
// direction (the projection's 6 x 6 solve with partial pivoting stays the scalar routine, run by lane 0 — rare).
__device__ __forceinline__ void wave_icp_gn_solve(const double* __restrict__ JTJ, const double* __restrict__ JTb, int iter, int lane,
                                                  double (&x)[6], double* __restrict__ ws) {
  wave_qr_solve6(JTJ, JTb, lane, x);
  if (iter != 0) return;
  double w[6], V[6];
  wave_eig_sym6(JTJ, lane, w, V);
  bool degenerate = false, zero_row[6];
  {
    bool run = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      run = run && w[i] < 10.;
      zero_row[i] = run;
      degenerate = degenerate || run;
    }
  }
  if (!degenerate) return;  // (wave-uniform)
  double *Vc = ws, *V2 = ws + 36, *xs = ws + 72;
  if (lane < 6) {
#pragma unroll
    for (int i = 0; i < 6; ++i) Vc[i * 6 + lane] = V[i], V2[i * 6 + lane] = zero_row[i] ? 0.0 : V[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane == 0) {
    double* xin = ws + 78;
#pragma unroll
    for (int k = 0; k < 6; ++k) xin[k] = x[k];
    gauss_solve6(Vc, V2, 6);  // matP = matV^-1 matV2
    LINS_ICP_NO_UNROLL
    for (int i = 0; i < 6; ++i) {
      double s = 0;
      LINS_ICP_NO_UNROLL
      for (int k = 0; k < 6; ++k) s += V2[i * 6 + k] * xin[k];
      xs[i] = s;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = xs[i];
}

}  // namespace lins
#endif
