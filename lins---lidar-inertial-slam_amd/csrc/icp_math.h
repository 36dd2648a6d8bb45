// icp_math.h — the 6-DoF Gauss-Newton step of the ICP fallback (calculateTransformation,
// StateEstimator.hpp:1198-1320) as host/device code: the device tail of the ICP kernel
// (ieskf_lds_impl.h, ICP = true) and the host path for clouds the LDS kernels cannot take
// (host/ieskf_host.cpp) run the very same arithmetic.
//   qr_solve6     x = JtJ^-1 Jtb, column-pivoted Householder QR with Eigen's rank rule (SE:1264)
//   eig_sym6      cyclic Jacobi eigen-decomposition of JtJ (SE:1274)
//   icp_gn_solve  the step x, with the degenerate directions projected out on iteration 0 (SE:1269-1302)
//   icp_apply     q <- q * rpy2Quat(x_R), t += x_T, stop rule 0.1 deg / 0.1 cm (SE:1305-1317)
// Every array these routines index with a run-time subscript lives in a caller-provided workspace `ws` of
// kIcpWorkspace doubles: on the device that is LDS (round 2 kept them in private arrays — 1.7 KB of scratch per lane
// and 124 spilled registers in the ICP kernel), on the host a local array.  Same arithmetic either way.
#pragma once
#include "lins_math.h"

#ifdef __HIPCC__
#define LINS_ICP_FN __host__ __device__ inline
// The 6 x 6 factorizations run on ONE lane a handful of times per divergence: kept as loops (fully unrolled, the
// straight-line code wants > 250 live registers and the ICP kernel spilled 120 of its 128).
#define LINS_ICP_NO_UNROLL _Pragma("unroll 1")
#else
#define LINS_ICP_FN inline
#define LINS_ICP_NO_UNROLL
#endif

namespace lins {

constexpr int kIcpWorkspace = 36 * 5 + 24;  // [0, 78): scratch of qr_solve6 / eig_sym6; then V, V2, Vc (36 each), w, x2

template <class T>
LINS_ICP_FN void icp_swap(T& a, T& b) {
  T t = a;
  a = b;
  b = t;
}

// x = A^-1 b by Householder QR with column pivoting; columns whose pivot falls
// below eps * n * max|pivot| are treated as rank-deficient (solution component 0),
// the rule Eigen's ColPivHouseholderQR::solve applies (SE:1264).
LINS_ICP_FN void qr_solve6(const double* A, const double* b_in, double* x, double* ws) {
  const int n = 6;
  double *a = ws, *b = ws + 36, *diag = ws + 42, *v = ws + 48, *y = ws + 54, *perm = ws + 60;  // (perm: small integers as doubles)
  LINS_ICP_NO_UNROLL
  for (int i = 0; i < 36; ++i) a[i] = A[i];
  LINS_ICP_NO_UNROLL
  for (int i = 0; i < 6; ++i) b[i] = b_in[i], perm[i] = i;
  double maxpiv = 0;
  LINS_ICP_NO_UNROLL
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = -1;
    LINS_ICP_NO_UNROLL
    for (int j = k; j < n; ++j) {
      double s = 0;
      LINS_ICP_NO_UNROLL
      for (int i = k; i < n; ++i) s += a[i * n + j] * a[i * n + j];
      if (s > best) best = s, p = j;
    }
    if (p != k) {
      LINS_ICP_NO_UNROLL
      for (int i = 0; i < n; ++i) icp_swap(a[i * n + k], a[i * n + p]);
      icp_swap(perm[k], perm[p]);
    }
    double nrm = sqrt(best > 0 ? best : 0);
    if (nrm > 0) {
      double alpha = a[k * n + k] >= 0 ? -nrm : nrm;
      LINS_ICP_NO_UNROLL
      for (int i = 0; i < n; ++i) v[i] = i >= k ? a[i * n + k] : 0.0;
      v[k] -= alpha;
      double vv = 0;
      LINS_ICP_NO_UNROLL
      for (int i = k; i < n; ++i) vv += v[i] * v[i];
      if (vv > 0) {
        LINS_ICP_NO_UNROLL
        for (int j = k; j < n; ++j) {
          double s = 0;
          LINS_ICP_NO_UNROLL
          for (int i = k; i < n; ++i) s += v[i] * a[i * n + j];
          s = 2 * s / vv;
          LINS_ICP_NO_UNROLL
          for (int i = k; i < n; ++i) a[i * n + j] -= s * v[i];
        }
        double s = 0;
        LINS_ICP_NO_UNROLL
        for (int i = k; i < n; ++i) s += v[i] * b[i];
        s = 2 * s / vv;
        LINS_ICP_NO_UNROLL
        for (int i = k; i < n; ++i) b[i] -= s * v[i];
      }
    }
    diag[k] = a[k * n + k];
    maxpiv = maxpiv > fabs(diag[k]) ? maxpiv : fabs(diag[k]);
  }
  int rank = 0;
  LINS_ICP_NO_UNROLL
  for (int k = 0; k < n; ++k)
    if (fabs(diag[k]) > maxpiv * 2.220446049250313e-16 * n) ++rank;
  LINS_ICP_NO_UNROLL
  for (int i = 0; i < n; ++i) y[i] = 0;
  LINS_ICP_NO_UNROLL
  for (int i = rank - 1; i >= 0; --i) {
    double s = b[i];
    LINS_ICP_NO_UNROLL
    for (int j = i + 1; j < rank; ++j) s -= a[i * n + j] * y[j];
    y[i] = s / a[i * n + i];
  }
  LINS_ICP_NO_UNROLL
  for (int i = 0; i < n; ++i) x[(int)perm[i]] = y[i];
}

// symmetric 6x6 eigen-decomposition (cyclic Jacobi); ascending eigenvalues,
// eigenvectors in columns, sign: largest-magnitude component positive.
LINS_ICP_FN void eig_sym6(const double* A, double* w, double* V, double* ws) {
  const int n = 6;
  double *a = ws, *Vs = ws + 36, *ord = ws + 72;  // (ord: small integers as doubles)
  LINS_ICP_NO_UNROLL
  for (int i = 0; i < 36; ++i) a[i] = A[i];
  LINS_ICP_NO_UNROLL
  for (int i = 0; i < n; ++i)
    LINS_ICP_NO_UNROLL
    for (int j = 0; j < n; ++j) V[i * n + j] = i == j;
  LINS_ICP_NO_UNROLL
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0;
    LINS_ICP_NO_UNROLL
    for (int i = 0; i < n; ++i)
      LINS_ICP_NO_UNROLL
      for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
    if (off < 1e-300) break;
    LINS_ICP_NO_UNROLL
    for (int p = 0; p < n; ++p)
      LINS_ICP_NO_UNROLL
      for (int q = p + 1; q < n; ++q) {
        double apq = a[p * n + q];
        if (apq == 0) continue;
        double th = (a[q * n + q] - a[p * n + p]) / (2 * apq);
        double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
        double c = 1 / sqrt(t * t + 1), s = t * c;
        LINS_ICP_NO_UNROLL
        for (int k = 0; k < n; ++k) {
          double x = a[k * n + p], y = a[k * n + q];
          a[k * n + p] = c * x - s * y, a[k * n + q] = s * x + c * y;
        }
        LINS_ICP_NO_UNROLL
        for (int k = 0; k < n; ++k) {
          double x = a[p * n + k], y = a[q * n + k];
          a[p * n + k] = c * x - s * y, a[q * n + k] = s * x + c * y;
        }
        LINS_ICP_NO_UNROLL
        for (int k = 0; k < n; ++k) {
          double x = V[k * n + p], y = V[k * n + q];
          V[k * n + p] = c * x - s * y, V[k * n + q] = s * x + c * y;
        }
      }
  }
  LINS_ICP_NO_UNROLL
  for (int i = 0; i < n; ++i) ord[i] = i;
  auto ev = [&](int k) { const int o = (int)ord[k]; return a[o * n + o]; };
  LINS_ICP_NO_UNROLL
  for (int i = 1; i < n; ++i)  // insertion sort by eigenvalue, ascending (stable)
    LINS_ICP_NO_UNROLL
    for (int j = i; j > 0 && ev(j) < ev(j - 1); --j) icp_swap(ord[j], ord[j - 1]);
  LINS_ICP_NO_UNROLL
  for (int j = 0; j < n; ++j) {
    const int oj = (int)ord[j];
    w[j] = a[oj * n + oj];
    int big = 0;
    LINS_ICP_NO_UNROLL
    for (int i = 1; i < n; ++i)
      if (fabs(V[i * n + oj]) > fabs(V[big * n + oj])) big = i;
    double sg = V[big * n + oj] < 0 ? -1.0 : 1.0;
    LINS_ICP_NO_UNROLL
    for (int i = 0; i < n; ++i) Vs[i * n + j] = sg * V[i * n + oj];
  }
  LINS_ICP_NO_UNROLL
  for (int i = 0; i < 36; ++i) V[i] = Vs[i];
}

LINS_ICP_FN void gauss_solve6(double* a, double* b, int m) {  // A X = B, partial pivoting
  const int n = 6;
  LINS_ICP_NO_UNROLL
  for (int k = 0; k < n; ++k) {
    int p = k;
    LINS_ICP_NO_UNROLL
    for (int i = k + 1; i < n; ++i)
      if (fabs(a[i * n + k]) > fabs(a[p * n + k])) p = i;
    if (p != k) {
      LINS_ICP_NO_UNROLL
      for (int j = 0; j < n; ++j) icp_swap(a[k * n + j], a[p * n + j]);
      LINS_ICP_NO_UNROLL
      for (int j = 0; j < m; ++j) icp_swap(b[k * m + j], b[p * m + j]);
    }
    LINS_ICP_NO_UNROLL
    for (int i = k + 1; i < n; ++i) {
      double f = a[i * n + k] / a[k * n + k];
      LINS_ICP_NO_UNROLL
      for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
      LINS_ICP_NO_UNROLL
      for (int j = 0; j < m; ++j) b[i * m + j] -= f * b[k * m + j];
    }
  }
  LINS_ICP_NO_UNROLL
  for (int i = n - 1; i >= 0; --i)
    LINS_ICP_NO_UNROLL
    for (int j = 0; j < m; ++j) {
      double s = b[i * m + j];
      LINS_ICP_NO_UNROLL
      for (int k = i + 1; k < n; ++k) s -= a[i * n + k] * b[k * m + j];
      b[i * m + j] = s / a[i * n + i];
    }
}


// the Gauss-Newton step (rotation part first: O_R = 0, O_P = 3, parameters.h:162-163)
LINS_ICP_FN void icp_gn_solve(const double* JTJ, const double* JTb, int iter, double* x, double* ws) {
  qr_solve6(JTJ, JTb, x, ws);
  if (iter == 0) {  // degeneracy projection (SE:1269-1302)
    double *V = ws + 36 * 2 + 6, *V2 = ws + 36 * 3 + 6, *Vc = ws + 36 * 4 + 6, *w = ws + 36 * 5 + 6, *x2 = ws + 36 * 5 + 12;
    eig_sym6(JTJ, w, V, ws);  // (its own scratch: ws[0 .. 78))
    LINS_ICP_NO_UNROLL
    for (int i = 0; i < 36; ++i) V2[i] = V[i];
    bool degenerate = false;
    LINS_ICP_NO_UNROLL
    for (int i = 0; i < 6; ++i) {
      if (w[i] < 10.) {
        LINS_ICP_NO_UNROLL
        for (int j = 0; j < 6; ++j) V2[i * 6 + j] = 0;  // the reference zeroes row i
        degenerate = true;
      } else {
        break;
      }
    }
    if (degenerate) {
      LINS_ICP_NO_UNROLL
      for (int i = 0; i < 36; ++i) Vc[i] = V[i];
      gauss_solve6(Vc, V2, 6);  // matP = matV^-1 matV2
      LINS_ICP_NO_UNROLL
      for (int i = 0; i < 6; ++i) {
        double s = 0;
        LINS_ICP_NO_UNROLL
        for (int k = 0; k < 6; ++k) s += V2[i * 6 + k] * x[k];
        x2[i] = s;
      }
      LINS_ICP_NO_UNROLL
      for (int i = 0; i < 6; ++i) x[i] = x2[i];
    }
  }
}

// true = converged
LINS_ICP_FN bool icp_apply(const double* x, double* t, Q4& q) {
  q = qnormalized(qmul(q, rpy2quat(V3{x[0], x[1], x[2]})));
  t[0] += x[3], t[1] += x[4], t[2] += x[5];
  const double r2d = 180.0 / 3.14159265358979323846;
  const double dR = sqrt((x[0] * r2d) * (x[0] * r2d) + (x[1] * r2d) * (x[1] * r2d) + (x[2] * r2d) * (x[2] * r2d));
  const double dT = sqrt((100 * x[3]) * (100 * x[3]) + (100 * x[4]) * (100 * x[4]) + (100 * x[5]) * (100 * x[5]));
  return dR < 0.1 && dT < 0.1;
}

// one row of the Gauss-Newton system from an accepted correspondence: J = [c^T (-R(s phi) [p]x), c^T],
// b = -0.05 res (SE:1246-1257); p = the raw keypoint, s its relative time
LINS_ICP_FN void icp_row(double inv_period, V3 phi, float px, float py, float pz, float intensity, const float* coeff,
                         double* J, double& b) {
  const float frac = intensity - (float)(int)intensity;
  const double s = inv_period * (double)frac;
  const M3 R = qmat(axis2quat(s * phi));
  M3 negR;
  for (int k = 0; k < 9; ++k) negR.m[k] = -R.m[k];
  const V3 cf{coeff[0], coeff[1], coeff[2]};
  const V3 jr = rowmul(cf, mmul(negR, skew(V3{px, py, pz})));
  J[0] = jr.x, J[1] = jr.y, J[2] = jr.z, J[3] = cf.x, J[4] = cf.y, J[5] = cf.z;
  b = -0.05 * (double)coeff[3];
}

}  // namespace lins
