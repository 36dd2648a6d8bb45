// ieskf_host.cpp — performIESKF() as lins_fusion_node sees it: the GPU IESKF
// loop, and when the filter diverges the reference's fallback
// (StateEstimator.hpp:585-592): estimateTransform (SE:1163-1196) = up to
// NUM_ITER rounds of {GPU correspondence pass, host 6-DoF Gauss-Newton step
// calculateTransformation (SE:1198-1320)} from the filter's pose, covariance
// left un-updated.
//
// The GN step is 6x6 scalar algebra on <= a few hundred rows: host work.  The
// correspondence search + residual/Jacobian rows — the data-parallel part — come
// from the same HIP kernels as the IESKF loop (lins_correspondences()).

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../../include/lins_host.h"
#include "../lins_math.h"

using namespace lins;

namespace {

// x = A^-1 b by Householder QR with column pivoting; columns whose pivot falls
// below eps * n * max|pivot| are treated as rank-deficient (solution component 0),
// the rule Eigen's ColPivHouseholderQR::solve applies (SE:1264).
void qr_solve6(const double* A, const double* b_in, double* x) {
  const int n = 6;
  double a[36], b[6], diag[6];
  std::memcpy(a, A, sizeof a);
  std::memcpy(b, b_in, sizeof b);
  int perm[6] = {0, 1, 2, 3, 4, 5};
  double maxpiv = 0;
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = -1;
    for (int j = k; j < n; ++j) {
      double s = 0;
      for (int i = k; i < n; ++i) s += a[i * n + j] * a[i * n + j];
      if (s > best) best = s, p = j;
    }
    if (p != k) {
      for (int i = 0; i < n; ++i) std::swap(a[i * n + k], a[i * n + p]);
      std::swap(perm[k], perm[p]);
    }
    double nrm = std::sqrt(best > 0 ? best : 0);
    if (nrm > 0) {
      double alpha = a[k * n + k] >= 0 ? -nrm : nrm;
      double v[6] = {0};
      for (int i = k; i < n; ++i) v[i] = a[i * n + k];
      v[k] -= alpha;
      double vv = 0;
      for (int i = k; i < n; ++i) vv += v[i] * v[i];
      if (vv > 0) {
        for (int j = k; j < n; ++j) {
          double s = 0;
          for (int i = k; i < n; ++i) s += v[i] * a[i * n + j];
          s = 2 * s / vv;
          for (int i = k; i < n; ++i) a[i * n + j] -= s * v[i];
        }
        double s = 0;
        for (int i = k; i < n; ++i) s += v[i] * b[i];
        s = 2 * s / vv;
        for (int i = k; i < n; ++i) b[i] -= s * v[i];
      }
    }
    diag[k] = a[k * n + k];
    maxpiv = std::max(maxpiv, std::fabs(diag[k]));
  }
  int rank = 0;
  for (int k = 0; k < n; ++k)
    if (std::fabs(diag[k]) > maxpiv * 2.220446049250313e-16 * n) ++rank;
  double y[6] = {0};
  for (int i = rank - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < rank; ++j) s -= a[i * n + j] * y[j];
    y[i] = s / a[i * n + i];
  }
  for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
}

// symmetric 6x6 eigen-decomposition (cyclic Jacobi); ascending eigenvalues,
// eigenvectors in columns, sign: largest-magnitude component positive.
void eig_sym6(const double* A, double* w, double* V) {
  const int n = 6;
  double a[36];
  std::memcpy(a, A, sizeof a);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = i == j;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = a[p * n + q];
        if (apq == 0) continue;
        double th = (a[q * n + q] - a[p * n + p]) / (2 * apq);
        double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1));
        double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < n; ++k) {
          double x = a[k * n + p], y = a[k * n + q];
          a[k * n + p] = c * x - s * y, a[k * n + q] = s * x + c * y;
        }
        for (int k = 0; k < n; ++k) {
          double x = a[p * n + k], y = a[q * n + k];
          a[p * n + k] = c * x - s * y, a[q * n + k] = s * x + c * y;
        }
        for (int k = 0; k < n; ++k) {
          double x = V[k * n + p], y = V[k * n + q];
          V[k * n + p] = c * x - s * y, V[k * n + q] = s * x + c * y;
        }
      }
  }
  int ord[6] = {0, 1, 2, 3, 4, 5};
  std::sort(ord, ord + n, [&](int x, int y) { return a[x * n + x] < a[y * n + y]; });
  double Vs[36];
  for (int j = 0; j < n; ++j) {
    w[j] = a[ord[j] * n + ord[j]];
    int big = 0;
    for (int i = 1; i < n; ++i)
      if (std::fabs(V[i * n + ord[j]]) > std::fabs(V[big * n + ord[j]])) big = i;
    double sg = V[big * n + ord[j]] < 0 ? -1.0 : 1.0;
    for (int i = 0; i < n; ++i) Vs[i * n + j] = sg * V[i * n + ord[j]];
  }
  std::memcpy(V, Vs, sizeof Vs);
}

void gauss_solve6(double* a, double* b, int m) {  // A X = B, partial pivoting
  const int n = 6;
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(a[i * n + k]) > std::fabs(a[p * n + k])) p = i;
    if (p != k) {
      for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]);
      for (int j = 0; j < m; ++j) std::swap(b[k * m + j], b[p * m + j]);
    }
    for (int i = k + 1; i < n; ++i) {
      double f = a[i * n + k] / a[k * n + k];
      for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
      for (int j = 0; j < m; ++j) b[i * m + j] -= f * b[k * m + j];
    }
  }
  for (int i = n - 1; i >= 0; --i)
    for (int j = 0; j < m; ++j) {
      double s = b[i * m + j];
      for (int k = i + 1; k < n; ++k) s -= a[i * n + k] * b[k * m + j];
      b[i * m + j] = s / a[i * n + i];
    }
}

// calculateTransformation (SE:1198-1320); true = converged
bool gauss_newton_step(const lins_params& prm, double* t, Q4& q, const lins_scan_pair& in,
                       const std::vector<lins_corr>& cs, const std::vector<lins_corr>& cc, int iter) {
  double JTJ[36] = {0}, JTb[6] = {0};
  V3 phi = quat2axis(q);
  auto add_row = [&](const lins_point& kp, const lins_corr& c) {
    float frac = kp.intensity - (float)(int)kp.intensity;
    double s = (double)(1.f / prm.scan_period) * (double)frac;
    M3 R = qmat(axis2quat(s * phi));
    M3 negR;
    for (int k = 0; k < 9; ++k) negR.m[k] = -R.m[k];
    V3 cf{c.coeff[0], c.coeff[1], c.coeff[2]};
    V3 jr = rowmul(cf, mmul(negR, skew(V3{kp.x, kp.y, kp.z})));
    double J[6] = {jr.x, jr.y, jr.z, cf.x, cf.y, cf.z};  // O_R = 0, O_P = 3 (parameters.h:162-163)
    double b = -0.05 * (double)c.coeff[3];
    for (int a = 0; a < 6; ++a) {
      for (int e = 0; e < 6; ++e) JTJ[a * 6 + e] += J[a] * J[e];
      JTb[a] += J[a] * b;
    }
  };
  for (int i = 0; i < in.n_surf_flat; ++i)
    if (cs[i].accepted) add_row(in.surf_flat[i], cs[i]);
  for (int i = 0; i < in.n_corner_sharp; ++i)
    if (cc[i].accepted) add_row(in.corner_sharp[i], cc[i]);
  double x[6];
  qr_solve6(JTJ, JTb, x);
  if (iter == 0) {  // degeneracy projection (SE:1269-1302)
    double w[6], V[36], V2[36];
    eig_sym6(JTJ, w, V);
    std::memcpy(V2, V, sizeof V);
    bool degenerate = false;
    for (int i = 0; i < 6; ++i) {
      if (w[i] < 10.) {
        for (int j = 0; j < 6; ++j) V2[i * 6 + j] = 0;  // the reference zeroes row i
        degenerate = true;
      } else {
        break;
      }
    }
    if (degenerate) {
      double Vc[36];
      std::memcpy(Vc, V, sizeof V);
      gauss_solve6(Vc, V2, 6);  // matP = matV^-1 matV2
      double x2[6];
      for (int i = 0; i < 6; ++i) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += V2[i * 6 + k] * x[k];
        x2[i] = s;
      }
      std::memcpy(x, x2, sizeof x2);
    }
  }
  q = qnormalized(qmul(q, rpy2quat(V3{x[0], x[1], x[2]})));
  t[0] += x[3], t[1] += x[4], t[2] += x[5];
  const double r2d = 180.0 / M_PI;
  double dR = std::sqrt((x[0] * r2d) * (x[0] * r2d) + (x[1] * r2d) * (x[1] * r2d) + (x[2] * r2d) * (x[2] * r2d));
  double dT = std::sqrt((100 * x[3]) * (100 * x[3]) + (100 * x[4]) * (100 * x[4]) + (100 * x[5]) * (100 * x[5]));
  return dR < 0.1 && dT < 0.1;
}

}  // namespace

extern "C" int lins_host_perform_ieskf(lins_ctx* ctx, const lins_params* prm, const lins_scan_pair* in,
                                        lins_result* out, int32_t* used_icp_fallback) {
  if (!ctx || !prm || !in || !out) return LINS_E_ARG;
  if (used_icp_fallback) *used_icp_fallback = 0;
  int rc = lins_ieskf_update(ctx, in, out);
  if (rc != LINS_OK || !out->diverged) return rc;
  // ---- "======Using ICP Method======" (SE:585-592) --------------------------
  if (used_icp_fallback) *used_icp_fallback = 1;
  double lin[LINS_STATE_DIM];
  std::memcpy(lin, in->state, sizeof lin);
  double t[3] = {lin[0], lin[1], lin[2]};
  Q4 q{lin[6], lin[7], lin[8], lin[9]};
  std::vector<lins_corr> cs(std::max(in->n_surf_flat, 1)), cc(std::max(in->n_corner_sharp, 1));
  for (int iter = 0; iter < prm->num_iter; ++iter) {
    lin[0] = t[0], lin[1] = t[1], lin[2] = t[2];
    lin[6] = q.w, lin[7] = q.x, lin[8] = q.y, lin[9] = q.z;
    rc = lins_correspondences(ctx, in, lin, iter, cs.data(), cc.data());
    if (rc != LINS_OK) return rc;
    int ms = 0, mc = 0;
    for (int i = 0; i < in->n_surf_flat; ++i) ms += cs[i].accepted;
    for (int i = 0; i < in->n_corner_sharp; ++i) mc += cc[i].accepted;
    if (ms < 10) continue;  // SE:1175-1178
    if (mc < 5) continue;   // SE:1181-1184
    if (gauss_newton_step(*prm, t, q, *in, cs, cc, iter)) break;
  }
  std::memcpy(out->state, in->state, sizeof in->state);  // filterState with rn_, qbn_ replaced
  out->state[0] = t[0], out->state[1] = t[1], out->state[2] = t[2];
  out->state[6] = q.w, out->state[7] = q.x, out->state[8] = q.y, out->state[9] = q.z;
  std::memcpy(out->cov, in->cov, sizeof in->cov);  // Pk_ un-updated
  return LINS_OK;
}
