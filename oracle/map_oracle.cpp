// map_oracle.cpp — CPU oracle of the scan-to-map row.  TEST INFRASTRUCTURE ONLY:
// nothing in the product may include, link or call it.  Pinned since round 3 to the reference's own text as far as
// that text goes: oracle/ref_map_driver.cpp compiles /root/reference/lins/src/lidar_mapping_node.cpp verbatim against
// stand-in ROS / PCL / GTSAM / OpenCV headers and tests/test_ref.py holds this file against its
// scan2MapOptimization (correspondences, selected rows, rounds, transform).  The OpenCV numerics themselves (below)
// are restated, not pinned — OpenCV is not on this machine.  A plain restatement of
//   pointAssociateToMap   LM:579-607      cornerOptimization LM:1351-1453
//   surfOptimization      LM:1455-1521    LMOptimization     LM:1523-1633
//   scan2MapOptimization  LM:1635-1652
// in the reference's f32 arithmetic, expression by expression.  Its third-party calls are restated as:
//   kdtree nearestKSearch(5)  exhaustive search, 5 smallest ((dx^2 + dy^2) + dz^2, index)
//   cv::eigen (symmetric)     cyclic Jacobi in f32, eigenvalues descending, eigenvectors as rows
//   cv::solve(DECOMP_QR)      Householder QR in f32 (least squares for the 5x3 plane fit)
//   cv::Mat::inv()            Gauss-Jordan with partial pivoting in f32
//   matAt * matA, matAt * matB   f64 accumulation in row order, rounded to f32 (the order of OpenCV's
//                             blocked f32 GEMM is unknowable; the device uses an f64 tree)
// The product's versions of these routines (csrc/map_math.h) are written separately to the same
// operation sequences.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/lins_map.h"
#include "ref_shim/lins_ref_shim/cv_restated.h"

namespace {

struct P3 {
  float x, y, z;
};

struct Assoc {  // updatePointAssociateToMapSinCos (LM:579-592)
  float cRoll, sRoll, cPitch, sPitch, cYaw, sYaw, tX, tY, tZ;
};
Assoc make_assoc(const float* t) {
  Assoc a;
  a.cRoll = std::cos(t[0]), a.sRoll = std::sin(t[0]);
  a.cPitch = std::cos(t[1]), a.sPitch = std::sin(t[1]);
  a.cYaw = std::cos(t[2]), a.sYaw = std::sin(t[2]);
  a.tX = t[3], a.tY = t[4], a.tZ = t[5];
  return a;
}
P3 associate(const Assoc& a, const lins_point& pi) {  // pointAssociateToMap (LM:594-607)
  float x1 = a.cYaw * pi.x - a.sYaw * pi.y;
  float y1 = a.sYaw * pi.x + a.cYaw * pi.y;
  float z1 = pi.z;
  float x2 = x1;
  float y2 = a.cRoll * y1 - a.sRoll * z1;
  float z2 = a.sRoll * y1 + a.cRoll * z1;
  P3 po;
  po.x = a.cPitch * x2 + a.sPitch * z2 + a.tX;
  po.y = y2 + a.tY;
  po.z = -a.sPitch * x2 + a.cPitch * z2 + a.tZ;
  return po;
}

// 5 nearest by (squared distance, index); sq[4] = inf and ind = -1 when the cloud has fewer than 5 points
void knn5(const lins_point* pts, int n, P3 q, int* ind, float* sq) {
  for (int k = 0; k < 5; ++k) ind[k] = -1, sq[k] = INFINITY;
  for (int i = 0; i < n; ++i) {
    float dx = q.x - pts[i].x, dy = q.y - pts[i].y, dz = q.z - pts[i].z;
    float d = (dx * dx + dy * dy) + dz * dz;
    if (!(d < sq[4])) continue;  // ascending index: a later equal distance never displaces an earlier one
    int k = 4;
    while (k > 0 && d < sq[k - 1]) sq[k] = sq[k - 1], ind[k] = ind[k - 1], --k;
    sq[k] = d, ind[k] = i;
  }
}

// The OpenCV routines of the path (cv::eigen, cv::solve(DECOMP_QR), Mat::inv, Mat * Mat) are restated once, in
// ref_shim/lins_ref_shim/cv_restated.h — the stand-in OpenCV header through which the reference's own text runs
// (oracle/_ref) uses the same restatement, so that this file and the reference can only differ in the reference's own glue.
template <int N>
void jacobi_eig(float* a, float* w, float* V) {
  lins_cvr::jacobi_eig(a, N, w, V);
}
template <int M, int N>
void qr_solve(float* a, float* b, float* x) {
  lins_cvr::qr_solve(a, M, N, b, x);
}
void inv6(const float* A, float* inv) { lins_cvr::inv(A, 6, inv); }

// cornerOptimization's body for one point (LM:1354-1452)
void corner_one(const Assoc& as, const lins_point* map, int n_map, const lins_point& ori, lins_map_corr& out) {
  const P3 sel = associate(as, ori);
  out.sel[0] = sel.x, out.sel[1] = sel.y, out.sel[2] = sel.z;
  float sq[5];
  knn5(map, n_map, sel, out.ind, sq);
  out.sq5 = sq[4];
  out.accepted = 0;
  out.coeff[0] = out.coeff[1] = out.coeff[2] = out.coeff[3] = 0.f;
  if (!(sq[4] < 1.0)) {  // fewer than 5 map points within 1 m: nothing is used of the search (LM:1360)
    for (int k = 0; k < 5; ++k) out.ind[k] = -1;
    out.sq5 = INFINITY;
    return;
  }
  float cx = 0, cy = 0, cz = 0;
  for (int j = 0; j < 5; j++) cx += map[out.ind[j]].x, cy += map[out.ind[j]].y, cz += map[out.ind[j]].z;
  cx /= 5, cy /= 5, cz /= 5;
  float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
  for (int j = 0; j < 5; j++) {
    float ax = map[out.ind[j]].x - cx, ay = map[out.ind[j]].y - cy, az = map[out.ind[j]].z - cz;
    a11 += ax * ax, a12 += ax * ay, a13 += ax * az, a22 += ay * ay, a23 += ay * az, a33 += az * az;
  }
  a11 /= 5, a12 /= 5, a13 /= 5, a22 /= 5, a23 /= 5, a33 /= 5;
  float A[9] = {a11, a12, a13, a12, a22, a23, a13, a23, a33}, D[3], V[9];
  jacobi_eig<3>(A, D, V);
  if (!(D[0] > 3 * D[1])) return;
  float x0 = sel.x, y0 = sel.y, z0 = sel.z;
  float x1 = cx + 0.1 * V[0], y1 = cy + 0.1 * V[1], z1 = cz + 0.1 * V[2];
  float x2 = cx - 0.1 * V[0], y2 = cy - 0.1 * V[1], z2 = cz - 0.1 * V[2];
  float a012 = std::sqrt(((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                         ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                         ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)));
  float l12 = std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
  float la = ((y1 - y2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
              (z1 - z2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))) /
             a012 / l12;
  float lb = -((x1 - x2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) -
               (z1 - z2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) /
             a012 / l12;
  float lc = -((x1 - x2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
               (y1 - y2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) /
             a012 / l12;
  float ld2 = a012 / l12;
  float s = 1 - 0.9 * std::fabs(ld2);
  out.coeff[0] = s * la, out.coeff[1] = s * lb, out.coeff[2] = s * lc, out.coeff[3] = s * ld2;
  if (s > 0.1) out.accepted = 1;
}

// surfOptimization's body for one point (LM:1458-1519)
void surf_one(const Assoc& as, const lins_point* map, int n_map, const lins_point& ori, lins_map_corr& out) {
  const P3 sel = associate(as, ori);
  out.sel[0] = sel.x, out.sel[1] = sel.y, out.sel[2] = sel.z;
  float sq[5];
  knn5(map, n_map, sel, out.ind, sq);
  out.sq5 = sq[4];
  out.accepted = 0;
  out.coeff[0] = out.coeff[1] = out.coeff[2] = out.coeff[3] = 0.f;
  if (!(sq[4] < 1.0)) {
    for (int k = 0; k < 5; ++k) out.ind[k] = -1;
    out.sq5 = INFINITY;
    return;
  }
  float A[15], B[5] = {-1, -1, -1, -1, -1}, X[3];
  for (int j = 0; j < 5; j++) A[j * 3 + 0] = map[out.ind[j]].x, A[j * 3 + 1] = map[out.ind[j]].y, A[j * 3 + 2] = map[out.ind[j]].z;
  qr_solve<5, 3>(A, B, X);
  float pa = X[0], pb = X[1], pc = X[2], pd = 1;
  float ps = std::sqrt(pa * pa + pb * pb + pc * pc);
  pa /= ps, pb /= ps, pc /= ps, pd /= ps;
  bool planeValid = true;
  for (int j = 0; j < 5; j++)
    if (std::fabs(pa * map[out.ind[j]].x + pb * map[out.ind[j]].y + pc * map[out.ind[j]].z + pd) > 0.2) {
      planeValid = false;
      break;
    }
  if (!planeValid) return;
  float pd2 = pa * sel.x + pb * sel.y + pc * sel.z + pd;
  float s = 1 - 0.9 * std::fabs(pd2) / std::sqrt(std::sqrt(sel.x * sel.x + sel.y * sel.y + sel.z * sel.z));
  out.coeff[0] = s * pa, out.coeff[1] = s * pb, out.coeff[2] = s * pc, out.coeff[3] = s * pd2;
  if (s > 0.1) out.accepted = 1;
}

struct LmState {
  bool degenerate = false;
  float P[36];
};

// LMOptimization (LM:1523-1633) on the selected rows; true = converged
bool lm_step(float* T, const std::vector<lins_point>& ori, const std::vector<lins_map_corr>& sel, int iter, LmState& st) {
  float srx = std::sin(T[0]), crx = std::cos(T[0]), sry = std::sin(T[1]), cry = std::cos(T[1]);
  float srz = std::sin(T[2]), crz = std::cos(T[2]);
  const int n = (int)ori.size();
  if (n < 50) return false;
  double AtA[36] = {0}, AtB[6] = {0};
  for (int i = 0; i < n; i++) {
    const lins_point& pointOri = ori[i];
    const float cx = sel[i].coeff[0], cy = sel[i].coeff[1], cz = sel[i].coeff[2], ci = sel[i].coeff[3];
    float arx = (crx * sry * srz * pointOri.x + crx * crz * sry * pointOri.y - srx * sry * pointOri.z) * cx +
                (-srx * srz * pointOri.x - crz * srx * pointOri.y - crx * pointOri.z) * cy +
                (crx * cry * srz * pointOri.x + crx * cry * crz * pointOri.y - cry * srx * pointOri.z) * cz;
    float ary = ((cry * srx * srz - crz * sry) * pointOri.x + (sry * srz + cry * crz * srx) * pointOri.y +
                 crx * cry * pointOri.z) * cx +
                ((-cry * crz - srx * sry * srz) * pointOri.x + (cry * srz - crz * srx * sry) * pointOri.y -
                 crx * sry * pointOri.z) * cz;
    float arz = ((crz * srx * sry - cry * srz) * pointOri.x + (-cry * crz - srx * sry * srz) * pointOri.y) * cx +
                (crx * crz * pointOri.x - crx * srz * pointOri.y) * cy +
                ((sry * srz + cry * crz * srx) * pointOri.x + (crz * sry - cry * srx * srz) * pointOri.y) * cz;
    const float row[6] = {arx, ary, arz, cx, cy, cz};
    const float b = -ci;
    for (int a = 0; a < 6; ++a) {
      for (int e = 0; e < 6; ++e) AtA[a * 6 + e] += (double)row[a] * (double)row[e];
      AtB[a] += (double)row[a] * (double)b;
    }
  }
  float A[36], B[6], X[6], Aq[36], Bq[6];
  for (int i = 0; i < 36; ++i) A[i] = (float)AtA[i];
  for (int i = 0; i < 6; ++i) B[i] = (float)AtB[i];
  std::memcpy(Aq, A, sizeof A), std::memcpy(Bq, B, sizeof B);
  qr_solve<6, 6>(Aq, Bq, X);
  if (iter == 0) {
    float Ae[36], E[6], V[36], V2[36], Vi[36];
    std::memcpy(Ae, A, sizeof A);
    jacobi_eig<6>(Ae, E, V);
    std::memcpy(V2, V, sizeof V);
    st.degenerate = false;
    for (int i = 5; i >= 0; i--) {
      if (E[i] < 100) {
        for (int j = 0; j < 6; j++) V2[i * 6 + j] = 0;
        st.degenerate = true;
      } else {
        break;
      }
    }
    inv6(V, Vi);
    lins_cvr::matmul(Vi, 6, 6, V2, 6, st.P);  // matP = matV.inv() * matV2
  }
  if (st.degenerate) {
    float X2[6];
    std::memcpy(X2, X, sizeof X);
    lins_cvr::matmul(st.P, 6, 6, X2, 1, X);  // matX = matP * matX2
  }
  for (int i = 0; i < 6; ++i) T[i] += X[i];
  auto rad2deg = [](float a) { return (float)(a * 57.29578f); };
  float deltaR = std::sqrt(std::pow(rad2deg(X[0]), 2) + std::pow(rad2deg(X[1]), 2) + std::pow(rad2deg(X[2]), 2));
  float deltaT = std::sqrt(std::pow(X[3] * 100, 2) + std::pow(X[4] * 100, 2) + std::pow(X[5] * 100, 2));
  return deltaR < 0.05 && deltaT < 0.05;
}

}  // namespace

extern "C" {

int oracle_map_correspondences(const lins_map_problem* in, lins_map_corr* corner, lins_map_corr* surf) {
  if (!in) return LINS_E_ARG;
  const Assoc as = make_assoc(in->transform);
  for (int i = 0; i < in->n_scan_corner; ++i) corner_one(as, in->map_corner, in->n_map_corner, in->scan_corner[i], corner[i]);
  for (int i = 0; i < in->n_scan_surf; ++i) surf_one(as, in->map_surf, in->n_map_surf, in->scan_surf[i], surf[i]);
  return LINS_OK;
}

int oracle_scan2map(const lins_map_problem* in, lins_map_result* out) {
  if (!in || !out) return LINS_E_ARG;
  std::memcpy(out->transform, in->transform, sizeof out->transform);
  out->iters = 0, out->converged = 0, out->degenerate = 0, out->n_sel = 0;
  if (!(in->n_map_corner > 10 && in->n_map_surf > 100)) return LINS_OK;  // LM:1636
  lins_map_problem p = *in;
  LmState st;
  std::vector<lins_map_corr> c(in->n_scan_corner), s(in->n_scan_surf);
  for (int iter = 0; iter < 10; ++iter) {
    std::memcpy(p.transform, out->transform, sizeof p.transform);
    oracle_map_correspondences(&p, c.data(), s.data());
    std::vector<lins_point> ori;
    std::vector<lins_map_corr> sel;
    for (int i = 0; i < in->n_scan_corner; ++i)
      if (c[i].accepted) ori.push_back(in->scan_corner[i]), sel.push_back(c[i]);
    for (int i = 0; i < in->n_scan_surf; ++i)
      if (s[i].accepted) ori.push_back(in->scan_surf[i]), sel.push_back(s[i]);
    out->n_sel = (int)ori.size();
    out->iters = iter + 1;
    if (lm_step(out->transform, ori, sel, iter, st)) {
      out->converged = 1;
      break;
    }
  }
  out->degenerate = st.degenerate ? 1 : 0;
  return LINS_OK;
}

}  // extern "C"
