// lins_oracle.cpp — CPU oracle for the LINS IESKF update path.
//
// TEST INFRASTRUCTURE ONLY (see lins_oracle.h).  Pinned to the reference's own
// compiled text, oracle/_ref (tests/test_ref.py; lins_oracle.h says how).
//
// A from-scratch restatement (flat arrays, own linear algebra, no PCL/Eigen) of
// the behaviour of, relative to /root/reference/lins/include:
//   performIESKF                      StateEstimator.hpp:465-600
//   findCorrespondingSurfFeatures     StateEstimator.hpp:829-953
//   findCorrespondingCornerFeatures   StateEstimator.hpp:955-1063
//   transformToStart                  StateEstimator.hpp:1066-1080
//   estimateTransform / calculateTransformation   StateEstimator.hpp:1163-1320
//   GlobalState::boxPlus / boxMinus   KalmanFilter.hpp:71-94
//   axis2Quat / Quat2axis / wrap_pi / skew / Rinvleft / rpy2Quat
//                                     math_utils.h:27-88,131-148,196-204,304-321
// Third-party arithmetic that is not in the reference tree is restated from its
// published semantics: pcl::KdTreeFLANN 1-NN (exact, L2_Simple<float> order
// ((dx*dx+dy*dy)+dz*dz); ties -> lowest index, our rule), Eigen quaternion
// product / rotate / toRotationMatrix, dense LLT, column-pivoted Householder QR,
// self-adjoint eigen decomposition.
//
// Built with -O3 -ffp-contract=off and no -march flags: the reference is built
// -O3 for baseline x86-64 (lins/CMakeLists.txt:3), i.e. SSE2, no FMA.

#include "lins_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ----------------------------------------------------------------------------
// small fixed-size algebra
// ----------------------------------------------------------------------------
struct V3 {
  double x, y, z;
};
struct Q4 {
  double w, x, y, z;
};
struct M3 {
  double m[9];  // row-major
};

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

// Eigen::Quaternion product
inline Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Q4 qnormalized(Q4 q) {
  double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return {q.w / n, q.x / n, q.y / n, q.z / n};
}
inline Q4 qinverse(Q4 q) {
  double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
// Eigen::Quaternion::_transformVector : v + 2w(qv x v) + 2 qv x (qv x v)
inline V3 qrot(Q4 q, V3 v) {
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}
// Eigen::Quaternion::toRotationMatrix
inline M3 qmat(Q4 q) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 r;
  r.m[0] = 1 - (tyy + tzz);
  r.m[1] = txy - twz;
  r.m[2] = txz + twy;
  r.m[3] = txy + twz;
  r.m[4] = 1 - (txx + tzz);
  r.m[5] = tyz - twx;
  r.m[6] = txz - twy;
  r.m[7] = tyz + twx;
  r.m[8] = 1 - (txx + tyy);
  return r;
}
inline M3 skew(V3 q) { return {{0, -q.z, q.y, q.z, 0, -q.x, -q.y, q.x, 0}}; }
inline M3 mmul(const M3& a, const M3& b) {
  M3 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      c.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return c;
}
inline V3 rowmul(V3 r, const M3& a) {  // r^T * A
  return {r.x * a.m[0] + r.y * a.m[3] + r.z * a.m[6], r.x * a.m[1] + r.y * a.m[4] + r.z * a.m[7],
          r.x * a.m[2] + r.y * a.m[5] + r.z * a.m[8]};
}

// math_utils.h:27-37
inline double wrap_pi(double x) {
  while (x >= M_PI) x -= 2.0 * M_PI;
  while (x < -M_PI) x += 2.0 * M_PI;
  return x;
}
// math_utils.h:43-73 (the theta<1e-10 early-out of the 2-arg overload is dead
// code in the reference — it falls through — so only the 1-arg guard matters)
inline Q4 axis2quat(V3 v) {
  double theta = norm(v);
  if (theta < 1e-10) return {1, 0, 0, 0};
  V3 a = v / theta;
  double mag = std::sin(theta / 2.0);
  return {std::cos(theta / 2.0), a.x * mag, a.y * mag, a.z * mag};
}
// math_utils.h:75-88
inline V3 quat2axis(Q4 q) {
  double mag = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  V3 v{q.x, q.y, q.z};
  if (mag >= 1e-10) {
    v = v / mag;
    v = wrap_pi(2.0 * std::atan2(mag, q.w)) * v;
  }
  return v;
}
// math_utils.h:304-321
inline M3 rinvleft(V3 axis) {
  double theta = norm(axis);
  M3 r{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  if (theta < 1e-10) return r;
  double h = theta / 2.0;
  V3 a = axis / theta;
  double cot = std::cos(h) / std::sin(h);
  double s = h * cot;
  M3 k = skew(a);
  double av[3] = {a.x, a.y, a.z};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r.m[i * 3 + j] = (s * (i == j ? 1.0 : 0.0) + (1.0 - s) * av[i] * av[j]) - h * k.m[i * 3 + j];
  return r;
}
// math_utils.h:131-148 (the trailing Q.normalized() discards its result)
inline Q4 rpy2quat(V3 rpy) {
  double hy = rpy.z * 0.5, hp = rpy.y * 0.5, hr = rpy.x * 0.5;
  double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp);
  double cr = std::cos(hr), sr = std::sin(hr);
  Q4 q;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  q.w = cr * cp * cy + sr * sp * sy;
  return q;
}

// GlobalState as 19 doubles: p v q(wxyz) ba bw g
struct State {
  V3 p, v;
  Q4 q;
  V3 ba, bw, g;
};
State load_state(const double* s) {
  return {{s[0], s[1], s[2]}, {s[3], s[4], s[5]}, {s[6], s[7], s[8], s[9]},
          {s[10], s[11], s[12]}, {s[13], s[14], s[15]}, {s[16], s[17], s[18]}};
}
void store_state(const State& st, double* s) {
  double v[19] = {st.p.x, st.p.y, st.p.z, st.v.x, st.v.y, st.v.z, st.q.w, st.q.x, st.q.y, st.q.z,
                  st.ba.x, st.ba.y, st.ba.z, st.bw.x, st.bw.y, st.bw.z, st.g.x, st.g.y, st.g.z};
  std::memcpy(s, v, sizeof v);
}
// KalmanFilter.hpp:71-81
State box_plus(const State& a, const double* dx) {
  State o;
  o.p = a.p + V3{dx[0], dx[1], dx[2]};
  o.v = a.v + V3{dx[3], dx[4], dx[5]};
  o.ba = a.ba + V3{dx[9], dx[10], dx[11]};
  o.bw = a.bw + V3{dx[12], dx[13], dx[14]};
  o.q = qnormalized(qmul(a.q, axis2quat({dx[6], dx[7], dx[8]})));
  o.g = a.g + V3{dx[15], dx[16], dx[17]};
  return o;
}
// KalmanFilter.hpp:84-94 : xk = a (-) b
void box_minus(const State& a, const State& b, double* xk) {
  V3 dp = a.p - b.p, dv = a.v - b.v, dba = a.ba - b.ba, dbw = a.bw - b.bw, dg = a.g - b.g;
  V3 da = quat2axis(qmul(qinverse(b.q), a.q));
  double v[18] = {dp.x, dp.y, dp.z, dv.x, dv.y, dv.z, da.x, da.y, da.z,
                  dba.x, dba.y, dba.z, dbw.x, dbw.y, dbw.z, dg.x, dg.y, dg.z};
  std::memcpy(xk, v, sizeof v);
}

// ----------------------------------------------------------------------------
// exact 1-NN: brute force (index truth) and a kd-tree (honest CPU cost)
// ----------------------------------------------------------------------------
inline float sqdist(const lins_point& a, const lins_point& b) {
  float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return dx * dx + dy * dy + dz * dz;  // ((dx*dx + dy*dy) + dz*dz), no contraction
}

struct NnResult {
  int idx;
  float d;
};

NnResult nn_brute(const lins_point* t, int n, const lins_point& q) {
  NnResult r{-1, INFINITY};
  for (int j = 0; j < n; ++j) {
    float d = sqdist(t[j], q);
    if (d < r.d) r = {j, d};  // ascending j + strict < : lowest index wins ties
  }
  return r;
}

class KdTree {
 public:
  void build(const lins_point* pts, int n) {
    pts_ = pts;
    n_ = n;
    order_.resize(n);
    for (int i = 0; i < n; ++i) order_[i] = i;
    nodes_.clear();
    nodes_.reserve(n / 4 + 8);
    if (n > 0) build_rec(0, n);
  }
  NnResult query(const lins_point& q) const {
    NnResult best{-1, INFINITY};
    if (n_ > 0) search(0, q, best);
    return best;
  }

 private:
  struct Node {
    int lo, hi;       // leaf: range in order_
    int left, right;  // children (-1 for leaf)
    int axis;
    float split;
  };
  static float coord(const lins_point& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }
  int build_rec(int lo, int hi) {
    int id = (int)nodes_.size();
    nodes_.push_back({lo, hi, -1, -1, 0, 0.f});
    if (hi - lo <= 10) return id;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; ++i)
      for (int a = 0; a < 3; ++a) {
        float c = coord(pts_[order_[i]], a);
        mn[a] = std::min(mn[a], c);
        mx[a] = std::max(mx[a], c);
      }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
      if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
    if (!(mx[axis] > mn[axis])) return id;  // all coincident (or NaN): keep leaf
    int mid = (lo + hi) / 2;
    std::nth_element(order_.begin() + lo, order_.begin() + mid, order_.begin() + hi,
                     [&](int a, int b) { return coord(pts_[a], axis) < coord(pts_[b], axis); });
    float split = coord(pts_[order_[mid]], axis);
    int l = build_rec(lo, mid);
    int r = build_rec(mid, hi);
    nodes_[id].left = l;
    nodes_[id].right = r;
    nodes_[id].axis = axis;
    nodes_[id].split = split;
    return id;
  }
  void search(int id, const lins_point& q, NnResult& best) const {
    const Node& nd = nodes_[id];
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; ++i) {
        int j = order_[i];
        float d = sqdist(pts_[j], q);
        if (d < best.d || (d == best.d && j < best.idx)) best = {j, d};
      }
      return;
    }
    float diff = coord(q, nd.axis) - nd.split;
    int near = diff < 0 ? nd.left : nd.right;
    int far = diff < 0 ? nd.right : nd.left;
    search(near, q, best);
    // rounding of (dx*dx+dy*dy)+dz*dz is monotone, so diff*diff is a valid lower
    // bound on every far-side distance; '<=' keeps equal-distance ties reachable.
    if (diff * diff <= best.d) search(far, q, best);
  }
  const lins_point* pts_ = nullptr;
  int n_ = 0;
  std::vector<int> order_;
  std::vector<Node> nodes_;
};

struct Targets {
  const lins_point* pts;
  int n;
  KdTree tree;
  int mode;
  NnResult nn(const lins_point& q) const {
    return mode == ORACLE_NN_KDTREE ? tree.query(q) : nn_brute(pts, n, q);
  }
};

// ----------------------------------------------------------------------------
// transformToStart  (StateEstimator.hpp:1066-1080)
// ----------------------------------------------------------------------------
inline double rel_time_scale(const lins_params& prm, const lins_point& p) {
  float frac = p.intensity - (float)(int)p.intensity;    // f32 subtract
  return (double)(1.f / prm.scan_period) * (double)frac;  // (1.f/SCAN_PERIOD) is double
}
lins_point transform_to_start(const lins_params& prm, const State& lin, const lins_point& pi) {
  double s = rel_time_scale(prm, pi);
  V3 p2{pi.x, pi.y, pi.z};
  V3 phi = quat2axis(lin.q);
  Q4 r = axis2quat(s * phi);  // reference's .normalized() result is discarded
  V3 p1 = qrot(r, p2) + s * lin.p;
  return {(float)p1.x, (float)p1.y, (float)p1.z, pi.intensity};
}

// ----------------------------------------------------------------------------
// correspondence passes
// ----------------------------------------------------------------------------
inline int ring_of(const lins_point& p) { return (int)p.intensity; }

void find_surf(const lins_params& prm, const State& lin, const lins_point* q, int nq,
               const Targets& tg, int iter, lins_corr* out, bool search) {
  const float thr = (float)prm.nearest_sq_dist;
  const lins_point* t = tg.pts;
  for (int i = 0; i < nq; ++i) {
    lins_corr& c = out[i];
    lins_point sel = transform_to_start(prm, lin, q[i]);
    c.sel[0] = sel.x, c.sel[1] = sel.y, c.sel[2] = sel.z, c.sel[3] = sel.intensity;
    if (search) {  // iterCount % ICP_FREQ == 0  (SE:844)
      int closest = -1, m2 = -1, m3 = -1;
      NnResult nn = tg.n > 0 ? tg.nn(sel) : NnResult{-1, INFINITY};
      if (nn.idx >= 0 && (double)nn.d < prm.nearest_sq_dist) {
        closest = nn.idx;
        int ring = ring_of(t[closest]);
        float d2 = thr, d3 = thr;
        // forward walk bounded by the QUERY count (SE:859) and, our guard, by the
        // target count (the reference would read out of bounds).
        int fend = std::min(nq, tg.n);
        for (int j = closest + 1; j < fend; ++j) {
          if (ring_of(t[j]) > ring + 2.5) break;
          float d = sqdist(t[j], sel);
          if (ring_of(t[j]) <= ring) {
            if (d < d2) d2 = d, m2 = j;
          } else {
            if (d < d3) d3 = d, m3 = j;
          }
        }
        for (int j = closest - 1; j >= 0; --j) {
          if (ring_of(t[j]) < ring - 2.5) break;
          float d = sqdist(t[j], sel);
          if (ring_of(t[j]) >= ring) {
            if (d < d2) d2 = d, m2 = j;
          } else {
            if (d < d3) d3 = d, m3 = j;
          }
        }
      }
      c.ind1 = closest, c.ind2 = m2, c.ind3 = m3;
    }
    c.accepted = 0;
    c.coeff[0] = c.coeff[1] = c.coeff[2] = c.coeff[3] = 0.f;
    if (c.ind2 >= 0 && c.ind3 >= 0) {
      V3 p0{sel.x, sel.y, sel.z};
      V3 p1{t[c.ind1].x, t[c.ind1].y, t[c.ind1].z};
      V3 p2{t[c.ind2].x, t[c.ind2].y, t[c.ind2].z};
      V3 p3{t[c.ind3].x, t[c.ind3].y, t[c.ind3].z};
      V3 m = cross(p1 - p2, p1 - p3);
      double r = dot(p0 - p1, m);
      double mn = norm(m);
      float res = (float)(r / mn);
      V3 jac = m / mn;
      float s = 1;
      if (iter >= prm.icp_freq) {
        float n2 = sel.x * sel.x + sel.y * sel.y + sel.z * sel.z;
        s = (float)(1 - 1.8 * std::fabs(res) / std::sqrt(std::sqrt(n2)));  // f32 sqrt twice
      }
      if (s > 0.1 && res != 0) {
        c.accepted = 1;
        c.coeff[0] = (float)(s * jac.x);
        c.coeff[1] = (float)(s * jac.y);
        c.coeff[2] = (float)(s * jac.z);
        c.coeff[3] = s * res;
      }
    }
  }
}

void find_corner(const lins_params& prm, const State& lin, const lins_point* q, int nq,
                 const Targets& tg, int iter, lins_corr* out, bool search) {
  const float thr = (float)prm.nearest_sq_dist;
  const lins_point* t = tg.pts;
  for (int i = 0; i < nq; ++i) {
    lins_corr& c = out[i];
    lins_point sel = transform_to_start(prm, lin, q[i]);
    c.sel[0] = sel.x, c.sel[1] = sel.y, c.sel[2] = sel.z, c.sel[3] = sel.intensity;
    if (search) {
      int closest = -1, m2 = -1;
      NnResult nn = tg.n > 0 ? tg.nn(sel) : NnResult{-1, INFINITY};
      if (nn.idx >= 0 && (double)nn.d < prm.nearest_sq_dist) {
        closest = nn.idx;
        int ring = ring_of(t[closest]);
        float d2 = thr;
        int fend = std::min(nq, tg.n);
        for (int j = closest + 1; j < fend; ++j) {
          if (ring_of(t[j]) > ring + 2.5) break;
          float d = sqdist(t[j], sel);
          if (ring_of(t[j]) > ring && d < d2) d2 = d, m2 = j;
        }
        for (int j = closest - 1; j >= 0; --j) {
          if (ring_of(t[j]) < ring - 2.5) break;
          float d = sqdist(t[j], sel);
          if (ring_of(t[j]) < ring && d < d2) d2 = d, m2 = j;
        }
      }
      c.ind1 = closest, c.ind2 = m2, c.ind3 = -1;
    }
    c.accepted = 0;
    c.coeff[0] = c.coeff[1] = c.coeff[2] = c.coeff[3] = 0.f;
    if (c.ind2 >= 0) {
      V3 p0{sel.x, sel.y, sel.z};
      V3 p1{t[c.ind1].x, t[c.ind1].y, t[c.ind1].z};
      V3 p2{t[c.ind2].x, t[c.ind2].y, t[c.ind2].z};
      V3 P = cross(p0 - p1, p0 - p2);
      float r = (float)norm(P);
      float d12 = (float)norm(p1 - p2);
      float res = r / d12;
      V3 v = p2 - p1;
      double den = (double)(d12 * r);  // f32 product
      V3 jac{(P.y * v.z - P.z * v.y) / den, (P.z * v.x - P.x * v.z) / den,
             (P.x * v.y - P.y * v.x) / den};
      float s = 1;
      if (iter >= prm.icp_freq) s = (float)(1 - 1.8 * std::fabs(res));
      if (s > 0.1 && res != 0) {
        c.accepted = 1;
        c.coeff[0] = (float)(s * jac.x);
        c.coeff[1] = (float)(s * jac.y);
        c.coeff[2] = (float)(s * jac.z);
        c.coeff[3] = s * res;
      }
    }
  }
}

// ----------------------------------------------------------------------------
// dense helpers (row-major, double)
// ----------------------------------------------------------------------------
using Mat = std::vector<double>;

// in-place lower Cholesky of n x n; NaNs propagate like Eigen's LLT on non-PD
void cholesky(Mat& a, int n) {
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
    d = std::sqrt(d);
    a[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[i * n + j];
      for (int k = 0; k < j; ++k) s -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = s / d;
    }
  }
}
// solve L L^T X = B for X (B is n x m row-major, overwritten)
void chol_solve(const Mat& l, int n, Mat& b, int m) {
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < i; ++k) {
      double f = l[i * n + k];
      for (int c = 0; c < m; ++c) b[i * m + c] -= f * b[k * m + c];
    }
    double d = l[i * n + i];
    for (int c = 0; c < m; ++c) b[i * m + c] /= d;
  }
  for (int i = n - 1; i >= 0; --i) {
    for (int k = i + 1; k < n; ++k) {
      double f = l[k * n + i];
      for (int c = 0; c < m; ++c) b[i * m + c] -= f * b[k * m + c];
    }
    double d = l[i * n + i];
    for (int c = 0; c < m; ++c) b[i * m + c] /= d;
  }
}

// Gaussian elimination with partial pivoting: solves A X = B (n x n, n x m)
bool lu_solve(double* a, int n, double* b, int m) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(a[i * n + k]) > std::fabs(a[p * n + k])) p = i;
    if (p != k) {
      for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]);
      for (int j = 0; j < m; ++j) std::swap(b[k * m + j], b[p * m + j]);
    }
    double piv = a[k * n + k];
    for (int i = k + 1; i < n; ++i) {
      double f = a[i * n + k] / piv;
      for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
      for (int j = 0; j < m; ++j) b[i * m + j] -= f * b[k * m + j];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    for (int j = 0; j < m; ++j) {
      double s = b[i * m + j];
      for (int k = i + 1; k < n; ++k) s -= a[i * n + k] * b[k * m + j];
      b[i * m + j] = s / a[i * n + i];
    }
  }
  return true;
}

const int SIDX[6] = {0, 1, 2, 6, 7, 8};  // pos_ and att_ columns (KF:40-45)

// ----------------------------------------------------------------------------
// one measurement update: rows -> (dx, and on request the Joseph covariance)
// ----------------------------------------------------------------------------
struct Rows {
  int m = 0;
  std::vector<double> h;  // m x 6 : columns pos(3) att(3)
  std::vector<double> r;  // m
  double sums[28];
};

// H / residual assembly (SE:507-532) + the Appendix-C sums.
void assemble(const lins_params& prm, const State& lin, const lins_point* qs, const lins_corr* cs,
              int ns, const lins_point* qc, const lins_corr* cc, int nc, Rows& rows) {
  rows.m = 0;
  rows.h.clear();
  rows.r.clear();
  std::fill(rows.sums, rows.sums + 28, 0.0);
  V3 axis = quat2axis(lin.q);
  M3 R = qmat(lin.q);
  M3 negR;
  for (int k = 0; k < 9; ++k) negR.m[k] = -R.m[k];
  M3 G = rinvleft({-axis.x, -axis.y, -axis.z});
  auto push = [&](const lins_point& kp, const lins_corr& c) {
    V3 p{kp.x, kp.y, kp.z};
    V3 cf{c.coeff[0], c.coeff[1], c.coeff[2]};
    double res = prm.lidar_scale * (double)c.coeff[3];
    V3 hatt = rowmul(rowmul(cf, mmul(negR, skew(p))), G);  // c^T(-R[p]x) Rinvleft(-phi)
    double row[6] = {cf.x, cf.y, cf.z, hatt.x, hatt.y, hatt.z};
    rows.h.insert(rows.h.end(), row, row + 6);
    rows.r.push_back(res);
    rows.m++;
    // 28 sums over the H rows: upper triangle of H^T H, H^T r, r^T r
    double* s = rows.sums;
    int k = 0;
    for (int a = 0; a < 6; ++a)
      for (int b = a; b < 6; ++b) s[k++] += row[a] * row[b];
    for (int a = 0; a < 6; ++a) s[k++] += row[a] * res;
    s[k] += res * res;
  };
  for (int i = 0; i < ns; ++i)  // surf rows first, then corner rows (SE:499-504)
    if (cs[i].accepted) push(qs[i], cs[i]);
  for (int i = 0; i < nc; ++i)
    if (cc[i].accepted) push(qc[i], cc[i]);
}

struct Gain {
  // dense form keeps K (18 x m) for the Joseph update; reduced keeps 6x6 pieces
  std::vector<double> K;
  double A6[36], N6inv_z[6];
};

// Faithful dense form (SE:542-549): S = H P H^T + R, LLT, K = P H^T S^-1.
void update_dense(const lins_params& prm, const double* P, const Rows& rows, const double* d,
                  double* dx, Gain& g) {
  const int m = rows.m;
  // PHt (18 x m)
  std::vector<double> PHt(18 * (size_t)m, 0.0);
  for (int i = 0; i < 18; ++i)
    for (int r = 0; r < m; ++r) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += P[i * 18 + SIDX[k]] * rows.h[r * 6 + k];
      PHt[i * (size_t)m + r] = s;
    }
  Mat S((size_t)m * m);
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += rows.h[a * 6 + k] * PHt[SIDX[k] * (size_t)m + b];
      S[a * (size_t)m + b] = s + (a == b ? prm.lidar_std * prm.lidar_std : 0.0);
    }
  Mat Sinv((size_t)m * m, 0.0);
  for (int a = 0; a < m; ++a) Sinv[a * (size_t)m + a] = 1.0;
  cholesky(S, m);
  chol_solve(S, m, Sinv, m);
  g.K.assign(18 * (size_t)m, 0.0);
  for (int i = 0; i < 18; ++i)
    for (int b = 0; b < m; ++b) {
      double s = 0;
      for (int a = 0; a < m; ++a) s += PHt[i * (size_t)m + a] * Sinv[a * (size_t)m + b];
      g.K[i * (size_t)m + b] = s;
    }
  // dx = -K (r + H d) + d
  std::vector<double> v(m);
  for (int r = 0; r < m; ++r) {
    double s = 0;
    for (int k = 0; k < 6; ++k) s += rows.h[r * 6 + k] * d[SIDX[k]];
    v[r] = rows.r[r] + s;
  }
  for (int i = 0; i < 18; ++i) {
    double s = 0;
    for (int r = 0; r < m; ++r) s += g.K[i * (size_t)m + r] * v[r];
    dx[i] = -s + d[i];
  }
}

// Joseph update with the dense K,H of the last iteration (SE:595-598).
void joseph_dense(const lins_params& prm, const double* P, const Rows& rows, const Gain& g,
                  double* Pout) {
  const int m = rows.m;
  double IKH[324];
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) IKH[i * 18 + j] = (i == j ? 1.0 : 0.0);
  for (int i = 0; i < 18; ++i)
    for (int k = 0; k < 6; ++k) {
      double s = 0;
      for (int r = 0; r < m; ++r) s += g.K[i * (size_t)m + r] * rows.h[r * 6 + k];
      IKH[i * 18 + SIDX[k]] -= s;
    }
  double T[324], O[324];
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double s = 0;
      for (int k = 0; k < 18; ++k) s += IKH[i * 18 + k] * P[k * 18 + j];
      T[i * 18 + j] = s;
    }
  const double r2 = prm.lidar_std * prm.lidar_std;
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double s = 0;
      for (int k = 0; k < 18; ++k) s += T[i * 18 + k] * IKH[j * 18 + k];
      double kk = 0;
      for (int r = 0; r < m; ++r) kk += g.K[i * (size_t)m + r] * r2 * g.K[j * (size_t)m + r];
      O[i * 18 + j] = s + kk;
    }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) Pout[i * 18 + j] = 0.5 * (O[i * 18 + j] + O[j * 18 + i]);
}

// 6x6 block of A = H^T H and g = H^T r from the 28 sums.
void sums_to_normal(const double* s, double* A6, double* g6) {
  int k = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) A6[a * 6 + b] = A6[b * 6 + a] = s[k++];
  for (int a = 0; a < 6; ++a) g6[a] = s[k++];
}

// Reduced (push-through) form: dx = d - P[:,S] (sigma^2 I + A_SS P_SS)^-1 (g + A d)_S.
void update_reduced(const lins_params& prm, const double* P, const double* A6, const double* g6,
                    const double* d, double* dx) {
  const double r2 = prm.lidar_std * prm.lidar_std;
  double N[36], z[6];
  for (int i = 0; i < 6; ++i) {
    double s = g6[i];
    for (int k = 0; k < 6; ++k) s += A6[i * 6 + k] * d[SIDX[k]];
    z[i] = s;
    for (int j = 0; j < 6; ++j) {
      double t = 0;
      for (int k = 0; k < 6; ++k) t += A6[i * 6 + k] * P[SIDX[k] * 18 + SIDX[j]];
      N[i * 6 + j] = t + (i == j ? r2 : 0.0);
    }
  }
  lu_solve(N, 6, z, 1);
  for (int i = 0; i < 18; ++i) {
    double s = 0;
    for (int k = 0; k < 6; ++k) s += P[i * 18 + SIDX[k]] * z[k];
    dx[i] = d[i] - s;
  }
}

// Joseph update in the reduced form:  KH = P[:,S] Y E_S^T with Y = N^-1 A_SS,
// K R K^T = sigma^2 P[:,S] (N^-1 A_SS N^-T) P[:,S]^T.
void joseph_reduced(const lins_params& prm, const double* P, const double* A6, double* Pout) {
  const double r2 = prm.lidar_std * prm.lidar_std;
  double N[36], Y[36];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double t = 0;
      for (int k = 0; k < 6; ++k) t += A6[i * 6 + k] * P[SIDX[k] * 18 + SIDX[j]];
      N[i * 6 + j] = t + (i == j ? r2 : 0.0);
      Y[i * 6 + j] = A6[i * 6 + j];
    }
  double N2[36];
  std::memcpy(N2, N, sizeof N);
  lu_solve(N2, 6, Y, 6);  // Y = N^-1 A
  // Z = Y N^-T  : solve N Z^T = Y^T
  double Zt[36];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) Zt[i * 6 + j] = Y[j * 6 + i];
  std::memcpy(N2, N, sizeof N);
  lu_solve(N2, 6, Zt, 6);  // Zt = N^-1 Y^T  => Z = Y N^-T
  double PS[18 * 6], PSY[18 * 6];
  for (int i = 0; i < 18; ++i)
    for (int k = 0; k < 6; ++k) PS[i * 6 + k] = P[i * 18 + SIDX[k]];
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += PS[i * 6 + k] * Y[k * 6 + j];
      PSY[i * 6 + j] = s;
    }
  double IKH[324];
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) IKH[i * 18 + j] = (i == j ? 1.0 : 0.0);
  for (int i = 0; i < 18; ++i)
    for (int k = 0; k < 6; ++k) IKH[i * 18 + SIDX[k]] -= PSY[i * 6 + k];
  double T[324], O[324];
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double s = 0;
      for (int k = 0; k < 18; ++k) s += IKH[i * 18 + k] * P[k * 18 + j];
      T[i * 18 + j] = s;
    }
  double PSZ[18 * 6];
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += PS[i * 6 + k] * Zt[j * 6 + k];  // Z[k][j] = Zt[j][k]
      PSZ[i * 6 + j] = s;
    }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double s = 0;
      for (int k = 0; k < 18; ++k) s += T[i * 18 + k] * IKH[j * 18 + k];
      double kk = 0;
      for (int k = 0; k < 6; ++k) kk += PSZ[i * 6 + k] * PS[j * 6 + k];
      O[i * 18 + j] = s + r2 * kk;
    }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) Pout[i * 18 + j] = 0.5 * (O[i * 18 + j] + O[j * 18 + i]);
}

// ----------------------------------------------------------------------------
// the IESKF loop (SE:465-583, 594-598)
// ----------------------------------------------------------------------------
int ieskf(const lins_params& prm, const lins_scan_pair& in, int form, int nn_mode, lins_result* out,
          oracle_trace* tr) {
  if (prm.num_iter < 1 || prm.icp_freq < 1) return LINS_E_ARG;
  Targets ts{in.surf_less_flat_last, in.n_surf_last, {}, nn_mode};
  Targets tc{in.corner_less_sharp_last, in.n_corner_last, {}, nn_mode};
  if (nn_mode == ORACLE_NN_KDTREE) {
    ts.tree.build(ts.pts, ts.n);
    tc.tree.build(tc.pts, tc.n);
  }
  const double* P = in.cov;
  State filt = load_state(in.state);
  State lin = filt;
  std::vector<lins_corr> cs(in.n_surf_flat), cc(in.n_corner_sharp);
  for (auto& c : cs) c.ind1 = c.ind2 = c.ind3 = -1;
  for (auto& c : cc) c.ind1 = c.ind2 = c.ind3 = -1;
  Rows rows;
  Gain gain;
  double A6[36] = {0}, g6[6];
  double residual_norm = 1e6, update_norm = 0, last_res = 0;
  bool conv = false;
  int div = 0, iter = 0, ms = 0, mc = 0;
  for (; iter < prm.num_iter && !conv && !div; ++iter) {
    bool search = (iter % prm.icp_freq) == 0;
    if (tr && iter < tr->max_iters && tr->lin_state) store_state(lin, tr->lin_state + 19 * iter);
    find_surf(prm, lin, in.surf_flat, in.n_surf_flat, ts, iter, cs.data(), search);
    find_corner(prm, lin, in.corner_sharp, in.n_corner_sharp, tc, iter, cc.data(), search);
    if (tr && iter < tr->max_iters) {
      if (tr->surf) std::copy(cs.begin(), cs.end(), tr->surf + (size_t)iter * in.n_surf_flat);
      if (tr->corner) std::copy(cc.begin(), cc.end(), tr->corner + (size_t)iter * in.n_corner_sharp);
    }
    assemble(prm, lin, in.surf_flat, cs.data(), in.n_surf_flat, in.corner_sharp, cc.data(),
             in.n_corner_sharp, rows);
    ms = mc = 0;
    for (auto& c : cs) ms += c.accepted;
    for (auto& c : cc) mc += c.accepted;
    if (tr && iter < tr->max_iters && tr->sums28)
      std::memcpy(tr->sums28 + 28 * iter, rows.sums, sizeof rows.sums);
    double d[18], dx[18];
    box_minus(filt, lin, d);
    if (form == ORACLE_FORM_DENSE) {
      update_dense(prm, P, rows, d, dx, gain);
    } else {
      sums_to_normal(rows.sums, A6, g6);
      update_reduced(prm, P, A6, g6, d, dx);
    }
    if (tr && iter < tr->max_iters && tr->dx) std::memcpy(tr->dx + 18 * iter, dx, sizeof dx);
    double rn = 0;
    for (double r : rows.r) rn += r * r;
    rn = std::sqrt(rn);
    last_res = rn;
    bool has_nan = false;
    for (int i = 0; i < 18; ++i)
      if (std::isnan(dx[i])) dx[i] = 0, has_nan = true;
    if (has_nan) {  // SE:552-563
      div = 2;
      ++iter;
      break;
    }
    if (rn > residual_norm * 10) {  // SE:566-570
      div = 1;
      ++iter;
      break;
    }
    lin = box_plus(lin, dx);  // SE:573
    update_norm = 0;
    for (int i = 0; i < 18; ++i) update_norm += dx[i] * dx[i];
    update_norm = std::sqrt(update_norm);
    if (update_norm <= 1e-2 && !prm.fixed_iters) conv = true;  // SE:575-578
    residual_norm = rn;
  }
  std::memset(out, 0, sizeof *out);
  out->iters = iter;
  out->converged = conv;
  out->diverged = div;
  out->m_surf = ms;
  out->m_corner = mc;
  out->residual_norm = last_res;
  out->update_norm = update_norm;
  if (div) {
    std::memcpy(out->state, in.state, sizeof in.state);  // filterState, Pk_ un-updated
    std::memcpy(out->cov, in.cov, sizeof in.cov);
  } else {
    store_state(lin, out->state);
    if (form == ORACLE_FORM_DENSE)
      joseph_dense(prm, P, rows, gain, out->cov);
    else
      joseph_reduced(prm, P, A6, out->cov);
  }
  return LINS_OK;
}

// ----------------------------------------------------------------------------
// ICP fallback (SE:1163-1320)
// ----------------------------------------------------------------------------
// x = colPivHouseholderQr(A).solve(b), 6x6, Eigen's rank rule
void colpiv_qr_solve6(const double* Ain, const double* bin, double* x) {
  const int n = 6;
  double a[36], b[6];
  std::memcpy(a, Ain, sizeof a);
  std::memcpy(b, bin, sizeof b);
  int perm[6] = {0, 1, 2, 3, 4, 5};
  double maxpivot = 0;
  double rdiag[6];
  int rank = n;
  double thresh = 0;
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
    // Householder on column k
    double nrm = 0;
    for (int i = k; i < n; ++i) nrm += a[i * n + k] * a[i * n + k];
    nrm = std::sqrt(nrm);
    double alpha = a[k * n + k] >= 0 ? -nrm : nrm;
    if (nrm > 0) {
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
    rdiag[k] = a[k * n + k];
    maxpivot = std::max(maxpivot, std::fabs(rdiag[k]));
  }
  thresh = maxpivot * 2.220446049250313e-16 * n;
  rank = 0;
  for (int k = 0; k < n; ++k)
    if (std::fabs(rdiag[k]) > thresh) ++rank;
  double y[6] = {0};
  for (int i = rank - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < rank; ++j) s -= a[i * n + j] * y[j];
    y[i] = s / a[i * n + i];
  }
  for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
}

// cyclic Jacobi for symmetric 6x6: eigenvalues ascending, eigenvectors in columns,
// each column sign-normalised (largest |component| positive) — our convention,
// Eigen's sign is an implementation detail that cannot be reproduced here.
void sym_eig6(const double* Ain, double* evals, double* V) {
  const int n = 6;
  double a[36];
  std::memcpy(a, Ain, sizeof a);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (a[p * n + q] == 0) continue;
        double theta = (a[q * n + q] - a[p * n + p]) / (2 * a[p * n + q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < n; ++k) {
          double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk;
          a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  int ord[6] = {0, 1, 2, 3, 4, 5};
  std::sort(ord, ord + n, [&](int x, int y) { return a[x * n + x] < a[y * n + y]; });
  double Vs[36];
  for (int j = 0; j < n; ++j) {
    evals[j] = a[ord[j] * n + ord[j]];
    int big = 0;
    for (int i = 1; i < n; ++i)
      if (std::fabs(V[i * n + ord[j]]) > std::fabs(V[big * n + ord[j]])) big = i;
    double sg = V[big * n + ord[j]] < 0 ? -1.0 : 1.0;
    for (int i = 0; i < n; ++i) Vs[i * n + j] = sg * V[i * n + ord[j]];
  }
  std::memcpy(V, Vs, sizeof Vs);
}

// calculateTransformation (SE:1198-1320). Returns true on convergence.
bool gn_step(const lins_params& prm, State& lin, const lins_point* qs, const lins_corr* cs, int ns,
             const lins_point* qc, const lins_corr* cc, int nc, int iter) {
  double JTJ[36] = {0}, JTb[6] = {0};
  V3 phi = quat2axis(lin.q);
  auto row = [&](const lins_point& kp, const lins_corr& c) {
    V3 p{kp.x, kp.y, kp.z};
    V3 cf{c.coeff[0], c.coeff[1], c.coeff[2]};
    double s = rel_time_scale(prm, kp);
    M3 R = qmat(axis2quat(s * phi));
    M3 negR;
    for (int k = 0; k < 9; ++k) negR.m[k] = -R.m[k];
    V3 j1 = rowmul(cf, mmul(negR, skew(p)));
    double J[6] = {j1.x, j1.y, j1.z, cf.x, cf.y, cf.z};  // O_R = 0, O_P = 3
    double b = -0.05 * (double)c.coeff[3];
    for (int a = 0; a < 6; ++a) {
      for (int e = 0; e < 6; ++e) JTJ[a * 6 + e] += J[a] * J[e];
      JTb[a] += J[a] * b;
    }
  };
  for (int i = 0; i < ns; ++i)
    if (cs[i].accepted) row(qs[i], cs[i]);
  for (int i = 0; i < nc; ++i)
    if (cc[i].accepted) row(qc[i], cc[i]);
  double x[6];
  colpiv_qr_solve6(JTJ, JTb, x);
  if (iter == 0) {  // degeneracy projection, SE:1269-1302
    double E[6], V[36], V2[36];
    sym_eig6(JTJ, E, V);
    std::memcpy(V2, V, sizeof V);
    bool degenerate = false;
    for (int i = 0; i < 6; ++i) {
      if (E[i] < 10.) {
        for (int j = 0; j < 6; ++j) V2[i * 6 + j] = 0;  // zeroes ROW i (as the reference does)
        degenerate = true;
      } else {
        break;
      }
    }
    if (degenerate) {
      // matP = matV^-1 * matV2 ;  x = matP * x
      double Vc[36], Pm[36];
      std::memcpy(Vc, V, sizeof V);
      std::memcpy(Pm, V2, sizeof V2);
      lu_solve(Vc, 6, Pm, 6);
      double x2[6];
      for (int i = 0; i < 6; ++i) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += Pm[i * 6 + k] * x[k];
        x2[i] = s;
      }
      std::memcpy(x, x2, sizeof x2);
    }
  }
  lin.q = qnormalized(qmul(lin.q, rpy2quat({x[0], x[1], x[2]})));
  lin.p = lin.p + V3{x[3], x[4], x[5]};
  const double r2d = 180.0 / M_PI;
  double dR = std::sqrt((x[0] * r2d) * (x[0] * r2d) + (x[1] * r2d) * (x[1] * r2d) + (x[2] * r2d) * (x[2] * r2d));
  double dT = std::sqrt((100 * x[3]) * (100 * x[3]) + (100 * x[4]) * (100 * x[4]) + (100 * x[5]) * (100 * x[5]));
  return dR < 0.1 && dT < 0.1;
}

int icp(const lins_params& prm, const lins_scan_pair& in, double* t, double* q, int nn_mode,
        int32_t* iters_run) {
  Targets ts{in.surf_less_flat_last, in.n_surf_last, {}, nn_mode};
  Targets tc{in.corner_less_sharp_last, in.n_corner_last, {}, nn_mode};
  if (nn_mode == ORACLE_NN_KDTREE) {
    ts.tree.build(ts.pts, ts.n);
    tc.tree.build(tc.pts, tc.n);
  }
  State lin = load_state(in.state);
  lin.p = {t[0], t[1], t[2]};
  lin.q = {q[0], q[1], q[2], q[3]};
  std::vector<lins_corr> cs(in.n_surf_flat), cc(in.n_corner_sharp);
  for (auto& c : cs) c.ind1 = c.ind2 = c.ind3 = -1;
  for (auto& c : cc) c.ind1 = c.ind2 = c.ind3 = -1;
  int iter = 0;
  for (; iter < prm.num_iter; ++iter) {
    bool search = (iter % prm.icp_freq) == 0;
    find_surf(prm, lin, in.surf_flat, in.n_surf_flat, ts, iter, cs.data(), search);
    int ms = 0;
    for (auto& c : cs) ms += c.accepted;
    if (ms < 10) continue;  // SE:1175-1178
    find_corner(prm, lin, in.corner_sharp, in.n_corner_sharp, tc, iter, cc.data(), search);
    int mc = 0;
    for (auto& c : cc) mc += c.accepted;
    if (mc < 5) continue;  // SE:1181-1184
    if (gn_step(prm, lin, in.surf_flat, cs.data(), in.n_surf_flat, in.corner_sharp, cc.data(),
                in.n_corner_sharp, iter)) {
      ++iter;
      break;
    }
  }
  if (iters_run) *iters_run = iter;
  t[0] = lin.p.x, t[1] = lin.p.y, t[2] = lin.p.z;
  q[0] = lin.q.w, q[1] = lin.q.x, q[2] = lin.q.y, q[3] = lin.q.z;
  return LINS_OK;
}

}  // namespace

// ----------------------------------------------------------------------------
// C API
// ----------------------------------------------------------------------------
extern "C" {

int oracle_correspondences(const lins_params* prm, const lins_scan_pair* in, const double* lin_state,
                           int iter, int nn_mode, lins_corr* surf, lins_corr* corner) {
  if (!prm || !in || !lin_state || in->point_stride_bytes == 32) return LINS_E_ARG;  // (the checker takes packed points)
  Targets ts{in->surf_less_flat_last, in->n_surf_last, {}, nn_mode};
  Targets tc{in->corner_less_sharp_last, in->n_corner_last, {}, nn_mode};
  if (nn_mode == ORACLE_NN_KDTREE) {
    ts.tree.build(ts.pts, ts.n);
    tc.tree.build(tc.pts, tc.n);
  }
  State lin = load_state(lin_state);
  if (surf) find_surf(*prm, lin, in->surf_flat, in->n_surf_flat, ts, iter, surf, true);
  if (corner) find_corner(*prm, lin, in->corner_sharp, in->n_corner_sharp, tc, iter, corner, true);
  return LINS_OK;
}

int oracle_ieskf(const lins_params* prm, const lins_scan_pair* in, int form, int nn_mode,
                 lins_result* out, oracle_trace* trace) {
  if (!prm || !in || !out || in->point_stride_bytes == 32) return LINS_E_ARG;  // (the checker takes packed points)
  return ieskf(*prm, *in, form, nn_mode, out, trace);
}

int oracle_icp(const lins_params* prm, const lins_scan_pair* in, double* t, double* q, int nn_mode,
               int32_t* iters_run) {
  if (!prm || !in || !t || !q || in->point_stride_bytes == 32) return LINS_E_ARG;
  return icp(*prm, *in, t, q, nn_mode, iters_run);
}

int oracle_perform_ieskf(const lins_params* prm, const lins_scan_pair* in, int form, int nn_mode,
                         lins_result* out) {
  int rc = oracle_ieskf(prm, in, form, nn_mode, out, nullptr);
  if (rc != LINS_OK) return rc;
  if (out->diverged) {  // SE:585-592: ICP from the filter's pose, covariance un-updated
    double t[3] = {in->state[0], in->state[1], in->state[2]};
    double q[4] = {in->state[6], in->state[7], in->state[8], in->state[9]};
    rc = oracle_icp(prm, in, t, q, nn_mode, nullptr);
    std::memcpy(out->state, in->state, sizeof in->state);
    out->state[0] = t[0], out->state[1] = t[1], out->state[2] = t[2];
    out->state[6] = q[0], out->state[7] = q[1], out->state[8] = q[2], out->state[9] = q[3];
  }
  return rc;
}

// oracle_perform_ieskf behind the signature of lins_host_perform_ieskf (a context pointer first): what the CPU test of
// the in-situ checker hands to oracle/ref_seq_driver.cpp's hook instead of the GPU path (dense M x M form, kd-tree)
int oracle_perform_ieskf_hook(void* /*user*/, const lins_params* prm, const lins_scan_pair* in, lins_result* out,
                              int32_t* used_icp) {
  const int rc = oracle_perform_ieskf(prm, in, ORACLE_FORM_DENSE, ORACLE_NN_KDTREE, out);
  if (used_icp) *used_icp = (rc == LINS_OK && out->diverged) ? 1 : 0;
  return rc;
}

int oracle_nn(const lins_point* targets, int n_targets, const lins_point* queries, int n_queries,
              int nn_mode, int32_t* idx, float* sqd) {
  KdTree tree;
  if (nn_mode == ORACLE_NN_KDTREE) tree.build(targets, n_targets);
  for (int i = 0; i < n_queries; ++i) {
    NnResult r = nn_mode == ORACLE_NN_KDTREE ? tree.query(queries[i]) : nn_brute(targets, n_targets, queries[i]);
    idx[i] = r.idx;
    if (sqd) sqd[i] = r.d;
  }
  return LINS_OK;
}

void oracle_quat2axis(const double* q, double* a) {
  V3 v = quat2axis({q[0], q[1], q[2], q[3]});
  a[0] = v.x, a[1] = v.y, a[2] = v.z;
}
void oracle_axis2quat(const double* a, double* q) {
  Q4 r = axis2quat({a[0], a[1], a[2]});
  q[0] = r.w, q[1] = r.x, q[2] = r.y, q[3] = r.z;
}
void oracle_rinvleft(const double* a, double* m9) {
  M3 r = rinvleft({a[0], a[1], a[2]});
  std::memcpy(m9, r.m, sizeof r.m);
}
void oracle_box_plus(const double* s, const double* dx, double* out) {
  store_state(box_plus(load_state(s), dx), out);
}
void oracle_box_minus(const double* a, const double* b, double* out) {
  box_minus(load_state(a), load_state(b), out);
}
void oracle_transform_to_start(const lins_params* prm, const double* lin_state, const lins_point* in,
                               lins_point* out) {
  *out = transform_to_start(*prm, load_state(lin_state), *in);
}

int oracle_bench(const lins_params* prm, int n, const lins_scan_pair* in, int form, int nn_mode,
                 int threads, double* seconds, uint64_t* iters) {
  if (!prm || !in || n < 0 || threads < 1) return LINS_E_ARG;
  std::vector<uint64_t> it(threads, 0);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int w = 0; w < threads; ++w)
    pool.emplace_back([&, w] {
      lins_result r;
      for (int i = w; i < n; i += threads) {
        ieskf(*prm, in[i], form, nn_mode, &r, nullptr);
        it[w] += (uint64_t)r.iters;
      }
    });
  for (auto& th : pool) th.join();
  auto t1 = std::chrono::steady_clock::now();
  *seconds = std::chrono::duration<double>(t1 - t0).count();
  uint64_t tot = 0;
  for (auto v : it) tot += v;
  *iters = tot;
  return LINS_OK;
}

}  // extern "C"
