// lins_math.h — SO(3)/quaternion helpers shared by the HIP kernels (device) and
// the host-side C++ (front-end, predictor, ICP fallback).
//
// Behavioural contract (cited lines are /root/reference/lins/include/...):
//   wrap_pi, axis2Quat, Quat2axis, skew, Rinvleft, rpy2Quat
//                         math_utils.h:27-37, 43-88, 131-148, 196-204, 304-321
//   quaternion product / rotate / toRotationMatrix / normalized / inverse follow
//   the Eigen::Quaternion formulas the reference relies on, so that rounding
//   stays at the 1e-16 level against it.
// All f64.  Compiled with -ffp-contract=off on both sides.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LINS_HD __host__ __device__ __forceinline__
#else
#define LINS_HD inline
#endif

namespace lins {

struct V3 {
  double x, y, z;
};
struct Q4 {
  double w, x, y, z;
};
struct M3 {
  double m[9];  // row-major
};

LINS_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
LINS_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
LINS_HD V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
LINS_HD V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
LINS_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LINS_HD V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
LINS_HD double norm(V3 a) { return sqrt(dot(a, a)); }

LINS_HD Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
LINS_HD Q4 qnormalized(Q4 q) {
  double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return {q.w / n, q.x / n, q.y / n, q.z / n};
}
LINS_HD Q4 qinverse(Q4 q) {
  double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
// v + 2w (qv x v) + 2 qv x (qv x v)
LINS_HD V3 qrot(Q4 q, V3 v) {
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}
LINS_HD M3 qmat(Q4 q) {
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
LINS_HD M3 skew(V3 q) { return {{0, -q.z, q.y, q.z, 0, -q.x, -q.y, q.x, 0}}; }
LINS_HD M3 mmul(const M3& a, const M3& b) {
  M3 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      c.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return c;
}
LINS_HD M3 mtrans(const M3& a) {
  return {{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}};
}
LINS_HD V3 mvec(const M3& a, V3 v) {
  return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
          a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
LINS_HD V3 rowmul(V3 r, const M3& a) {  // (r^T A)^T
  return {r.x * a.m[0] + r.y * a.m[3] + r.z * a.m[6], r.x * a.m[1] + r.y * a.m[4] + r.z * a.m[7],
          r.x * a.m[2] + r.y * a.m[5] + r.z * a.m[8]};
}

// atan2 in f32 with a FIXED operation sequence (one division, a degree-9 odd polynomial after the
// classic two-octant reduction; <= 2 ulp of pi/4 from the true value): the feature front-end bins points by
// angle and tags them with angle-derived times, so the host restatement and the device kernels must
// agree to the bit — libm's and ocml's atan2f do not.  Same sign / quadrant conventions as atan2
// (atan2(+-0, x < 0) = +-pi, atan2(0, 0) = 0).
LINS_HD float lins_atan2f(float y, float x) {
  const float ax = x < 0.f ? -x : x, ay = y < 0.f ? -y : y;
  const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
  float r = 0.f;
  if (mx > 0.f) {
    float a = mn / mx, base = 0.f;  // a in [0, 1]
    if (a > 0.41421356f) {          // tan(pi/8): atan(a) = pi/4 + atan((a - 1) / (a + 1))
      base = 0.78539816f;
      a = (a - 1.f) / (a + 1.f);
    }
    const float z = a * a;
    r = base + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * a + a);
    if (ay > ax) r = 1.57079633f - r;
  }
  if (x < 0.f) r = 3.14159265f - r;
  return y < 0.f ? -r : ((y == 0.f && 1.f / y < 0.f) ? -r : r);
}

LINS_HD double wrap_pi(double x) {
  const double pi = 3.14159265358979323846;
  while (x >= pi) x -= 2.0 * pi;
  while (x < -pi) x += 2.0 * pi;
  return x;
}
LINS_HD Q4 axis2quat(V3 v) {
  double theta = norm(v);
  if (theta < 1e-10) return {1, 0, 0, 0};
  V3 a = v / theta;
  double mag = sin(theta / 2.0);
  return {cos(theta / 2.0), a.x * mag, a.y * mag, a.z * mag};
}
LINS_HD V3 quat2axis(Q4 q) {
  double mag = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  V3 v{q.x, q.y, q.z};
  if (mag >= 1e-10) {
    v = v / mag;
    v = wrap_pi(2.0 * atan2(mag, q.w)) * v;
  }
  return v;
}
LINS_HD M3 rinvleft(V3 axis) {
  double theta = norm(axis);
  M3 r{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  if (theta < 1e-10) return r;
  double h = theta / 2.0;
  V3 a = axis / theta;
  double s = h * (cos(h) / sin(h));
  M3 k = skew(a);
  double av[3] = {a.x, a.y, a.z};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r.m[i * 3 + j] = (s * (i == j ? 1.0 : 0.0) + (1.0 - s) * av[i] * av[j]) - h * k.m[i * 3 + j];
  return r;
}
LINS_HD Q4 rpy2quat(V3 rpy) {
  double hy = rpy.z * 0.5, hp = rpy.y * 0.5, hr = rpy.x * 0.5;
  double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
  Q4 q;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  q.w = cr * cp * cy + sr * sp * sy;
  return q;
}

}  // namespace lins
