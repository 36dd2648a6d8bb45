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

// atan2 to within 4e-3 rad in a dozen instructions: a (pi/4 + 0.273 (1 - a)) on the first octant (a = min / max of
// |x|, |y|; maximum error 3.8e-3 rad) and the usual reflections.  For BINNING only (the azimuth columns of the LDS
// grid, ieskf_lds_impl.h az_bin_lds, where the search windows carry a whole column of slack): never for geometry.
LINS_HD float lins_atan2_coarse(float y, float x) {
  const float ax = x < 0.f ? -x : x, ay = y < 0.f ? -y : y;
  const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
#if defined(__HIP_DEVICE_COMPILE__)
  const float a = mx > 0.f ? mn * __frcp_rn(mx) : 0.f;  // (the origin: angle 0)
#else
  const float a = mx > 0.f ? mn / mx : 0.f;
#endif
  float r = a * (0.78539816f + 0.273f * (1.f - a));
  if (ay > ax) r = 1.57079633f - r;
  if (x < 0.f) r = 3.14159265f - r;
  return y < 0.f ? -r : r;
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
// ---------------------------------------------------------------------------
// Short-series forms of the same maps for SMALL rotations (device: the serial tail between two IESKF iterations,
// where one wave walks through these formulas while the rest of the workgroup waits — every dependent instruction
// of that chain is paid in full).  libm's sin / cos / atan2 are general-range routines of 60-100 dependent
// instructions each; for the angles that occur here (an update step, the rotation over one scan) the Taylor series
// in z = (half angle)^2 or tan^2 is exact to the last ulp or two after a dozen fused multiply-adds, and the
// square root, the unit axis and most divisions of the textbook route cancel algebraically:
//   axis2Quat(v):  (cos h, (v / |v|) sin h),  h = |v| / 2      =  (C(z), v * S(z) / 2),  z = |v|^2 / 4
//   Quat2axis(q):  2 atan2(|qv|, w) qv / |qv|,  w > 0          =  qv * 2 A(z) / w,       z = |qv|^2 / w^2
// with S(z) = sin(sqrt z) / sqrt z, C(z) = cos(sqrt z), A(z) = atan(sqrt z) / sqrt z.  Outside the stated ranges the
// callers take the libm route; the 1e-10 branches of MU:61-88 are kept (on the squared norm).  fma() is correctly
// rounded on host and device, so the host build reproduces the device bits (tests/test_fastmath.py measures the
// distance to libm in long double: <= 2 ulp).
// ---------------------------------------------------------------------------
LINS_HD double lins_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fma(a, b, c);
#else
  return fma(a, b, c);
#endif
}
// sin(h) / h and cos(h), z = h^2 <= 0.25 (|h| <= 0.5): remainders below 1e-19 relative
LINS_HD void lins_sinc_cos_small(double z, double& sinc, double& c) {
  double s = -1.0 / 121645100408832000.0;  // 1/19!
  s = lins_fma(s, z, 1.0 / 355687428096000.0);   // 1/17!
  s = lins_fma(s, z, -1.0 / 1307674368000.0);    // 1/15!
  s = lins_fma(s, z, 1.0 / 6227020800.0);        // 1/13!
  s = lins_fma(s, z, -1.0 / 39916800.0);         // 1/11!
  s = lins_fma(s, z, 1.0 / 362880.0);            // 1/9!
  s = lins_fma(s, z, -1.0 / 5040.0);             // 1/7!
  s = lins_fma(s, z, 1.0 / 120.0);               // 1/5!
  s = lins_fma(s, z, -1.0 / 6.0);                // 1/3!
  sinc = lins_fma(s, z, 1.0);
  double k = 1.0 / 6402373705728000.0;   // 1/18!
  k = lins_fma(k, z, -1.0 / 20922789888000.0);   // 1/16!
  k = lins_fma(k, z, 1.0 / 87178291200.0);       // 1/14!
  k = lins_fma(k, z, -1.0 / 479001600.0);        // 1/12!
  k = lins_fma(k, z, 1.0 / 3628800.0);           // 1/10!
  k = lins_fma(k, z, -1.0 / 40320.0);            // 1/8!
  k = lins_fma(k, z, 1.0 / 720.0);               // 1/6!
  k = lins_fma(k, z, -1.0 / 24.0);               // 1/4!
  k = lins_fma(k, z, 0.5);                       // 1/2!
  c = lins_fma(-k, z, 1.0);
}
// A(z) = atan(t) / t and Q(z) = (1 - A(z)) / z for z = t^2 <= 1/64 (t <= 0.125): the alternating series
// 1 - z/3 + z^2/5 - ..., cut after z^11 / 23 (next term < 1.6e-23)
LINS_HD void lins_atanc_small(double z, double& A, double& Q) {
  double q = 1.0 / 25.0;
  q = lins_fma(-q, z, 1.0 / 23.0);
  q = lins_fma(-q, z, 1.0 / 21.0);
  q = lins_fma(-q, z, 1.0 / 19.0);
  q = lins_fma(-q, z, 1.0 / 17.0);
  q = lins_fma(-q, z, 1.0 / 15.0);
  q = lins_fma(-q, z, 1.0 / 13.0);
  q = lins_fma(-q, z, 1.0 / 11.0);
  q = lins_fma(-q, z, 1.0 / 9.0);
  q = lins_fma(-q, z, 1.0 / 7.0);
  q = lins_fma(-q, z, 1.0 / 5.0);
  q = lins_fma(-q, z, 1.0 / 3.0);
  Q = q;
  A = lins_fma(-q, z, 1.0);
}
constexpr double kFastHalfAngleSq = 0.25;     // axis2quat_fast: |v| <= 1 rad
constexpr double kFastTanSq = 1.0 / 64.0;     // quat2axis_fast: tan(angle / 2) <= 1/8, i.e. angle <= 0.2487 rad

LINS_HD Q4 axis2quat_fast(V3 v) {
  const double n2 = dot(v, v);
  if (n2 < 1e-20) return {1, 0, 0, 0};  // theta < 1e-10 (MU:63)
  const double z = 0.25 * n2;
  if (!(z <= kFastHalfAngleSq)) return axis2quat(v);
  double sinc, c;
  lins_sinc_cos_small(z, sinc, c);
  const double k = 0.5 * sinc;
  return {c, v.x * k, v.y * k, v.z * k};
}
// The coefficients of lins_sinc_cos_small as a table (highest power first: 9 of sin(h)/h — the final 1 is implied —
// then 9 of (1 - cos h)/h^2), for callers that want them read from memory where they are used instead of living in
// registers: axis2quat_tab below is axis2quat_fast with the table (same operations, same bits).
constexpr int kSincCosTab = 18;
LINS_HD void lins_sinc_cos_table(double* t) {
  t[0] = -1.0 / 121645100408832000.0, t[1] = 1.0 / 355687428096000.0, t[2] = -1.0 / 1307674368000.0;
  t[3] = 1.0 / 6227020800.0, t[4] = -1.0 / 39916800.0, t[5] = 1.0 / 362880.0, t[6] = -1.0 / 5040.0;
  t[7] = 1.0 / 120.0, t[8] = -1.0 / 6.0;
  t[9] = 1.0 / 6402373705728000.0, t[10] = -1.0 / 20922789888000.0, t[11] = 1.0 / 87178291200.0;
  t[12] = -1.0 / 479001600.0, t[13] = 1.0 / 3628800.0, t[14] = -1.0 / 40320.0, t[15] = 1.0 / 720.0;
  t[16] = -1.0 / 24.0, t[17] = 0.5;
}
LINS_HD Q4 axis2quat_tab(V3 v, const double* t) {
  const double n2 = dot(v, v);
  if (n2 < 1e-20) return {1, 0, 0, 0};  // theta < 1e-10 (MU:63)
  const double z = 0.25 * n2;
  if (!(z <= kFastHalfAngleSq)) return axis2quat(v);
  double s = t[0], k = t[9];
  for (int i = 1; i < 9; ++i) s = lins_fma(s, z, t[i]), k = lins_fma(k, z, t[9 + i]);
  const double sinc = lins_fma(s, z, 1.0), c = lins_fma(-k, z, 1.0);
  const double h = 0.5 * sinc;
  return {c, v.x * h, v.y * h, v.z * h};
}

LINS_HD V3 quat2axis_fast(Q4 q) {
  const double m2 = q.x * q.x + q.y * q.y + q.z * q.z;
  if (m2 < 1e-20) return {q.x, q.y, q.z};  // mag < 1e-10: the vector part as it is (MU:78-86)
  if (!(q.w > 0.0 && m2 <= kFastTanSq * (q.w * q.w))) return quat2axis(q);
  const double rw = 1.0 / q.w;
  double A, Q;
  lins_atanc_small((m2 * rw) * rw, A, Q);
  const double k = (2.0 * A) * rw;
  return {q.x * k, q.y * k, q.z * k};
}

// ---- transformToEnd (SE:1083-1101) of one point, device form -----------------------------------------------------
//   p1 = R(s phi) p + s t          (to the scan start: s = relative time of the point, phi = Quat2axis(q))
//   p2 = R(q)^-1 (p1 - t)          (to the scan end)
// The per-cloud constants once (phi, R(q)^-1 as a matrix), per point the short-series axis2quat_fast (|s phi| <= 1 rad —
// any inter-scan rotation; libm beyond) and one quaternion rotation: ~90 f64 operations and no libm call, against two
// libm calls and ~300 operations of the text's literal form (rounds 1-3) — the re-projection kernels are streaming
// kernels with this.  Same values to the last bits of f64 but not bit for bit; under the f32 rounding of the result one
// coordinate in ~1e5 moves by an ulp (tests: <= 1 ulp, <= 1e-3 of the coordinates, against the checker's libm form).
struct ToEnd {
  V3 t, phi;
  M3 rinv;
  double inv_period;
};
LINS_HD ToEnd make_to_end(V3 t, Q4 q, double inv_period) { return ToEnd{t, quat2axis(q), qmat(qinverse(q)), inv_period}; }
LINS_HD V3 to_end_point(const ToEnd& c, double x, double y, double z, float w) {
  const float frac = w - (float)(int)w;
  const double s = c.inv_period * (double)frac;
  const V3 p1 = qrot(axis2quat_fast(s * c.phi), V3{x, y, z}) + s * c.t;
  const V3 d = p1 - c.t;
  return V3{c.rinv.m[0] * d.x + c.rinv.m[1] * d.y + c.rinv.m[2] * d.z, c.rinv.m[3] * d.x + c.rinv.m[4] * d.y + c.rinv.m[5] * d.z,
            c.rinv.m[6] * d.x + c.rinv.m[7] * d.y + c.rinv.m[8] * d.z};
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
// phi = Quat2axis(q) and Gt = Rinvleft(-phi)^T in one go for a small rotation (w > 0, tan(|phi| / 2) <= 1/8).
// With h = |phi| / 2, t = tan h = |qv| / w, z = t^2 and the series A, Q above:
//   phi = qv * 2 A / w,   h cot h = A,   (1 - h cot h) / |phi|^2 = Q / (4 A^2),   h / |phi| = 1/2, so
//   Rinvleft(-phi) = s I + (1 - s) a a^T - h [a]x  with a = -phi / |phi|, s = h cot h      (MU:304-321)
//                  = A I + (Q / w^2) qv qv^T + (A / w) [qv]x
// — no square root, one division, two short fma chains.  Returns false (outputs untouched) outside that range.
LINS_HD bool phi_and_gt_small(const Q4& q, V3& phi, M3& Gt) {
  const double m2 = q.x * q.x + q.y * q.y + q.z * q.z;
  if (!(q.w > 0.0 && m2 >= 1e-20 && m2 <= kFastTanSq * (q.w * q.w))) return false;
  const double rw = 1.0 / q.w;
  double A, Q;
  lins_atanc_small((m2 * rw) * rw, A, Q);
  const double k = (2.0 * A) * rw, c1 = (Q * rw) * rw, c2 = A * rw;
  phi = V3{q.x * k, q.y * k, q.z * k};
  const double v[3] = {q.x, q.y, q.z};
  const M3 sk = skew(V3{q.x, q.y, q.z});
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Gt.m[j * 3 + i] = ((i == j ? A : 0.0) + c1 * v[i] * v[j]) + c2 * sk.m[i * 3 + j];
  return true;
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
