// map_math.h — f32 arithmetic of the scan-to-map row, host/device:
//   MapAssoc / map_associate   updatePointAssociateToMapSinCos + pointAssociateToMap   LM:579-607
//   map_corner_fit             cornerOptimization after the 5-NN                         LM:1360-1450
//   map_surf_fit               surfOptimization after the 5-NN                           LM:1464-1518
//   map_lm_row                 one row of LMOptimization's matA / matB                   LM:1543-1581
// ("LM" = /root/reference/lins/src/lidar_mapping_node.cpp).  Everything is f32 in the reference's
// expression order; cv::eigen is a cyclic Jacobi, cv::solve(DECOMP_QR) a Householder QR, both with
// fixed operation sequences (see include/lins_map.h).  Trigonometry of the 6-DoF transform is done
// once per round on the host (the same libm calls on every path) and handed over as MapAssoc /
// MapTrig, so that no device transcendental enters the per-point arithmetic.
#pragma once
#include <math.h>

#include "../../include/lins_map.h"
#include "lins_math.h"

namespace lins {

struct MapAssoc {
  float cRoll, sRoll, cPitch, sPitch, cYaw, sYaw, tX, tY, tZ;
};
struct MapTrig {  // LMOptimization's srx .. crz (LM:1524-1529)
  float srx, crx, sry, cry, srz, crz;
};

LINS_HD void map_associate(const MapAssoc& a, float px, float py, float pz, float& ox, float& oy, float& oz) {
  const float x1 = a.cYaw * px - a.sYaw * py;
  const float y1 = a.sYaw * px + a.cYaw * py;
  const float z1 = pz;
  const float x2 = x1;
  const float y2 = a.cRoll * y1 - a.sRoll * z1;
  const float z2 = a.sRoll * y1 + a.cRoll * z1;
  ox = a.cPitch * x2 + a.sPitch * z2 + a.tX;
  oy = y2 + a.tY;
  oz = -a.sPitch * x2 + a.cPitch * z2 + a.tZ;
}

// symmetric 3x3 eigen-decomposition, cyclic Jacobi in f32: d descending, rows of v = eigenvectors
LINS_HD void map_eig3(float (&a)[9], float (&d)[3], float (&v)[9]) {
  float u[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  for (int sweep = 0; sweep < 60; ++sweep) {
    float off = 0.f, diag = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      diag += fabsf(a[i * 3 + i]);
#pragma unroll
      for (int j = i + 1; j < 3; ++j) off += fabsf(a[i * 3 + j]);
    }
    if (!(off > 1e-12f * diag)) break;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const float apq = a[p * 3 + q];
        if (apq == 0.f) continue;
        const float theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.f * apq);
        const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
        const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float x = a[k * 3 + p], y = a[k * 3 + q];
          a[k * 3 + p] = c * x - s * y, a[k * 3 + q] = s * x + c * y;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float x = a[p * 3 + k], y = a[q * 3 + k];
          a[p * 3 + k] = c * x - s * y, a[q * 3 + k] = s * x + c * y;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float x = u[k * 3 + p], y = u[k * 3 + q];
          u[k * 3 + p] = c * x - s * y, u[k * 3 + q] = s * x + c * y;
        }
      }
  }
  int ord[3] = {0, 1, 2};
#pragma unroll
  for (int i = 1; i < 3; ++i)
#pragma unroll
    for (int j = i; j > 0 && a[ord[j] * 3 + ord[j]] > a[ord[j - 1] * 3 + ord[j - 1]]; --j) {
      const int tmp = ord[j];
      ord[j] = ord[j - 1], ord[j - 1] = tmp;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    d[i] = a[ord[i] * 3 + ord[i]];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[i * 3 + k] = u[k * 3 + ord[i]];
  }
}

// least squares of the 5x3 system A x = b by Householder QR in f32 (A, b destroyed)
#ifdef LINS_MAP_QR_NOINLINE  // (canary experiments, tools/repro/README.md)
__attribute__((noinline))
#endif
LINS_HD void map_qr_5x3(float (&a)[15], float (&b)[5], float (&x)[3]) {
  constexpr int M = 5, N = 3;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float nrm2 = 0.f;
#pragma unroll
    for (int i = k; i < M; ++i) nrm2 += a[i * N + k] * a[i * N + k];
    const float nrm = sqrtf(nrm2);
    const float alpha = a[k * N + k] >= 0.f ? -nrm : nrm;
    float v[M];
#pragma unroll
    for (int i = 0; i < M; ++i) v[i] = i >= k ? a[i * N + k] : 0.f;
    v[k] -= alpha;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LINS_MAP_QR_NO_GUARD)
    // Guard against an SLP-vectoriser miscompile of ROCm 7.2's clang (root-caused in round 6, tools/repro/README.md,
    // tools/repro/slp_surf_fit_gather.hip): below the pivot v[i] IS a[i * N + k], so the products of the loop over j start with
    // the pair { a_ik * a_ik, a_ik * a_i,k+1 }, whose first member is also a term of nrm2 above and already sits in another
    // vectorised tree; for the last row the pass then emits  fmul <2 x float> %row, %row  — (a_ik^2, a_i,k+1^2) — where
    // (a_ik * a_ik, a_ik * a_i,k+1) was asked for: the column right of the pivot is updated with the wrong reflection (plane
    // normals of y-walls came out tilted: 8 scan-to-map tests).  An opaque copy of the vector breaks the identity of the
    // two values, not the arithmetic; it replaces the file-wide -fno-slp-vectorize of rounds 3-5.
#pragma unroll
    for (int i = 0; i < M; ++i) asm volatile("" : "+v"(v[i]));
#endif
    float vv = 0.f;
#pragma unroll
    for (int i = k; i < M; ++i) vv += v[i] * v[i];
    if (nrm != 0.f && vv != 0.f) {  // (a zero column is left alone)
#pragma unroll
      for (int j = k; j < N; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = k; i < M; ++i) s += v[i] * a[i * N + j];
        s = 2.f * s / vv;
#pragma unroll
        for (int i = k; i < M; ++i) a[i * N + j] -= s * v[i];
      }
      float s = 0.f;
#pragma unroll
      for (int i = k; i < M; ++i) s += v[i] * b[i];
      s = 2.f * s / vv;
#pragma unroll
      for (int i = k; i < M; ++i) b[i] -= s * v[i];
    }
  }
  x[2] = b[2] / a[8];
  x[1] = (b[1] - a[5] * x[2]) / a[4];
  x[0] = ((b[0] - a[1] * x[1]) - a[2] * x[2]) / a[0];
}

// px/py/pz: the five neighbours in (distance, index) order; returns accepted, fills coeff
LINS_HD int map_corner_fit(const float (&px)[5], const float (&py)[5], const float (&pz)[5], float x0, float y0, float z0,
                           float (&coeff)[4]) {
  coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0.f;
  float cx = 0, cy = 0, cz = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) cx += px[j], cy += py[j], cz += pz[j];
  cx /= 5, cy /= 5, cz /= 5;
  float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const float ax = px[j] - cx, ay = py[j] - cy, az = pz[j] - cz;
    a11 += ax * ax, a12 += ax * ay, a13 += ax * az, a22 += ay * ay, a23 += ay * az, a33 += az * az;
  }
  a11 /= 5, a12 /= 5, a13 /= 5, a22 /= 5, a23 /= 5, a33 /= 5;
  float A[9] = {a11, a12, a13, a12, a22, a23, a13, a23, a33}, D[3], V[9];
  map_eig3(A, D, V);
  if (!(D[0] > 3 * D[1])) return 0;
  const float x1 = (float)(cx + 0.1 * V[0]), y1 = (float)(cy + 0.1 * V[1]), z1 = (float)(cz + 0.1 * V[2]);
  const float x2 = (float)(cx - 0.1 * V[0]), y2 = (float)(cy - 0.1 * V[1]), z2 = (float)(cz - 0.1 * V[2]);
  const float a012 = sqrtf(((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                           ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                           ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)));
  const float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
  const float la = ((y1 - y2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                    (z1 - z2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))) /
                   a012 / l12;
  const float lb = -((x1 - x2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) -
                     (z1 - z2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) /
                   a012 / l12;
  const float lc = -((x1 - x2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                     (y1 - y2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) /
                   a012 / l12;
  const float ld2 = a012 / l12;
  const float s = (float)(1 - 0.9 * fabsf(ld2));
  coeff[0] = s * la, coeff[1] = s * lb, coeff[2] = s * lc, coeff[3] = s * ld2;
  return s > 0.1 ? 1 : 0;
}

#ifdef LINS_MAP_FIT_NOINLINE
__attribute__((noinline))
#endif
LINS_HD int map_surf_fit(const float (&px)[5], const float (&py)[5], const float (&pz)[5], float sx, float sy, float sz,
                         float (&coeff)[4]) {
  coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0.f;
  float A[15], B[5] = {-1, -1, -1, -1, -1}, X[3];
#pragma unroll
  for (int j = 0; j < 5; j++) A[j * 3 + 0] = px[j], A[j * 3 + 1] = py[j], A[j * 3 + 2] = pz[j];
  map_qr_5x3(A, B, X);
  float pa = X[0], pb = X[1], pc = X[2], pd = 1;
  const float ps = sqrtf(pa * pa + pb * pb + pc * pc);
  pa /= ps, pb /= ps, pc /= ps, pd /= ps;
#pragma unroll
  for (int j = 0; j < 5; j++)
    if (fabsf(pa * px[j] + pb * py[j] + pc * pz[j] + pd) > 0.2) return 0;
  const float pd2 = pa * sx + pb * sy + pc * sz + pd;
  const float s = (float)(1 - 0.9 * fabsf(pd2) / sqrtf(sqrtf(sx * sx + sy * sy + sz * sz)));
  coeff[0] = s * pa, coeff[1] = s * pb, coeff[2] = s * pc, coeff[3] = s * pd2;
  return s > 0.1 ? 1 : 0;
}

// row (arx, ary, arz, cx, cy, cz | -ci) of LMOptimization for one selected point
LINS_HD void map_lm_row(const MapTrig& g, float ox, float oy, float oz, const float (&c)[4], float (&row)[6], float& b) {
  const float srx = g.srx, crx = g.crx, sry = g.sry, cry = g.cry, srz = g.srz, crz = g.crz;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float arx = (crx * sry * srz * ox + crx * crz * sry * oy - srx * sry * oz) * cx +
                    (-srx * srz * ox - crz * srx * oy - crx * oz) * cy +
                    (crx * cry * srz * ox + crx * cry * crz * oy - cry * srx * oz) * cz;
  const float ary = ((cry * srx * srz - crz * sry) * ox + (sry * srz + cry * crz * srx) * oy + crx * cry * oz) * cx +
                    ((-cry * crz - srx * sry * srz) * ox + (cry * srz - crz * srx * sry) * oy - crx * sry * oz) * cz;
  const float arz = ((crz * srx * sry - cry * srz) * ox + (-cry * crz - srx * sry * srz) * oy) * cx +
                    (crx * crz * ox - crx * srz * oy) * cy +
                    ((sry * srz + cry * crz * srx) * ox + (crz * sry - cry * srx * srz) * oy) * cz;
  row[0] = arx, row[1] = ary, row[2] = arz, row[3] = cx, row[4] = cy, row[5] = cz;
  b = -c[3];
}

}  // namespace lins
