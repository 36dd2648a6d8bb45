// lm_math.h — LMOptimization's 6x6 step (src/lidar_mapping_node.cpp "LM" 1583-1632) in f32 with FIXED operation
// sequences, shared by the device step kernel (map_kernels.hip) and host code: cyclic-Jacobi eigen-decomposition
// (cv::eigen restated: eigenvalues descending, rows = eigenvectors), Householder QR solve (cv::solve(DECOMP_QR)),
// Gauss-Jordan inverse (cv::Mat::inv), the degeneracy projection of round 0 (LM:1589-1614) and the stop rule.
#pragma once

#include <math.h>

#include "lins_math.h"
#include "map_math.h"

namespace lins {

struct LmCarry {  // what LMOptimization keeps between the rounds of one scan2MapOptimization
  int degenerate;
  float P[36];
};

LINS_HD void lm_eig6(float* a, float* w, float* V) {
  const int N = 6;
  float v[36];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) v[i * N + j] = i == j ? 1.f : 0.f;
  for (int sweep = 0; sweep < 60; ++sweep) {
    float off = 0.f, diag = 0.f;
    for (int i = 0; i < N; ++i) {
      diag += fabsf(a[i * N + i]);
      for (int j = i + 1; j < N; ++j) off += fabsf(a[i * N + j]);
    }
    if (!(off > 1e-12f * diag)) break;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        const float apq = a[p * N + q];
        if (apq == 0.f) continue;
        const float theta = (a[q * N + q] - a[p * N + p]) / (2.f * apq);
        const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
        const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
        for (int k = 0; k < N; ++k) {
          const float x = a[k * N + p], y = a[k * N + q];
          a[k * N + p] = c * x - s * y, a[k * N + q] = s * x + c * y;
        }
        for (int k = 0; k < N; ++k) {
          const float x = a[p * N + k], y = a[q * N + k];
          a[p * N + k] = c * x - s * y, a[q * N + k] = s * x + c * y;
        }
        for (int k = 0; k < N; ++k) {
          const float x = v[k * N + p], y = v[k * N + q];
          v[k * N + p] = c * x - s * y, v[k * N + q] = s * x + c * y;
        }
      }
  }
  int ord[6] = {0, 1, 2, 3, 4, 5};
  for (int i = 1; i < N; ++i)
    for (int j = i; j > 0 && a[ord[j] * N + ord[j]] > a[ord[j - 1] * N + ord[j - 1]]; --j) {
      const int tmp = ord[j];
      ord[j] = ord[j - 1], ord[j - 1] = tmp;
    }
  for (int i = 0; i < N; ++i) {
    w[i] = a[ord[i] * N + ord[i]];
    for (int k = 0; k < N; ++k) V[i * N + k] = v[k * N + ord[i]];
  }
}

LINS_HD void lm_qr6(float* a, float* b, float* x) {  // (a, b destroyed)
  const int N = 6;
  for (int k = 0; k < N; ++k) {
    float nrm2 = 0.f;
    for (int i = k; i < N; ++i) nrm2 += a[i * N + k] * a[i * N + k];
    const float nrm = sqrtf(nrm2);
    if (nrm == 0.f) continue;
    const float alpha = a[k * N + k] >= 0.f ? -nrm : nrm;
    float v[6];
    for (int i = 0; i < N; ++i) v[i] = i >= k ? a[i * N + k] : 0.f;
    v[k] -= alpha;
    float vv = 0.f;
    for (int i = k; i < N; ++i) vv += v[i] * v[i];
    if (vv == 0.f) continue;
    for (int j = k; j < N; ++j) {
      float s = 0.f;
      for (int i = k; i < N; ++i) s += v[i] * a[i * N + j];
      s = 2.f * s / vv;
      for (int i = k; i < N; ++i) a[i * N + j] -= s * v[i];
    }
    float s = 0.f;
    for (int i = k; i < N; ++i) s += v[i] * b[i];
    s = 2.f * s / vv;
    for (int i = k; i < N; ++i) b[i] -= s * v[i];
  }
  for (int i = N - 1; i >= 0; --i) {
    float s = b[i];
    for (int j = i + 1; j < N; ++j) s -= a[i * N + j] * x[j];
    x[i] = s / a[i * N + i];
  }
}

LINS_HD void lm_inv6(const float* A, float* inv) {  // Gauss-Jordan, partial pivoting
  const int n = 6;
  float a[36];
  for (int i = 0; i < 36; ++i) a[i] = A[i];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) inv[i * n + j] = i == j ? 1.f : 0.f;
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (fabsf(a[i * n + k]) > fabsf(a[p * n + k])) p = i;
    if (p != k)
      for (int j = 0; j < n; ++j) {
        float t = a[k * n + j];
        a[k * n + j] = a[p * n + j], a[p * n + j] = t;
        t = inv[k * n + j];
        inv[k * n + j] = inv[p * n + j], inv[p * n + j] = t;
      }
    const float d = a[k * n + k];
    for (int j = 0; j < n; ++j) a[k * n + j] /= d, inv[k * n + j] /= d;
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const float f = a[i * n + k];
      for (int j = 0; j < n; ++j) a[i * n + j] -= f * a[k * n + j], inv[i * n + j] -= f * inv[k * n + j];
    }
  }
}

// the step from the 28 sums (upper triangle of A^T A, A^T b, row count); true = converged (LM:1583-1632)
LINS_HD bool lm_step_from_sums(const double* sums, int iter, float* T, LmCarry& st) {
  if ((int)sums[27] < 50) return false;  // LM:1530-1532
  float A[36], B[6], X[6], Aq[36], Bq[6];
  int t = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) A[i * 6 + j] = A[j * 6 + i] = (float)sums[t++];
  for (int i = 0; i < 6; ++i) B[i] = (float)sums[21 + i];
  for (int i = 0; i < 36; ++i) Aq[i] = A[i];
  for (int i = 0; i < 6; ++i) Bq[i] = B[i];
  lm_qr6(Aq, Bq, X);
  if (iter == 0) {
    float Ae[36], E[6], V[36], V2[36], Vi[36];
    for (int i = 0; i < 36; ++i) Ae[i] = A[i];
    lm_eig6(Ae, E, V);
    for (int i = 0; i < 36; ++i) V2[i] = V[i];
    st.degenerate = 0;
    for (int i = 5; i >= 0; i--) {
      if (E[i] < 100) {
        for (int j = 0; j < 6; j++) V2[i * 6 + j] = 0;
        st.degenerate = 1;
      } else {
        break;
      }
    }
    lm_inv6(V, Vi);
    for (int i = 0; i < 6; ++i)  // matP = matV.inv() * matV2: like every Mat product of the path, accumulated in f64 over the
      for (int j = 0; j < 6; ++j) {  // inner index and rounded once (OpenCV's small f32 GEMM accumulates in double)
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += (double)Vi[i * 6 + k] * (double)V2[k * 6 + j];
        st.P[i * 6 + j] = (float)s;
      }
  }
  if (st.degenerate) {
    float X2[6];
    for (int i = 0; i < 6; ++i) X2[i] = X[i];
    for (int i = 0; i < 6; ++i) {  // matX = matP * matX2
      double s = 0.0;
      for (int k = 0; k < 6; ++k) s += (double)st.P[i * 6 + k] * (double)X2[k];
      X[i] = (float)s;
    }
  }
  for (int i = 0; i < 6; ++i) T[i] += X[i];
  // sqrt(pow(rad2deg(x), 2) + ...): pow(float, int) is evaluated in double (LM:1625-1630)
  double r2 = 0.0, t2 = 0.0;
  for (int i = 0; i < 3; ++i) {
    const double d = (double)(float)(X[i] * 57.29578f), m = (double)(float)(X[3 + i] * 100);
    r2 += d * d, t2 += m * m;
  }
  const float deltaR = (float)sqrt(r2), deltaT = (float)sqrt(t2);
  return deltaR < 0.05 && deltaT < 0.05;
}

struct MapRoundParams {  // (layout of map_kernels.hip's MapRound)
  MapAssoc as;
  MapTrig tg;
  float pad;
};
LINS_HD MapRoundParams lm_make_round(const float* T) {
  MapRoundParams r;
  r.as.cRoll = cosf(T[0]), r.as.sRoll = sinf(T[0]);
  r.as.cPitch = cosf(T[1]), r.as.sPitch = sinf(T[1]);
  r.as.cYaw = cosf(T[2]), r.as.sYaw = sinf(T[2]);
  r.as.tX = T[3], r.as.tY = T[4], r.as.tZ = T[5];
  r.tg.srx = sinf(T[0]), r.tg.crx = cosf(T[0]);
  r.tg.sry = sinf(T[1]), r.tg.cry = cosf(T[1]);
  r.tg.srz = sinf(T[2]), r.tg.crz = cosf(T[2]);
  r.pad = 0.f;
  return r;
}

}  // namespace lins
