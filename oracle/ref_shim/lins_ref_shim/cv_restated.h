// lins_ref_shim/cv_restated.h — the four OpenCV routines lidar_mapping_node.cpp's scan-to-map optimisation calls,
// RESTATED (OpenCV is not on this machine; nothing here is copied from it):
//   cv::eigen (symmetric)     cyclic Jacobi in f32, eigenvalues descending, eigenvectors as ROWS
//   cv::solve(DECOMP_QR)      Householder QR in f32 (least squares when there are more rows than columns)
//   cv::Mat::inv()            Gauss-Jordan with partial pivoting in f32
//   cv::Mat * cv::Mat         products accumulated in f64 over the inner index in ascending order, rounded to f32 once
//                             (OpenCV's small-matrix f32 GEMM accumulates in double; the order inside its blocked kernels
//                             for tall operands is not knowable from here — stated, not pinned)
// Shared by the stand-in <opencv2/opencv.hpp> (through which the reference's own text reaches them, oracle/_ref) and by
// oracle/map_oracle.cpp (the restatement the device kernels are compared with): the two then differ only in what is
// the reference's own — the glue of LM:1351-1652 — which is what tests/test_ref.py pins.  The product's versions
// (csrc/map_math.h, csrc/lm_math.h) are written separately to the same operation sequences.
#ifndef LINS_REF_SHIM_CV_RESTATED_
#define LINS_REF_SHIM_CV_RESTATED_
#include <cmath>
#include <cstring>
#include <vector>
namespace lins_cvr {

// a: n x n symmetric, row-major, destroyed; w[n] descending; V n x n, row i = eigenvector i
inline void jacobi_eig(float* a, int N, float* w, float* V) {
  std::vector<float> v((size_t)N * N);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) v[i * N + j] = i == j ? 1.f : 0.f;
  for (int sweep = 0; sweep < 60; ++sweep) {
    float off = 0.f, diag = 0.f;
    for (int i = 0; i < N; ++i) {
      diag += std::fabs(a[i * N + i]);
      for (int j = i + 1; j < N; ++j) off += std::fabs(a[i * N + j]);
    }
    if (!(off > 1e-12f * diag)) break;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        const float apq = a[p * N + q];
        if (apq == 0.f) continue;
        const float theta = (a[q * N + q] - a[p * N + p]) / (2.f * apq);
        const float t = (theta >= 0.f ? 1.f : -1.f) / (std::fabs(theta) + std::sqrt(theta * theta + 1.f));
        const float c = 1.f / std::sqrt(t * t + 1.f), s = t * c;
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
  std::vector<int> ord(N);
  for (int i = 0; i < N; ++i) ord[i] = i;
  for (int i = 1; i < N; ++i)  // insertion sort, descending, stable
    for (int j = i; j > 0 && a[ord[j] * N + ord[j]] > a[ord[j - 1] * N + ord[j - 1]]; --j) {
      const int tmp = ord[j];
      ord[j] = ord[j - 1], ord[j - 1] = tmp;
    }
  for (int i = 0; i < N; ++i) {
    w[i] = a[ord[i] * N + ord[i]];
    for (int k = 0; k < N; ++k) V[i * N + k] = v[k * N + ord[i]];
  }
}

// a: M x N row-major (M >= N), destroyed; b[M] destroyed; x[N]
inline void qr_solve(float* a, int M, int N, float* b, float* x) {
  std::vector<float> v(M);
  for (int k = 0; k < N; ++k) {
    float nrm2 = 0.f;
    for (int i = k; i < M; ++i) nrm2 += a[i * N + k] * a[i * N + k];
    const float nrm = std::sqrt(nrm2);
    if (nrm == 0.f) continue;
    const float alpha = a[k * N + k] >= 0.f ? -nrm : nrm;
    for (int i = 0; i < M; ++i) v[i] = i >= k ? a[i * N + k] : 0.f;
    v[k] -= alpha;
    float vv = 0.f;
    for (int i = k; i < M; ++i) vv += v[i] * v[i];
    if (vv == 0.f) continue;
    for (int j = k; j < N; ++j) {
      float s = 0.f;
      for (int i = k; i < M; ++i) s += v[i] * a[i * N + j];
      s = 2.f * s / vv;
      for (int i = k; i < M; ++i) a[i * N + j] -= s * v[i];
    }
    float s = 0.f;
    for (int i = k; i < M; ++i) s += v[i] * b[i];
    s = 2.f * s / vv;
    for (int i = k; i < M; ++i) b[i] -= s * v[i];
  }
  for (int i = N - 1; i >= 0; --i) {
    float s = b[i];
    for (int j = i + 1; j < N; ++j) s -= a[i * N + j] * x[j];
    x[i] = s / a[i * N + i];
  }
}

inline void inv(const float* A, int n, float* out) {  // Gauss-Jordan, partial pivoting
  std::vector<float> a(A, A + (size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) out[i * n + j] = i == j ? 1.f : 0.f;
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(a[i * n + k]) > std::fabs(a[p * n + k])) p = i;
    if (p != k)
      for (int j = 0; j < n; ++j) {
        float t = a[k * n + j];
        a[k * n + j] = a[p * n + j], a[p * n + j] = t;
        t = out[k * n + j];
        out[k * n + j] = out[p * n + j], out[p * n + j] = t;
      }
    const float d = a[k * n + k];
    for (int j = 0; j < n; ++j) a[k * n + j] /= d, out[k * n + j] /= d;
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const float f = a[i * n + k];
      for (int j = 0; j < n; ++j) a[i * n + j] -= f * a[k * n + j], out[i * n + j] -= f * out[k * n + j];
    }
  }
}

// out (r x c) = A (r x k) * B (k x c), every entry accumulated in f64 over the inner index, rounded once
inline void matmul(const float* A, int r, int k, const float* B, int c, float* out) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int t = 0; t < k; ++t) s += (double)A[i * k + t] * (double)B[t * c + j];
      out[i * c + j] = (float)s;
    }
}

}  // namespace lins_cvr
#endif
