// lm_wave.h — LMOptimization's 6x6 step (lm_math.h: lm_step_from_sums, LM:1583-1632) spread over one wave.
//
// lm_math.h is the definition (shared with the host): one thread walks the Householder QR, and on round 0 the cyclic
// Jacobi eigen-decomposition, the Gauss-Jordan inverse and the projection product — ~30 k dependent instructions, 100 us
// on a device that runs one lane of one wave for it (round 2 / 3: map_lm_kernel, one thread per problem).  Here lane
// 6 i + j of a wave holds element (i, j) of each 6 x 6 matrix: a Householder reflection updates every column at once, a
// Jacobi rotation its two columns, its two rows and the two eigenvector columns in three steps, an elimination step of
// the inverse every row at once.  Values cross lanes by ds_bpermute (per-lane source) or v_readlane (uniform source);
// every sum runs over its index in the scalar loops' order and every product / quotient / square root is the same f32
// (or, where lm_math.h says so, f64) operation on the same operands, so the results are the definition's bits —
// tests/test_gpu_math.py::test_lm_step_over_a_wave_is_bit_identical_to_the_one_thread_definition.
#pragma once

#include <hip/hip_runtime.h>

#include "lm_math.h"

namespace lins {

__device__ __forceinline__ float lmw_rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float lmw_sel6(const float (&v)[6], int i) {
  return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : (i == 3 ? v[3] : (i == 4 ? v[4] : v[5]))));
}

// lm_qr6: a = this lane's element (i, j) of A (lanes 0..35), b / x uniform
__device__ __forceinline__ void wave_lm_qr6(float a, float (&b)[6], int i, int j, bool in_mat, float (&x)[6]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float col[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) col[r] = lmw_rl(a, r * 6 + k);
    float nrm2 = 0.f;
#pragma unroll
    for (int r = k; r < 6; ++r) nrm2 += col[r] * col[r];
    const float nrm = sqrtf(nrm2);
    if (nrm == 0.f) continue;  // (uniform)
    const float alpha = col[k] >= 0.f ? -nrm : nrm;
    float v[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) v[r] = r >= k ? col[r] : 0.f;
    v[k] -= alpha;
    float vv = 0.f;
#pragma unroll
    for (int r = k; r < 6; ++r) vv += v[r] * v[r];
    if (vv == 0.f) continue;  // (uniform)
    float s = 0.f;  // this lane's column j: sum over the rows k..5 in order
#pragma unroll
    for (int r = k; r < 6; ++r) s += v[r] * __shfl(a, r * 6 + j);
    s = 2.f * s / vv;
    if (in_mat && j >= k && i >= k) a -= s * lmw_sel6(v, i);
    float sb = 0.f;
#pragma unroll
    for (int r = k; r < 6; ++r) sb += v[r] * b[r];
    sb = 2.f * sb / vv;
#pragma unroll
    for (int r = k; r < 6; ++r) b[r] -= sb * v[r];
  }
#pragma unroll
  for (int r = 5; r >= 0; --r) {
    float s = b[r];
#pragma unroll
    for (int c = r + 1; c < 6; ++c) s -= lmw_rl(a, r * 6 + c) * x[c];
    x[r] = s / lmw_rl(a, r * 6 + r);
  }
}

// lm_eig6: a = element (i, j) of the symmetric matrix; out: w uniform (descending), V element (i, k) = eigenvector i
__device__ __forceinline__ void wave_lm_eig6(float a, int i, int j, float (&w)[6], float& V) {
  float vm = i == j ? 1.f : 0.f;
#pragma unroll 1
  for (int sweep = 0; sweep < 60; ++sweep) {
    float off = 0.f, diag = 0.f;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      diag += fabsf(lmw_rl(a, r * 6 + r));
#pragma unroll
      for (int c = r + 1; c < 6; ++c) off += fabsf(lmw_rl(a, r * 6 + c));
    }
    if (!(off > 1e-12f * diag)) break;
#pragma unroll 1
    for (int p = 0; p < 6; ++p)
#pragma unroll 1
      for (int q = p + 1; q < 6; ++q) {
        const float apq = lmw_rl(a, p * 6 + q);
        if (apq == 0.f) continue;  // (uniform)
        const float theta = (lmw_rl(a, q * 6 + q) - lmw_rl(a, p * 6 + p)) / (2.f * apq);
        const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
        const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
        {  // columns p, q of a: rows are independent
          const float x = __shfl(a, i * 6 + p), y = __shfl(a, i * 6 + q);
          if (j == p)
            a = c * x - s * y;
          else if (j == q)
            a = s * x + c * y;
        }
        {  // rows p, q of a (after the columns, as in the scalar loops)
          const float x = __shfl(a, p * 6 + j), y = __shfl(a, q * 6 + j);
          if (i == p)
            a = c * x - s * y;
          else if (i == q)
            a = s * x + c * y;
        }
        {  // columns p, q of the eigenvector matrix
          const float x = __shfl(vm, i * 6 + p), y = __shfl(vm, i * 6 + q);
          if (j == p)
            vm = c * x - s * y;
          else if (j == q)
            vm = s * x + c * y;
        }
      }
  }
  float d[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) d[r] = lmw_rl(a, r * 6 + r);
  int ord[6] = {0, 1, 2, 3, 4, 5};  // (uniform; the definition's insertion sort, descending, stable)
#pragma unroll
  for (int r = 1; r < 6; ++r)
#pragma unroll
    for (int c = r; c > 0; --c) {
      const float dc = ord[c] == 0 ? d[0] : (ord[c] == 1 ? d[1] : (ord[c] == 2 ? d[2] : (ord[c] == 3 ? d[3] : (ord[c] == 4 ? d[4] : d[5]))));
      const int oc1 = ord[c - 1];
      const float dp = oc1 == 0 ? d[0] : (oc1 == 1 ? d[1] : (oc1 == 2 ? d[2] : (oc1 == 3 ? d[3] : (oc1 == 4 ? d[4] : d[5]))));
      if (!(dc > dp)) break;
      const int tmp = ord[c];
      ord[c] = ord[c - 1], ord[c - 1] = tmp;
    }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int o = ord[r];
    w[r] = o == 0 ? d[0] : (o == 1 ? d[1] : (o == 2 ? d[2] : (o == 3 ? d[3] : (o == 4 ? d[4] : d[5]))));
  }
  const int oi = i == 0 ? ord[0] : (i == 1 ? ord[1] : (i == 2 ? ord[2] : (i == 3 ? ord[3] : (i == 4 ? ord[4] : ord[5]))));
  V = __shfl(vm, j * 6 + oi);  // V[i][k] = v[k][ord[i]] (k = this lane's j)
}

// lm_inv6: Gauss-Jordan with partial pivoting; a = element (i, j) of A, returns element (i, j) of the inverse
__device__ __forceinline__ float wave_lm_inv6(float a, int i, int j) {
  float inv = i == j ? 1.f : 0.f;
#pragma unroll 1
  for (int k = 0; k < 6; ++k) {
    int p = k;  // (uniform: first maximum of |a[r][k]|, r >= k)
    float best = fabsf(lmw_rl(a, k * 6 + k));
    for (int r = k + 1; r < 6; ++r) {
      const float v = fabsf(lmw_rl(a, r * 6 + k));
      if (v > best) best = v, p = r;
    }
    if (p != k) {  // (uniform) swap rows k and p
      const int src = i == k ? p * 6 + j : (i == p ? k * 6 + j : i * 6 + j);
      a = __shfl(a, src), inv = __shfl(inv, src);
    }
    const float d = lmw_rl(a, k * 6 + k);
    if (i == k) a /= d, inv /= d;
    const float f = __shfl(a, i * 6 + k), akj = __shfl(a, k * 6 + j), ikj = __shfl(inv, k * 6 + j);
    if (i != k) a -= f * akj, inv -= f * ikj;
  }
  return inv;
}

// lm_step_from_sums over a wave.  sums: uniform pointer to the 28 f64 sums (global or LDS); T (6, uniform registers),
// carry: this problem's LmCarry in global memory.  Every lane returns the same flag / T.  lane = 0..63.
__device__ __forceinline__ bool wave_lm_step_from_sums(const double* sums, int iter, float (&T)[6], LmCarry* carry, int lane,
                                                       int& degenerate_out) {
  degenerate_out = carry->degenerate;
  if ((int)sums[27] < 50) return false;  // LM:1530-1532
  const bool in_mat = lane < 36;
  const int l = in_mat ? lane : 35, i = l / 6, j = l % 6;
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const float A = (float)sums[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];  // upper triangle, row-major: (lo, hi)
  float B[6], X[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) B[r] = (float)sums[21 + r], X[r] = 0.f;
  wave_lm_qr6(A, B, i, j, in_mat, X);
  int degenerate = degenerate_out;
  float P = 0.f;
  if (iter == 0) {
    float E[6], V;
    wave_lm_eig6(A, i, j, E, V);
    int m = 6;  // rows m..5 of V2 are zeroed (eigenvalues below 100, from the smallest up)
    degenerate = 0;
#pragma unroll
    for (int r = 5; r >= 0; --r) {
      if (E[r] < 100)
        m = r, degenerate = 1;
      else
        break;
    }
    const float V2 = i >= m ? 0.f : V;
    const float Vi = wave_lm_inv6(V, i, j);
    double s = 0.0;  // matP = matV.inv() * matV2, accumulated in f64 over the inner index and rounded once
#pragma unroll
    for (int k = 0; k < 6; ++k) s += (double)__shfl(Vi, i * 6 + k) * (double)__shfl(V2, k * 6 + j);
    P = (float)s;
    if (in_mat) carry->P[lane] = P;
    if (lane == 0) carry->degenerate = degenerate;
    degenerate_out = degenerate;
  } else if (degenerate) {
    P = carry->P[l];
  }
  if (degenerate) {
    float X2[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) X2[r] = X[r];
#pragma unroll
    for (int r = 0; r < 6; ++r) {  // matX = matP * matX2
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += (double)lmw_rl(P, r * 6 + k) * (double)X2[k];
      X[r] = (float)s;
    }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) T[r] += X[r];
  double r2 = 0.0, t2 = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double d = (double)(float)(X[r] * 57.29578f), mm = (double)(float)(X[3 + r] * 100);
    r2 += d * d, t2 += mm * mm;
  }
  const float deltaR = (float)sqrt(r2), deltaT = (float)sqrt(t2);
  return deltaR < 0.05 && deltaT < 0.05;
}

}  // namespace lins
