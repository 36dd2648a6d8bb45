// ieskf_joseph.h — the Joseph covariance update after the iterated update (SE:594-598), in the push-through form of
// with N = sigma^2 I + A P_SS (6 x 6, S = {0,1,2,6,7,8}),
//   Y = N^-1 A,  KH = P[:,S] Y E_S^T,  K R K^T = sigma^2 P[:,S] (Y N^-T) P[:,S]^T,
//   P+ = (I - KH) P (I - KH)^T + K R K^T, symmetrised (enforceSymmetry).  A diverged scan keeps its Pk_ (SE:592).
// Called by ieskf_joseph_kernel (ieskf_kernels.hip: its own launch, 128 threads per scan, after every update kernel).
// Written against a block size so that an update kernel can apply it in its own epilogue, in the LDS its search grid
// no longer needs; measured for the batch kernel in round 2 and NOT kept: the 512-thread workgroup spends 27 us per
// launch in the block-wide 6 x 12 eliminations (fourteen barriers each) to save an 18 us kernel and a launch
// (step 0.718 -> 0.723 ms).  Every element goes through the same operations in the same order whatever the block size.
#pragma once

#include <hip/hip_runtime.h>

#include "ieskf_device.h"

namespace lins {

// ---------------------------------------------------------------------------
// 6 x 6 pivoted elimination in LDS, cooperative over the block.
// aug = [N | B] row-major 6 x nc; sol (nrhs = nc - 6 columns, row-major 6 x nrhs) = N^-1 B.
// Pivot rows are chosen per column among the not-yet-used rows (implicit row
// exchange).  Every thread of the block must call this (it contains barriers).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void block_solve6(double* aug, int nc, double* sol, int* piv, int* used, int tid) {
  if (tid < 6) used[tid] = 0;
  __syncthreads();
  for (int k = 0; k < 6; ++k) {
    if (tid == 0) {
      int p = -1;
      double best = -1.0;
      for (int i = 0; i < 6; ++i)
        if (!used[i]) {
          double v = fabs(aug[i * nc + k]);
          if (p < 0 || v > best) best = v, p = i;
        }
      piv[k] = p;
      used[p] = 1;
    }
    __syncthreads();
    const int p = piv[k];
    const int i = tid / nc, j = tid - i * nc;
    if (i < 6 && !used[i] && j > k) {
      double f = aug[i * nc + k] / aug[p * nc + k];
      aug[i * nc + j] -= f * aug[p * nc + j];
    }
    __syncthreads();
  }
  const int nrhs = nc - 6;
  if (tid < nrhs) {
    const int col = 6 + tid;
    double x[6];
#pragma unroll
    for (int k = 5; k >= 0; --k) {
      const int p = piv[k];
      double s = aug[p * nc + col];
#pragma unroll
      for (int j = k + 1; j < 6; ++j) s -= aug[p * nc + j] * x[j];
      x[k] = s / aug[p * nc + k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) sol[k * nrhs + tid] = x[k];
  }
  __syncthreads();
}

struct JosephScratch {
  double IKH[324], T[324], O[324];
  double aug[72], Y[36], Zt[36], PSZ[108];
  int piv[6], used[6];
};

// P: the prior covariance (LDS), A: the 21 sums of the last iteration's H^T H (LDS), out: this scan's 324 doubles in
// global memory.  Every thread of the block calls (barriers inside); P and A must be visible to all on entry.
template <int BLOCK>
__device__ __forceinline__ void joseph_update(double r2, bool diverged, const double* P, const double* A, JosephScratch& s,
                                              double* __restrict__ out, int tid) {
  static_assert(BLOCK >= 72, "block_solve6 spreads a 6 x 12 system over the block");
  if (diverged) {  // (block-uniform)
    for (int k = tid; k < 324; k += BLOCK) out[k] = P[k];
    return;
  }
  if (tid < 36) {
    int i = tid / 6, j = tid - i * 6;
    double t = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) t += sym6(A, i, k) * P[sidx(k) * 18 + sidx(j)];
    s.aug[i * 12 + j] = t + (i == j ? r2 : 0.0);
    s.aug[i * 12 + 6 + j] = sym6(A, i, j);
  }
  __syncthreads();
  block_solve6(s.aug, 12, s.Y, s.piv, s.used, tid);  // Y = N^-1 A
  if (tid < 36) {
    int i = tid / 6, j = tid - i * 6;
    double t = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) t += sym6(A, i, k) * P[sidx(k) * 18 + sidx(j)];
    s.aug[i * 12 + j] = t + (i == j ? r2 : 0.0);
    s.aug[i * 12 + 6 + j] = s.Y[j * 6 + i];  // Y^T
  }
  __syncthreads();
  block_solve6(s.aug, 12, s.Zt, s.piv, s.used, tid);  // Zt = N^-1 Y^T  => Z = Y N^-T
  for (int e = tid; e < 324; e += BLOCK) {
    int i = e / 18, j = e - i * 18;
    double v = (i == j) ? 1.0 : 0.0;
    int kj = (j < 3) ? j : ((j >= 6 && j < 9) ? j - 3 : -1);  // KH has non-zero columns only in S
    if (kj >= 0) {
      double acc = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc += P[i * 18 + sidx(k)] * s.Y[k * 6 + kj];
      v -= acc;
    }
    s.IKH[e] = v;
  }
  for (int e = tid; e < 108; e += BLOCK) {
    int i = e / 6, j = e - i * 6;
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += P[i * 18 + sidx(k)] * s.Zt[j * 6 + k];  // Z[k][j] = Zt[j][k]
    s.PSZ[e] = acc;
  }
  __syncthreads();
  for (int e = tid; e < 324; e += BLOCK) {
    int i = e / 18, j = e - i * 18;
    double acc = 0;
    for (int k = 0; k < 18; ++k) acc += s.IKH[i * 18 + k] * P[k * 18 + j];
    s.T[e] = acc;
  }
  __syncthreads();
  for (int e = tid; e < 324; e += BLOCK) {
    int i = e / 18, j = e - i * 18;
    double acc = 0;
    for (int k = 0; k < 18; ++k) acc += s.T[i * 18 + k] * s.IKH[j * 18 + k];
    double kk = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) kk += s.PSZ[i * 6 + k] * P[j * 18 + sidx(k)];
    s.O[e] = acc + r2 * kk;
  }
  __syncthreads();
  for (int e = tid; e < 324; e += BLOCK) {
    int i = e / 18, j = e - i * 18;
    out[e] = 0.5 * (s.O[i * 18 + j] + s.O[j * 18 + i]);
  }
}

}  // namespace lins
