// ieskf_lds_tail.h — the serial tail of one iteration of the LDS IESKF kernels and the covariance epilogue: the 6 x 6 solve +
// boxPlus on one wave (solve_wave0), the next iteration's constants (next_iter_consts), the Gauss-Newton row and step of
// the ICP fallback (icp_row_dev, icp_solve_and_update), the Joseph update (joseph_epilogue).  Included by ieskf_lds_impl.h
// INSIDE its instantiation namespace, behind the definition of g_lds / LdsStore (code motion of round 6: the hot header
// had grown past 2 200 lines; nothing here changed).  Reference: SE:542-580 (gain, increment, stop rules), SE:594-598
// (Joseph update), SE:1163-1320 (estimateTransform).
#pragma once

// ---------------------------------------------------------------------------
// The serial tail of one iteration, kept out of line: its register needs (a 6x7 system in
// registers, the 19-state) are allocated on their own instead of inflating — and spilling —
// the search loop it would otherwise be fused with.  Called by every thread (barriers inside).
// ---------------------------------------------------------------------------
// Round 4: the out-of-line bodies are entered by the waves that work in them only — solve_wave0 by wave 0,
// next_iter_consts by waves 0-2 — and the barriers between them are the caller's.  As one function called by all eight
// waves (rounds 2-3) every wave ran its prologue and epilogue, nine callee-saved registers to scratch and back per wave
// and iteration: ~190 MB of scratch stores per launch of 1024 scans, most of the 208 MB WRITE_SIZE counted (the
// hand-over of the several-part updates is 35 MB of it).
__device__ __noinline__ long long solve_wave0(double prm_r2, int prm_fixed_iters, int lane, bool prof) {
  long long t3 = 0;
  LdsStore& L = g_lds;
  // ---- wave 0: (sigma^2 I + A P_SS) w = g + A d_S  (push-through form of SE:542-549) solved across the wave
  // (wave_gj_solve6: Gauss-Jordan, one element per lane, no back-substitution), dx = d - P[:,S] w, NaN / divergence /
  // convergence tests and boxPlus (SE:552-580).  This wave walks a chain of dependent f64 operations while the other
  // seven wait at the barrier, so the chain is kept short: the rotation maps take their short-series forms
  // (lins_math.h axis2quat_fast, quat2axis_fast, phi_and_gt_small — no libm call, no square root, one division for
  // the rotations an update sees) and fall back to libm outside their range.  Round 2, measured in isolation
  // (tools/tail_cycles.py): solve 3418 -> 2487 cycles, boxPlus' axis2quat 1487 -> 582, phi + Rinvleft 2087 -> 863;
  // in the kernel the tail's share of a late iteration fell from 12.8 to 5.4 us per launch.  The new linearisation state is STAGED in LDS — the old one may
  // still be read by the other waves until the barrier — and after it three waves split the constants of the next
  // iteration: wave 0 -> linState_, R^T;  wave 1 -> phi, Rinvleft(-phi)^T;  wave 2 -> x_filter (-) x_lin.
  // (Until round 2 the first three waves each ran the whole solve redundantly to save the staging: 2 x ~2.5 k
  // issued instructions per iteration for nothing.)
  double* const stage = &L.aug[0][0];  // 22 doubles: linState_ (19), |r|, |r| kept, |dx|
  int* const stage_flags = reinterpret_cast<int*>(&L.aug[1][0]);  // diverged, converged
  {
    double v = 0.0;
    if (lane < 42) {
      const int i = lane / 7, j = lane % 7;
      if (j < 6) {
        v = (i == j ? prm_r2 : 0.0);
#pragma unroll
        for (int k = 0; k < 6; ++k) v += sym6(L.sums, i, k) * L.P[sidx(k) * 18 + sidx(j)];
      } else {
        v = L.sums[21 + i];
#pragma unroll
        for (int k = 0; k < 6; ++k) v += sym6(L.sums, i, k) * L.ic.d[sidx(k)];
      }
    }
    // (what dx needs from LDS besides the solution is read BEFORE the solve: the reads then wait behind nothing)
    double pls[6] = {0, 0, 0, 0, 0, 0}, dl = 0;
    if (lane < 18) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pls[k] = L.P[lane * 18 + sidx(k)];
      dl = L.ic.d[lane];
    }
    double wsol[6];
#ifdef LINS_PROF_TAIL
    if (prof && lane == 0) L.prof_tail[0] = clock64();
#endif
    wave_gj_solve6(v, lane, wsol);
#ifdef LINS_PROF_TAIL
    asm volatile("" ::"v"(wsol[0]) : "memory");
    if (prof && lane == 0) L.prof_tail[1] = clock64();
#endif
    double dxi = 0;
    if (lane < 18) {
      double sacc = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) sacc += pls[k] * wsol[k];
      dxi = dl - sacc;
    }
    if (prof) t3 = clock64();
    double lin[19];
#pragma unroll
    for (int k = 0; k < 19; ++k) lin[k] = L.ic.lin[k];
    double dth[3] = {0, 0, 0};
    bool has_nan = false;
    double un = 0;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const double vk = readlane_f64(dxi, k);
      has_nan = has_nan || isnan(vk);
      un += vk * vk;
      if (k >= 6 && k < 9)
        dth[k - 6] = vk;
      else
        lin[k < 6 ? k : k + 1] += vk;  // p,v at 0..5; ba,bw,g at 10..18 (q occupies 6..9)
    }
    un = sqrt(un);
    const double rn = sqrt(L.sums[27]);
    double res_prev = L.res_prev;
    int div = 0, conv = 0;
    if (has_nan) {
      div = 2, un = L.upd_norm;
    } else if (rn > res_prev * 10) {
      div = 1, un = L.upd_norm;
    } else {
      const Q4 qn = qnormalized(qmul(Q4{lin[6], lin[7], lin[8], lin[9]}, axis2quat_fast(V3{dth[0], dth[1], dth[2]})));
      lin[6] = qn.w, lin[7] = qn.x, lin[8] = qn.y, lin[9] = qn.z;
      if (un <= 1e-2 && !prm_fixed_iters) conv = 1;
      res_prev = rn;
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 19; ++k) stage[k] = lin[k];
      stage[19] = rn, stage[20] = res_prev, stage[21] = un;
      stage_flags[0] = div, stage_flags[1] = conv;
    }
  }
  return t3;
}
// the constants of the next iteration from the staged linearisation state: wave 0 -> linState_, R^T; wave 1 -> phi,
// Rinvleft(-phi)^T; wave 2 -> x_filter (-) x_lin
__device__ __noinline__ void next_iter_consts(int wave, int lane) {
  LdsStore& L = g_lds;
  const double* const stage = &L.aug[0][0];
  {
    const Q4 q{stage[6], stage[7], stage[8], stage[9]};
    // (static indices only: a lane-indexed register array would be spilled to scratch)
    if (wave == 0) {
      const M3 Rt = mtrans(qmat(q));
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 19; ++k) L.ic.lin[k] = stage[k];
        L.ic.Rt = Rt;
      }
    } else if (wave == 1) {
      V3 phi;
      M3 Gt;
      phi_and_Gt(q, phi, Gt);
      if (lane == 0) L.ic.phi = phi, L.ic.Gt = Gt;
    } else {
      // boxMinus(filter, lin), KF:84-94
      const Q4 qf{L.filt[6], L.filt[7], L.filt[8], L.filt[9]};
      const V3 da = quat2axis_fast(qmul(qinverse(q), qf));
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          L.ic.d[0 + k] = L.filt[0 + k] - stage[0 + k];
          L.ic.d[3 + k] = L.filt[3 + k] - stage[3 + k];
          L.ic.d[9 + k] = L.filt[10 + k] - stage[10 + k];
          L.ic.d[12 + k] = L.filt[13 + k] - stage[13 + k];
          L.ic.d[15 + k] = L.filt[16 + k] - stage[16 + k];
        }
        L.ic.d[6] = da.x, L.ic.d[7] = da.y, L.ic.d[8] = da.z;
      }
    }
  }
}
// (Scalars by value, the profile stamp returned: a reference to the kernel's parameter struct or to a local would
// force them into scratch for the whole kernel — every later read of a parameter a scratch load.)
__device__ __forceinline__ long long solve_and_update(double prm_r2, int prm_fixed_iters, int prm_pad, int tid, int iter, bool prof) {
  long long t3 = 0;
  LdsStore& L = g_lds;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (prm_pad & 0x10000) {  // counting aid (LINS_DEBUG_SKIP bit 0x10000): no solve, no update — the state stands still
    if (tid == 0) L.iter = iter + 1;
    __syncthreads();
    return t3;
  }
  const double* const stage = &L.aug[0][0];  // 22 doubles: linState_ (19), |r|, |r| kept, |dx|
  const int* const stage_flags = reinterpret_cast<const int*>(&L.aug[1][0]);  // diverged, converged
#ifdef LINS_PROF_TAIL  // (build with -DLINS_PROF_WAVES=99 -DLINS_PROF_TAIL=1: slots 6..11 of the phase profile = the tail's sub-phases on thread 0, tools/tail_phases.py)
  const long long a0 = prof ? clock64() : 0;
#endif
  if (wave == 0) t3 = solve_wave0(prm_r2, prm_fixed_iters, lane, prof);
#ifdef LINS_PROF_TAIL
  const long long a1 = prof ? clock64() : 0;
#endif
  __syncthreads();  // every reader of the old linearisation state is done; the staged one is visible
#ifdef LINS_PROF_TAIL
  const long long a2 = prof ? clock64() : 0;
#endif
  const int div = stage_flags[0];
  if (wave < 3 && !div) next_iter_consts(wave, lane);
#ifdef LINS_PROF_TAIL
  const long long a3 = prof ? clock64() : 0;
#endif
  if (tid == 0) {
    L.res_last = stage[19], L.res_prev = stage[20], L.upd_norm = stage[21];
    L.conv = stage_flags[1], L.div = div;
    L.iter = iter + 1;
  }
  __syncthreads();
#ifdef LINS_PROF_TAIL
  if (prof && tid == 0) {
    const long long a4 = clock64();
    L.prof_acc[6] += L.prof_tail[0] - a0, L.prof_acc[7] += L.prof_tail[1] - L.prof_tail[0], L.prof_acc[8] += a1 - L.prof_tail[1];
    L.prof_acc[9] += a2 - a1, L.prof_acc[10] += a3 - a2, L.prof_acc[11] += a4 - a3;
  }
#endif
  return t3;
}

// The Gauss-Newton row of the fallback, out of line: inlined into the search loop its rotation matrix and the 3 x 3
// product (R(s phi), -R [p]x) cost the loop 124 spilled registers (1.7 KB of scratch per lane, round 2) for a path that
// runs once per accepted row.  Scalars by value, the row comes back by value.
struct IcpRow {
  double v[7];
};
__device__ __noinline__ IcpRow icp_row_dev(double inv_period, double phx, double phy, double phz, float px, float py, float pz,
                                           float intensity, float c0, float c1, float c2, float c3) {
  IcpRow r;
  const float c[4] = {c0, c1, c2, c3};
  icp_row(inv_period, V3{phx, phy, phz}, px, py, pz, intensity, c, r.v, r.v[6]);
  return r;
}

// ---------------------------------------------------------------------------
// Serial tail of one ICP iteration (estimateTransform's loop body after the correspondences,
// SE:1170-1195): needs >= 10 plane and >= 5 line rows, else the iteration is spent without a step
// (SE:1175-1184); Gauss-Newton step + degeneracy projection + stop rule in icp_math.h.  A handful
// of 6x6 factorizations per divergence: one lane, its arrays in LDS.
// ---------------------------------------------------------------------------
// (called by wave 0 only — the out-of-line call's register saves then cost one wave, not eight; the caller's barrier
// publishes the new state)
__device__ __noinline__ void icp_solve_and_update(int lane, int iter) {
  LdsStore& L = g_lds;
  {  // the step over the wave (icp_wave.h: a matrix column per lane, the scalar routine's bits)
    int conv = 0;
    if (L.m_surf >= 10 && L.m_corner >= 5) {  // (uniform)
      static_assert(sizeof(L.partial) >= kIcpWorkspace * sizeof(double) && sizeof(L.aug) >= 48 * sizeof(double), "ICP workspace");
      double* const ws = L.partial;  // (the wave partials and the solve's staging area are idle here)
      double *const JTJ = &L.aug[0][0], *const JTb = JTJ + 36;
      if (lane < 36) {
        const int i = lane / 6, j = lane % 6;
        JTJ[lane] = L.sums[i <= j ? tri6(i, j) : tri6(j, i)];
      }
      if (lane < 6) JTb[lane] = L.sums[21 + lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double x[6];
      wave_icp_gn_solve(JTJ, JTb, iter, lane, x, ws);
      double t[3] = {L.ic.lin[0], L.ic.lin[1], L.ic.lin[2]};
      Q4 q{L.ic.lin[6], L.ic.lin[7], L.ic.lin[8], L.ic.lin[9]};
      conv = icp_apply(x, t, q) ? 1 : 0;
      const V3 phi = quat2axis(q);
      if (lane == 0) {
        L.ic.lin[0] = t[0], L.ic.lin[1] = t[1], L.ic.lin[2] = t[2];
        L.ic.lin[6] = q.w, L.ic.lin[7] = q.x, L.ic.lin[8] = q.y, L.ic.lin[9] = q.z;
        L.ic.phi = phi;
      }
    }
    if (lane == 0) {
      L.conv = conv;
      L.iter = iter + 1;
    }
  }
}

// ---------------------------------------------------------------------------
// Joseph covariance update (SE:594-598) as the workgroup's epilogue, in a form that is a rank-6 update of the prior:
// with S = {0,1,2,6,7,8}, C = P[:,S] (18 x 6), R = P[S,:], N = sigma^2 I + A P_SS, Y = N^-1 A, Z = Y N^-T (so that
// KH = C Y E_S^T and K R K^T = sigma^2 C Z C^T),
//   (I - KH) P (I - KH)^T + K R K^T  =  P  -  C Y R  -  C Y^T C^T  +  C (Y P_SS Y^T + sigma^2 Z) C^T
// — the four 18 x 18 x 18 products of the textbook form collapse to 18 x 6 x 6 and 18 x 18 x 6 ones, and the two
// 6 x 12 eliminations to ONE inverse: two waves run the 6 x 9 Gauss-Jordan of ieskf_rowsum.h side by side (left and
// right half of the identity), no block-wide elimination with its fourteen barriers.  Rounds 1-2 ran this update as
// a kernel of its own (ieskf_joseph_kernel, 128 threads per scan, ~20 us + a launch after every update kernel; a
// fused version of that block-wide algorithm cost as much as it saved); this epilogue costs a few microseconds of a
// workgroup that is about to exit.  The scratch is the grid's point storage, dead by now.  diverged: Pk_ is passed
// through un-updated (SE:592).  Called by every thread (barriers inside).
// ---------------------------------------------------------------------------
template <int BLOCK>
__device__ __noinline__ void joseph_epilogue(double r2, int diverged, double* __restrict__ out, int tid) {
  LdsStore& L = g_lds;
  const double* P = L.P;
  if (diverged) {  // (block-uniform)
    for (int k = tid; k < 324; k += BLOCK) out[k] = P[k];
    return;
  }
  double* const sc = reinterpret_cast<double*>(L.pt);  // >= 66 KB, no longer read
  double* const Ninv = sc;          // 36
  double* const Y = sc + 36;        // 36  Y = N^-1 A
  double* const T1 = sc + 72;       // 36  Y P_SS
  double* const M = sc + 108;       // 36  Y P_SS Y^T + sigma^2 Z - Y^T
  double* const D = sc + 144;       // 108 C M
  double* const E2 = sc + 252;      // 108 C Y
  double* const O = sc + 360;       // 324
  static_assert(sizeof(L.pt) >= (360 + 324) * sizeof(double), "scratch of the Joseph epilogue");
  __syncthreads();  // (every reader of the grid is done)
  const int lane = tid & 63, wave = tid >> 6;
  if (wave < 2) {  // N^-1, columns 3 wave .. 3 wave + 2
    double v = 0.0;
    if (lane < 54) {
      const int i = lane / 9, j = lane % 9;
      if (j < 6) {
        v = (i == j ? r2 : 0.0);
#pragma unroll
        for (int k = 0; k < 6; ++k) v += sym6(L.sums, i, k) * P[sidx(k) * 18 + sidx(j)];
      } else {
        v = (i == 3 * wave + (j - 6)) ? 1.0 : 0.0;
      }
    }
    v = wave_gj_solve6x3(v, lane);
    if (lane < 54 && lane % 9 >= 6) Ninv[(lane / 9) * 6 + 3 * wave + (lane % 9 - 6)] = v;
  }
  __syncthreads();
  if (tid < 36) {
    const int i = tid / 6, j = tid % 6;
    double y = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) y += Ninv[i * 6 + k] * sym6(L.sums, k, j);
    Y[tid] = y;
  }
  __syncthreads();
  if (tid < 36) {
    const int i = tid / 6, j = tid % 6;
    double t = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) t += Y[i * 6 + k] * P[sidx(k) * 18 + sidx(j)];
    T1[tid] = t;
  }
  __syncthreads();
  if (tid < 36) {
    const int i = tid / 6, j = tid % 6;
    double m = 0, z = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) m += T1[i * 6 + k] * Y[j * 6 + k], z += Y[i * 6 + k] * Ninv[j * 6 + k];
    M[tid] = (m + r2 * z) - Y[j * 6 + i];
  }
  __syncthreads();
  if (tid < 216) {
    const int e = tid < 108 ? tid : tid - 108, i = e / 6, b = e % 6;
    const double* W = tid < 108 ? M : Y;
    double acc = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) acc += P[i * 18 + sidx(a)] * W[a * 6 + b];
    (tid < 108 ? D : E2)[e] = acc;
  }
  __syncthreads();
  for (int e = tid; e < 324; e += BLOCK) {
    const int i = e / 18, j = e % 18;
    double acc = P[e];
#pragma unroll
    for (int b = 0; b < 6; ++b) acc += D[i * 6 + b] * P[j * 18 + sidx(b)] - E2[i * 6 + b] * P[sidx(b) * 18 + j];
    O[e] = acc;
  }
  __syncthreads();
  for (int e = tid; e < 324; e += BLOCK) {
    const int i = e / 18, j = e % 18;
    out[e] = 0.5 * (O[i * 18 + j] + O[j * 18 + i]);  // enforceSymmetry (MU:39-41)
  }
}

