/*
 * lins_oracle.h — C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a dependency-free CPU restatement
 * of the reference's IESKF update path (see lins_oracle.cpp for the file:line
 * map).  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg may load it; the product (liblins_ieskf.so) never links or calls it.
 *
 * PINNED since round 3: the reference ships no tests, fixtures or golden
 * vectors for this path and its build needs ROS/PCL/Eigen/OpenCV (none on
 * disk) — but the path is header-only, and oracle/_ref/liblins_ref.so is the
 * reference's own StateEstimator.hpp compiled verbatim against stand-in
 * headers (oracle/ref_shim, oracle/ref_driver.cpp, `make -C oracle _ref`).
 * tests/test_ref.py holds this oracle against it: correspondence indices,
 * accepted sets and f32 rows bit for bit, states and covariances to 1e-12 on
 * 1536 seeded pairs, the ICP fallback, the divergence and NaN branches.  What
 * the stand-ins can only restate (Eigen's summation order, FLANN's and
 * std::sort's tie order) is listed in DESIGN.md §3.  Before round 3 the oracle
 * was pinned only by (a) hand-computed known-answer geometry, (b) algebraic
 * identities, (c) scipy.cKDTree for the 1-NN indices, (d) its own
 * dense-vs-reduced cross-check — those tests still run.
 */
#ifndef LINS_ORACLE_H_
#define LINS_ORACLE_H_

#include "../include/lins_ieskf.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_FORM_DENSE = 0, ORACLE_FORM_REDUCED = 1 };
enum { ORACLE_NN_KDTREE = 0, ORACLE_NN_BRUTE = 1 };

/* Optional per-iteration trace. All pointers may be NULL individually. */
typedef struct oracle_trace {
  int32_t max_iters;   /* capacity of every array below, in iterations          */
  lins_corr* surf;     /* [max_iters][n_surf_flat]                              */
  lins_corr* corner;   /* [max_iters][n_corner_sharp]                           */
  double* lin_state;   /* [max_iters][19] linearisation state entering iter k   */
  double* dx;          /* [max_iters][18] updateVec_ of iter k                  */
  double* sums28;      /* [max_iters][28] H^T H upper triangle (21), H^T r (6), r^T r */
} oracle_trace;

/* findCorrespondingSurfFeatures / findCorrespondingCornerFeatures for one
 * linearisation state (StateEstimator.hpp:829-953, 955-1063).                  */
int oracle_correspondences(const lins_params* prm, const lins_scan_pair* in,
                           const double* lin_state, int iter, int nn_mode,
                           lins_corr* surf, lins_corr* corner);

/* performIESKF without the ICP fallback (StateEstimator.hpp:465-583, 594-598):
 * on divergence the result carries the un-updated filter state/covariance.     */
int oracle_ieskf(const lins_params* prm, const lins_scan_pair* in, int form,
                 int nn_mode, lins_result* out, oracle_trace* trace);

/* estimateTransform (StateEstimator.hpp:1163-1196, 1198-1320): 6-DoF
 * Gauss-Newton on the same correspondences.  t[3], q[4]=(w,x,y,z) in/out.      */
int oracle_icp(const lins_params* prm, const lins_scan_pair* in, double* t,
               double* q, int nn_mode, int32_t* iters_run);

/* performIESKF including the divergence branch (StateEstimator.hpp:585-592).   */
int oracle_perform_ieskf(const lins_params* prm, const lins_scan_pair* in,
                         int form, int nn_mode, lins_result* out);

/* Exact 1-NN (lowest index wins ties), f32 L2_Simple order ((dx²+dy²)+dz²).    */
int oracle_perform_ieskf_hook(void* user, const lins_params* prm, const lins_scan_pair* in, lins_result* out,
                              int32_t* used_icp);
int oracle_nn(const lins_point* targets, int n_targets, const lins_point* queries,
              int n_queries, int nn_mode, int32_t* idx, float* sqdist);

/* helpers exposed for identity tests (math_utils.h / KalmanFilter.hpp)         */
void oracle_quat2axis(const double* q_wxyz, double* axis3);
void oracle_axis2quat(const double* axis3, double* q_wxyz);
void oracle_rinvleft(const double* axis3, double* m9);
void oracle_box_plus(const double* state19, const double* dx18, double* out19);
void oracle_box_minus(const double* a19, const double* b19, double* out18);
void oracle_transform_to_start(const lins_params* prm, const double* lin_state,
                               const lins_point* in, lins_point* out);

/* Timed throughput loop for bench.py's cpu_baseline: runs oracle_ieskf over
 * `n` pairs with `threads` std::threads, returns wall seconds and total
 * iterations executed.                                                         */
int oracle_bench(const lins_params* prm, int n, const lins_scan_pair* in, int form,
                 int nn_mode, int threads, double* seconds, uint64_t* iters);

#ifdef __cplusplus
}
#endif
#endif
