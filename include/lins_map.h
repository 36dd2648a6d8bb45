/* lins_map.h — C ABI of the scan-to-map correspondence / optimisation row.
 *
 * Replaces, in the reference's mapping node (src/lidar_mapping_node.cpp, "LM"):
 *   cornerOptimization  LM:1351-1453   5-NN in the local corner map, 3x3 covariance eigen-fit,
 *                                      point-to-line coefficients
 *   surfOptimization    LM:1455-1521   5-NN in the local surf map, 5-point plane fit, point-to-plane
 *   LMOptimization      LM:1523-1633   6-DoF Gauss-Newton step on (rx, ry, rz, tx, ty, tz) with the
 *                                      degeneracy projection of iteration 0
 *   scan2MapOptimization LM:1635-1652  up to 10 rounds of the three
 * with pointAssociateToMap (LM:579-607).  All arithmetic is f32 as in the reference; its third-party
 * pieces (FLANN 5-NN, cv::eigen, cv::solve(DECOMP_QR), cv::Mat::inv) are restated with fixed
 * operation sequences — exact 5-NN ordered by (distance, index), cyclic-Jacobi eigen-decomposition,
 * Householder QR — see DESIGN.md; parity is against oracle/map_oracle.cpp (unpinned like the rest:
 * the reference cannot be built here).
 */
#ifndef LINS_MAP_H_
#define LINS_MAP_H_

#include "lins_ieskf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lins_map_problem {
  const lins_point* map_corner; /* laserCloudCornerFromMapDS */
  const lins_point* map_surf;   /* laserCloudSurfFromMapDS   */
  const lins_point* scan_corner; /* laserCloudCornerLastDS    */
  const lins_point* scan_surf;   /* laserCloudSurfTotalLastDS */
  int32_t n_map_corner, n_map_surf, n_scan_corner, n_scan_surf;
  float transform[6]; /* transformTobeMapped: rx, ry, rz, tx, ty, tz */
  int32_t reserved[2]; /* [0]: flags (LINS_MAP_REUSE), [1]: 0 */
} lins_map_problem;

/* reserved[0] flag: the two map clouds of this problem are the ones of the previous call at the same batch index
 * (the mapping node's local map only changes with its key frames): when EVERY problem of a batch says so and the
 * sizes match, the maps already resident on the device — uploaded and bucketed into 1 m cells by the last call —
 * are used as they are; map_corner / map_surf are not read.                                                     */
#define LINS_MAP_REUSE 1

/* one query of cornerOptimization / surfOptimization */
typedef struct lins_map_corr {
  int32_t ind[5];   /* the 5 nearest map points, ascending (distance, index); -1 when fewer than 5 lie within 1 m */
  int32_t accepted; /* row pushed to laserCloudOri / coeffSel (LM:1446-1449, 1514-1517) */
  float coeff[4];   /* (s la, s lb, s lc, s ld2) / (s pa, s pb, s pc, s pd2) */
  float sel[3];     /* pointSel = pointAssociateToMap(pointOri) */
  float sq5;        /* pointSearchSqDis[4] (inf when fewer than 5 within 1 m) */
} lins_map_corr;

typedef struct lins_map_result {
  float transform[6];
  int32_t iters;      /* rounds run (<= 10) */
  int32_t converged;  /* LMOptimization returned true (deltaR < 0.05 deg && deltaT < 0.05 cm) */
  int32_t degenerate; /* isDegenerate after round 0 */
  int32_t n_sel;      /* rows selected in the last round */
} lins_map_result;

/* one correspondence pass at in->transform: n_scan_corner + n_scan_surf records */
int lins_map_correspondences(lins_ctx* ctx, const lins_map_problem* in, lins_map_corr* corner, lins_map_corr* surf);
/* scan2MapOptimization for n independent problems, entirely on the device: the maps are bucketed into 1 m cells by
 * a counting-sort kernel (or reused, LINS_MAP_REUSE), then the up to 10 rounds of {correspondence + row + reduction
 * kernel, 6x6 Gauss-Newton step kernel with the degeneracy projection} run back to back; the precondition of LM:1636
 * (> 10 corner and > 100 surf map points) not met => transform returned unchanged with iters = 0 */
int lins_scan2map_batch(lins_ctx* ctx, int n, const lins_map_problem* in, lins_map_result* out);
/* HIP-event time (ms) of the device sequence of the last call (the rounds' kernels; lins_map_correspondences: its
 * one pass) and the number of query evaluations it did */
int lins_last_map_stats(lins_ctx* ctx, float* kernel_ms, uint64_t* queries);

#ifdef __cplusplus
}
#endif
#endif /* LINS_MAP_H_ */
