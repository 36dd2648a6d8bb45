/*
 * lins_host.h — host-side (CPU, C++ behind a C ABI) pieces that sit either
 * side of the IESKF hot path.  None of this touches the GPU except
 * lins_host_perform_ieskf(), which drives liblins_ieskf.so.
 *
 *   lins_filter_*            mirrors filter::StatePredictor
 *                            (lins/include/KalmanFilter.hpp:118-380): IMU
 *                            propagation + reset(1); produces the prior (x,P)
 *                            performIESKF() starts from.
 *   lins_frontend_*          mirrors image_projection_node's projection /
 *                            ground / segmentation (lins/src/image_projection_node.cpp:191-415)
 *                            and StateEstimator's feature front-end
 *                            (StateEstimator.hpp:619-827); produces the four
 *                            feature clouds performIESKF() reads.
 *   lins_transform_to_end    StateEstimator::transformToEnd (SE:1083-1101), the
 *                            re-projection updatePointCloud() applies to make
 *                            the next scan's target clouds (SE:1116-1139).
 *   lins_synth_*             seeded synthetic VLP-16 scan pairs.
 *   lins_host_perform_ieskf  StateEstimator::performIESKF() as the node sees it
 *                            (SE:465-600): GPU IESKF loop, and on divergence the
 *                            ICP fallback estimateTransform (SE:585-592,
 *                            1163-1320) with GPU correspondences + host 6x6 GN.
 */
#ifndef LINS_HOST_H_
#define LINS_HOST_H_

#include "lins_ieskf.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LINS_LINE_NUM 16   /* LINE_NUM  (exp_port.yaml:9)  */
#define LINS_SCAN_NUM 1800 /* SCAN_NUM  (exp_port.yaml:10) */
#define LINS_CLOUD_MAX (LINS_LINE_NUM * LINS_SCAN_NUM)

/* ---- StatePredictor ------------------------------------------------------ */
typedef struct lins_filter_params {
  double acc_n, gyr_n, acc_w, gyr_w;                   /* yaml:29-32           */
  double init_pos_std[3], init_vel_std[3], init_att_std[3]; /* yaml:34-50      */
  double init_acc_std[3], init_gyr_std[3];             /* yaml:52-62           */
} lins_filter_params;

typedef struct lins_filter {
  double state[LINS_STATE_DIM];
  double cov[LINS_ERR_DIM * LINS_ERR_DIM];
  double noise[12 * 12];
  double acc_last[3], gyr_last[3];
  double time;
  int32_t has_imu;
  int32_t pad;
  lins_filter_params prm;
} lins_filter;

void lins_filter_default_params(lins_filter_params* p);           /* exp_port.yaml */
/* StatePredictor::initialization-like start: identity state with given v, ba,
 * bw, gravity (0,0,-9.81), covariance = initializeCovariance(0) (KF:247-311).  */
void lins_filter_init(lins_filter* f, const lins_filter_params* p, const double* vn,
                      const double* ba, const double* bw);
void lins_filter_predict(lins_filter* f, double dt, const double* acc,
                         const double* gyr);                      /* KF:125-186  */
void lins_filter_reset1(lins_filter* f);                          /* KF:320-352  */

/* ---- front-end ------------------------------------------------------------ */
typedef struct lins_features {
  lins_point* corner_sharp;      int32_t n_corner_sharp;      /* cap 192  */
  lins_point* corner_less_sharp; int32_t n_corner_less_sharp; /* cap 1920 */
  lins_point* surf_flat;         int32_t n_surf_flat;         /* cap 1024 */
  lins_point* surf_less_flat;    int32_t n_surf_less_flat;    /* cap LINS_CLOUD_MAX */
  int32_t n_segmented;           /* size of the segmented cloud              */
  int32_t n_outlier;
} lins_features;

/* raw: unorganised cloud in firing order (what /velodyne_points carries).
 * Caller allocates the four output arrays with the capacities noted above.   */
int lins_frontend_extract(const lins_point* raw, int n_raw, double scan_period,
                          lins_features* out);

/* ---- front-end, in the reference's two stages ------------------------------ */
/* What image_projection_node publishes and StateEstimator::processPCL consumes: the segmented
 * cloud (sensor_msgs/PointCloud2 of PointXYZI, ring-major, ascending column, intensity =
 * row + col / 10000, IP:234) + cloud_msgs/cloud_info (cloud_info.msg:1-12).                  */
typedef struct lins_segmented_scan {
  const lins_point* cloud;
  const float* range;       /* segmentedCloudRange      */
  const uint32_t* col;      /* segmentedCloudColInd     */
  const uint8_t* ground;    /* segmentedCloudGroundFlag */
  int32_t n;
  int32_t start_ring[LINS_LINE_NUM], end_ring[LINS_LINE_NUM];  /* startRingIndex / endRingIndex */
  float start_ori, end_ori, ori_diff;                          /* startOrientation, ...         */
  int32_t n_outlier;
} lins_segmented_scan;

/* image_projection_node (IP:191-415) on the host: caller-allocated arrays of LINS_CLOUD_MAX
 * entries, `out` is pointed at them.                                                          */
int lins_frontend_segment(const lins_point* raw, int n_raw, lins_point* cloud, float* range, uint32_t* col,
                          uint8_t* ground, lins_segmented_scan* out);
/* The same stage on the device (csrc/segment_kernels.hip): n raw clouds in firing order; out[k]'s four array
 * pointers must point at caller-allocated arrays of LINS_CLOUD_MAX entries (they are written).  Identical
 * to lins_frontend_segment() — the BFS labelling is restated as an order-free min-label propagation.    */
int lins_segment_batch(lins_ctx* ctx, int n, const lins_point* const* raw, const int32_t* n_raw,
                       lins_segmented_scan* out);
int lins_last_segment_ms(lins_ctx* ctx, float* kernel_ms);
/* StateEstimator's feature stage (undistortPcl .. extractFeatures, SE:619-827) on the host — the
 * CPU restatement the device version is checked against.                                      */
int lins_frontend_extract_segmented(const lins_segmented_scan* in, double scan_period, lins_features* out);
/* The same stage on the device (csrc/frontend_kernels.hip): n scans, one
 * workgroup each; out[k]'s four arrays are caller-allocated with the capacities of lins_features.
 * Identical picks and clouds (the relative-time tag within 1 ulp of f32 where atan2f differs).   */
int lins_extract_features_batch(lins_ctx* ctx, int n, const lins_segmented_scan* in, double scan_period,
                                lins_features* out);
/* HIP-event time (ms) of the front-end kernel of the last call and its algorithmic bytes
 * (25 B read per segmented point + 16 B per emitted feature point).                            */
int lins_last_frontend_stats(lins_ctx* ctx, float* kernel_ms, uint64_t* bytes);

/* ---- device-resident streams: front-end -> IESKF update -> re-projection, the clouds never leave HBM ----
 * n independent streams (sensors / robots / replayed logs) advance one scan per call:
 *   1. the feature stage (SE:619-827) of every stream's new segmented scan, into device slots;
 *   2. performIESKF (SE:465-600) of the new sharp / flat clouds against the stream's RESIDENT less-sharp /
 *      less-flat clouds of the previous scan, from the prior (state, covariance) the caller's StatePredictor
 *      supplies; diverged filters take the device ICP fallback (SE:585-592);
 *   3. updatePointCloud (SE:1116-1139): the new less-sharp / less-flat clouds re-projected to the scan end
 *      with the final pose, in place — the next call's targets.
 * A stream's first scan has nothing to match against: no update is run (out[k].iters = 0, state / covariance
 * returned as given) and its clouds are re-projected with the pose in prior_state[k] (the caller's bootstrap
 * guess, SE:331-425 uses the IMU-integrated one).  prior_state: n x 19, prior_cov: n x 324, feature_counts
 * (optional): n x 4 = sharp, less sharp, flat, less flat.
 * A step completes for EVERY stream or not at all: a diverged stream whose clouds cannot take the device ICP
 * fallback (ICP_FREQ != 1, or clouds beyond the grid kernels' limits) keeps its un-updated filter — out[k].diverged
 * set and out[k].reserved[0] = LINS_E_UNSUPPORTED — while the other streams advance normally; a rejected input
 * (LINS_E_INPUT / _ARG / _CAPACITY) advances nothing; a HIP error (LINS_E_HIP) leaves the resident clouds in an
 * unknown state, every later step then returns LINS_E_STATE until lins_streams_init is called again.       */
int lins_streams_init(lins_ctx* ctx, int n_streams);   /* n_streams <= the context's max_batch */
int lins_streams_step(lins_ctx* ctx, const lins_segmented_scan* scans, const double* prior_state,
                      const double* prior_cov, double scan_period, lins_result* out, int32_t* feature_counts);
/* the same step from RAW clouds (firing order): the image_projection stage (IP:191-415) runs on the device
 * too (lins_last_segment_ms reports its kernel time) and hands the segmented scan to the front-end in HBM */
int lins_streams_step_raw(lins_ctx* ctx, const lins_point* const* raw, const int32_t* n_raw, const double* prior_state,
                          const double* prior_cov, double scan_period, lins_result* out, int32_t* feature_counts);
/* HIP-event times (ms) of the three stages of the last step */
int lins_streams_stats(lins_ctx* ctx, float* frontend_ms, float* update_ms, float* reproject_ms);
/* test aid: a resident cloud of the last scan back to the host (which: 0 less sharp, 1 less flat);
 * returns the point count                                                                            */
int lins_streams_peek(lins_ctx* ctx, int stream, int which, lins_point* out, int cap);

/* the front-end's atan2 (csrc/lins_math.h: a fixed f32 operation sequence shared bit for bit by the host
 * restatement and the device kernels; within 2 ulp(pi/4) of the true value).
 * KNOWN DEVIATION from the reference, which calls libm's atan2 where points are BINNED by angle (image row / column
 * IP:217-225, ground angle IP:262, segmentation angle IP:381, relative time SE:630): a point whose angle lies within
 * the last bit of a bin edge may fall into the neighbouring bin.  Measured against an independent glibc-based checker
 * (oracle/frontend_oracle.cpp, tools/frontend_vs_libm.py, profiles/r02_frontend_vs_libm.txt): on clouds with generic
 * azimuths 0 of 18.4 M cells, 0 picks and 0 coordinates differ, 0.8 % of the relative-time tags differ by at most two
 * f32 roundings; on clouds whose every firing sits exactly ON a column edge (the stock synthetic sensor), 22 % of the
 * cells differ — between any two atan2f implementations.                                               */
float lins_host_atan2f(float y, float x);

/* transformToEnd for every point, with the scan's final relative pose
 * (t = linState_.rn_, q = linState_.qbn_ as w,x,y,z). In-place allowed.       */
void lins_transform_to_end(const double* t, const double* q_wxyz, double scan_period,
                           const lins_point* in, int n, lins_point* out);

/* ---- synthetic scan pairs -------------------------------------------------- */
typedef struct lins_synth_pair {
  /* caller-allocated, capacities as in lins_features */
  lins_point* surf_flat;      int32_t n_surf_flat;
  lins_point* corner_sharp;   int32_t n_corner_sharp;
  lins_point* surf_last;      int32_t n_surf_last;
  lins_point* corner_last;    int32_t n_corner_last;
  double state[LINS_STATE_DIM];            /* prior x for performIESKF          */
  double cov[LINS_ERR_DIM * LINS_ERR_DIM]; /* prior P                            */
  double true_t[3], true_q[4];             /* ground-truth relative pose (w,x,y,z) */
  double speed, yaw_rate;
  int32_t n_raw_last, n_raw_new;
} lins_synth_pair;

#define LINS_SYNTH_SEED 0x4C494E53u /* "LINS" */
int lins_synth_generate(uint32_t seed, uint32_t scan_index, lins_synth_pair* out);
/* raw distorted cloud of one synthetic scan (k = 0 or 1), firing order        */
int lins_synth_raw_scan(uint32_t seed, uint32_t scan_index, int k, lins_point* out,
                        int cap);
/* The same for a scene FAMILY: 0 = the room of SURVEY.md section 8d (what the two calls above generate); 1 = "open":
 * open ground to the range limit, ~60 trunks, six far wall segments, 30 % of the returns lost, one box that moves at
 * up to 5 m/s (csrc/host/synth.cpp) — the second workload the parity sweeps and the bench line's scene_b block run.   */
int lins_synth_generate_scene(int scene, uint32_t seed, uint32_t scan_index, lins_synth_pair* out);
int lins_synth_raw_scan_scene(int scene, uint32_t seed, uint32_t scan_index, int k, lins_point* out, int cap);

/* A SEQUENCE of scans along one seeded trajectory (a circle inside the synthetic room) with its IMU: what a
 * lins_fusion_node would receive over any number of consecutive sweeps — the input of the in-situ test of the drop-in
 * boundary (tests/test_gpu_sequence.py).  Sweep k covers [0.1 k, 0.1 (k + 1)) s; 40 IMU samples per sweep.       */
int lins_synth_seq_raw_scan(uint32_t seed, int k, lins_point* out, int cap);
int lins_synth_seq_imu(uint32_t seed, int k, double* acc /* 40 x 3 */, double* gyr /* 40 x 3 */);
int lins_synth_seq_truth(uint32_t seed, double tau, double* xyyaw, double* speed, double* yaw_rate);

/* ---- performIESKF as the node sees it -------------------------------------- */
int lins_host_perform_ieskf(lins_ctx* ctx, const lins_params* prm,
                            const lins_scan_pair* in, lins_result* out,
                            int32_t* used_icp_fallback);

#ifdef __cplusplus
}
#endif
#endif
