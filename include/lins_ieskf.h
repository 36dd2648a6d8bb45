/*
 * lins_ieskf.h — C ABI of the MI355X-native LINS IESKF update path.
 *
 * This is the drop-in boundary for the one hot path of the reference
 * (ChaoqinRobotics/LINS---LiDAR-inertial-SLAM):
 *
 *     StateEstimator::performIESKF()              lins/include/StateEstimator.hpp:465-600
 *       findCorrespondingSurfFeatures()           lins/include/StateEstimator.hpp:829-953
 *       findCorrespondingCornerFeatures()         lins/include/StateEstimator.hpp:955-1063
 *       transformToStart()                        lins/include/StateEstimator.hpp:1066-1080
 *       H / residual assembly, gain, boxPlus      lins/include/StateEstimator.hpp:507-580
 *       Joseph covariance update                  lins/include/StateEstimator.hpp:594-598
 *
 * Everything crossing the boundary is a plain pointer, a size or a POD; no
 * C++/torch types.  Host pointers unless a parameter says "device".
 *
 * Conventions
 *   - lins_point mirrors pcl::PointXYZI (lins/include/parameters.h:52) without
 *     PCL's 16-byte padding lanes: x,y,z,intensity as four f32 = 16 B.
 *     `intensity` carries  ring + SCAN_PERIOD*relTime  (StateEstimator.hpp:649-650).
 *   - state vector (19 f64) mirrors filter::GlobalState (KalmanFilter.hpp:35-116):
 *       [0..2] rn_  [3..5] vn_  [6..9] qbn_ as (w,x,y,z)  [10..12] ba_
 *       [13..15] bw_  [16..18] gn_
 *   - covariance: 18x18 f64 row-major, error-state order pos,vel,att,acc,gyr,gra
 *     (KalmanFilter.hpp:40-45).
 *   - return value 0 = ok, negative = API / HIP error (lins_strerror()).
 *     Numerical divergence of the filter is NOT an error: it is reported in
 *     lins_result.diverged, exactly as the reference handles it internally
 *     (StateEstimator.hpp:552-570, 585-592).
 *   - a lins_ctx is not thread-safe (one HIP stream inside); distinct contexts
 *     may be driven from distinct threads / processes / GPUs.
 */
#ifndef LINS_IESKF_H_
#define LINS_IESKF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LINS_STATE_DIM 19  /* GlobalState as stored: p,v,q(wxyz),ba,bw,g           */
#define LINS_ERR_DIM 18    /* GlobalState::DIM_OF_STATE_ (KalmanFilter.hpp:39)     */
#define LINS_MAX_QUERY 1024  /* per cloud; reference caps: 192 sharp / 144 flat    */
#define LINS_MAX_RING 64     /* intensity in (-1, LINS_MAX_RING): int() in [0, 63]  */

/* error codes */
#define LINS_OK 0
#define LINS_E_ARG (-1)       /* bad argument (null pointer, negative size, ...)   */
#define LINS_E_HIP (-2)       /* HIP runtime error; see lins_last_hip_error()      */
#define LINS_E_CAPACITY (-3)  /* batch / cloud larger than the context was sized   */
#define LINS_E_INPUT (-4)     /* cloud violates the input contract (NaN, ring id)  */
#define LINS_E_NODEVICE (-5)  /* no usable gfx950 device                           */
#define LINS_E_STATE (-6)     /* call sequence error (run before upload, ...)      */
#define LINS_E_UNSUPPORTED (-7) /* this input cannot take the requested device path (nothing was run) */

typedef struct lins_point {
  float x, y, z, intensity;
} lins_point;

/* Parameters read on the path (parameters.h / exp_port.yaml:11-20).           */
typedef struct lins_params {
  int32_t num_iter;        /* NUM_ITER                    (yaml: 30)  SE:475    */
  int32_t icp_freq;        /* ICP_FREQ                    (yaml: 1)   SE:844,935*/
  int32_t fixed_iters;     /* 0 = reference semantics (stop on ||dx||<=1e-2,
                              SE:575-578).  1 = throughput mode: run exactly
                              num_iter iterations unless the filter diverges
                              (BASELINE.json configs[0]: "10 ESKF iterations") */
  int32_t reserved;
  double lidar_std;        /* LIDAR_STD                   (yaml: 0.01) SE:537   */
  double lidar_scale;      /* LIDAR_SCALE                 (yaml: 1)    SE:524   */
  double nearest_sq_dist;  /* NEAREST_FEATURE_SEARCH_SQ_DIST (yaml: 25) SE:851  */
  double scan_period;      /* SCAN_PERIOD                 (yaml: 0.1)  SE:1067  */
} lins_params;

/* One IESKF problem = what performIESKF() reads:
 * queries from scan_new_, targets from scan_last_, filter state + covariance. */
typedef struct lins_scan_pair {
  const lins_point* surf_flat;          /* scan_new_->surfPointsFlat_           */
  const lins_point* corner_sharp;       /* scan_new_->cornerPointsSharp_        */
  const lins_point* surf_less_flat_last;    /* scan_last_->surfPointsLessFlat_  */
  const lins_point* corner_less_sharp_last; /* scan_last_->cornerPointsLessSharp_ */
  int32_t n_surf_flat;
  int32_t n_corner_sharp;
  int32_t n_surf_last;
  int32_t n_corner_last;
  /* Bytes from one point of the four clouds to the next: 0 or 16 = packed lins_point arrays; 32 = the clouds as the
   * reference holds them, pcl::PointXYZI (parameters.h:52: x, y, z at bytes 0 / 4 / 8, intensity at byte 16) — pass
   * `reinterpret_cast<const lins_point*>(cloud->points.data())` and the library reads the 16 payload bytes of every
   * point where they lie: no repacking loop in the caller (lins_point_load below is the one definition of the access). */
  int32_t point_stride_bytes;
  int32_t reserved;
  double state[LINS_STATE_DIM];         /* filter_->state_                      */
  double cov[LINS_ERR_DIM * LINS_ERR_DIM]; /* filter_->covariance_              */
} lins_scan_pair;

/* Point i of a cloud of a lins_scan_pair with the given point_stride_bytes.                                           */
static inline lins_point lins_point_load(const lins_point* cloud, int32_t point_stride_bytes, int32_t i) {
  const float* f = (const float*)((const char*)cloud + (size_t)(point_stride_bytes == 32 ? 32 : 16) * (size_t)i);
  lins_point p;
  p.x = f[0], p.y = f[1], p.z = f[2], p.intensity = f[point_stride_bytes == 32 ? 4 : 3];
  return p;
}

/* What performIESKF() leaves behind (linState_, Pk_) + the flags the
 * reference keeps in locals (SE:471-474). When diverged != 0, `state`/`cov`
 * hold the un-updated filter state / covariance (SE:585-592 passes Pk_
 * un-updated) and the caller must run the ICP fallback
 * (lins_host_perform_ieskf() in lins_host.h does exactly that).                */
typedef struct lins_result {
  double state[LINS_STATE_DIM];
  double cov[LINS_ERR_DIM * LINS_ERR_DIM];
  double residual_norm;   /* ||residual_|| of the last executed iteration       */
  double update_norm;     /* ||updateVec_|| of the last executed iteration      */
  int32_t iters;          /* iterations executed (incl. the diverging one)      */
  int32_t converged;
  int32_t diverged;       /* 1 = residual blow-up (SE:566), 2 = NaN (SE:552)    */
  int32_t m_surf;         /* accepted surf rows in the last iteration           */
  int32_t m_corner;       /* accepted corner rows in the last iteration         */
  int32_t reserved[3];
} lins_result;

/* Fixed-size pose record used for the multi-GPU gather:
 * 19 f64 + 5 i32 + pad = 192 B.                                                */
typedef struct lins_pose_record {
  double state[LINS_STATE_DIM];
  double residual_norm;
  int32_t iters, converged, diverged, m_surf, m_corner, scan_id;
  int32_t pad[2];
} lins_pose_record;

/* Per-query output of one correspondence pass (A2/A3 of SURVEY.md §8a): what
 * the reference keeps in pointSearch{Surf,Corner}Ind{1,2,3} (SE:205-211) and
 * pushes to jacobianCoff{Surfs,Corns} (SE:943-949, 1053-1059).                 */
typedef struct lins_corr {
  int32_t ind1, ind2, ind3; /* closest / second / third target index, -1 = none
                               (corner rows: ind3 is always -1)                 */
  int32_t accepted;         /* 1 iff a row was pushed (s > 0.1 && res != 0)     */
  float coeff[4];           /* (s*jac.x, s*jac.y, s*jac.z, s*res) as f32        */
  float sel[4];             /* pointSel = transformToStart(query), f32          */
} lins_corr;

typedef struct lins_ctx lins_ctx;

/* --- lifetime ------------------------------------------------------------ */
/* max_batch scans, each with at most max_targets points per target cloud.    */
int lins_create(const lins_params* params, int device, int max_batch,
                int max_targets, lins_ctx** out);
void lins_destroy(lins_ctx* ctx);
const char* lins_strerror(int code);
const char* lins_last_hip_error(const lins_ctx* ctx);
/* Search strategy — every mode returns the same (exact) correspondences:
 *   "auto"   (default of the Python harness / bench) "mr" for batches larger than the
 *            device's CU count, "lds" otherwise
 *   "lds"    (ring x azimuth-column) grid of the targets resident in LDS, one 1024-thread
 *            workgroup per scan and CU, 3 lanes per query — shortest latency for one scan
 *   "mr"     multi-resident: the low part of the grid in LDS, the rest in a sorted global
 *            copy, 512 threads, 1 lane per query, two scans per CU — batch throughput
 *   "lds1"   whole grid in LDS, 384 threads, 1 lane per query
 *   "binned" the grid in global memory (any cloud size; automatic fallback of the above)
 *   "brute"  all-pairs search + the literal index walk (any input; the fallback for
 *            clouds that are not ring-sorted or carry ring ids >= 16)                   */
int lins_set_search(lins_ctx* ctx, const char* mode);

/* --- replaces StateEstimator::performIESKF() (SE:465-600) ----------------- */
/* synchronous, host buffers in and out                                        */
int lins_ieskf_update(lins_ctx* ctx, const lins_scan_pair* in, lins_result* out);
/* n independent scan pairs.  Batches of 512 and more go through in chunks of 256: a pool of host threads validates
 * and packs the next chunks while the copies and kernels of the previous ones run.  The kernel family ("auto" and the
 * eligibility fall-backs, lins_set_search) is then chosen PER CHUNK from that chunk's scans — a scan's result does not
 * depend on the family that computed it — so one oversized or unsorted scan sends its own chunk, not the batch, to the
 * any-size kernel; lins_last_search() and lins_last_kernel_ms() report the LAST chunk.  A contract violation in any
 * scan fails the whole call (results of earlier chunks are not delivered).                                         */
int lins_ieskf_update_batch(lins_ctx* ctx, int n, const lins_scan_pair* in,
                            lins_result* out);

/* --- staged (device-resident) form of the same call, for batches ---------- */
/* Takes the batch into HBM and builds the SEARCH INDEX of every scan's target clouds (grid_index_kernel: both clouds
 * counting-sorted into a (ring x azimuth-column) grid — a sorted copy + cell tables per scan) — the device counterpart
 * of kdtreeCorner_/kdtreeSurf_->setInputCloud(), which the reference runs where it produces the clouds
 * (updatePointCloud, SE:1156-1160; SE:363-364 for the first scan), not inside performIESKF.  Every later
 * lins_batch_run / lins_icp_update_batch / correspondence pass on this upload searches that index.  The host-buffer
 * entry points (lins_ieskf_update, lins_ieskf_update_batch) build it inside the call.                                 */
int lins_batch_upload(lins_ctx* ctx, int n, const lins_scan_pair* in);
/* Optional zero-copy staging.  The library sends host clouds from a pinned staging arena; lins_batch_map() lays that
 * arena out for n scans with the given cloud sizes (counts[4 s + c], c = 0 surf_flat, 1 corner_sharp, 2
 * surf_less_flat_last, 3 corner_less_sharp_last) and returns where each cloud belongs (clouds[4 s + c], packed 16-byte
 * points, writable until the next map / upload of this context).  A caller that writes its clouds there — e.g. as the
 * output buffers of its feature extraction — and passes those very pointers in the lins_scan_pair array of the next
 * lins_batch_upload / lins_ieskf_update(_batch) call (same n, same sizes) skips the library's copy into the arena:
 * the points are validated where they lie and sent by DMA.  Any other pointer is copied as before.                  */
int lins_batch_map(lins_ctx* ctx, int n, const int32_t* counts, lins_point** clouds);
/* HIP-event time (ms) of the index build of the last lins_batch_upload; 0 when the batch cannot take the grid kernels.
 * (The single-call entry points lins_ieskf_update(_batch) do not time it: LINS_E_STATE after them.)                   */
int lins_last_index_ms(lins_ctx* ctx, float* ms);
/* Runs the full IESKF loop for the uploaded batch on the context's stream.
 * d_poses: optional DEVICE pointer to n lins_pose_record (e.g. a torch tensor
 * that RCCL gathers afterwards); may be NULL. Asynchronous; lins_sync() waits. */
/* A batch with more scans than the device has workgroup slots runs every update in PARTS of a few iterations that hand
 * the loop state over through global memory; as many persistent workgroups as the device holds at once pull the parts
 * from a work queue inside the one launch (a part is queued by the workgroup that produced its input, and only while its
 * scan has not met the stop rule) — a shorter step, bit-identical results (ICP_FREQ 1; other batches run whole
 * updates).  Nothing in it depends on the order workgroups are dispatched in; a wait at the queue is bounded, and a
 * launch in which one ran out is reported by lins_sync() (LINS_E_HIP).  lins_last_cut() reports how the last run was cut. */
int lins_batch_run(lins_ctx* ctx, void* d_poses, int32_t scan_id_base);
/* parts = pieces every update of the last lins_batch_run() was cut into (1 = whole updates); queue_timeouts = waits at
 * the work queue that ran out over the life of the context (0 in a healthy process: the production counter of degraded
 * launches).                                                                                                          */
int lins_last_cut(const lins_ctx* ctx, int* parts, int* queue_timeouts);
int lins_sync(lins_ctx* ctx);
int lins_batch_download(lins_ctx* ctx, int n, lins_result* out);
/* HIP-event time (ms), on the context's stream, of the update kernel(s) of the last lins_batch_run(): the persistent
 * IESKF kernel — one launch, also when the run was cut into parts (lins_last_cut) — whose epilogue is the
 * Joseph covariance update in the "lds" / "mr" / "lds1" families; for "binned" / "brute" the update kernel and the
 * separate Joseph kernel.  After lins_ieskf_update_batch(): from the first chunk's kernels to the last chunk's, i.e.
 * including the index builds and the waits for the copies in between — the call's device time, not one kernel's.   */
int lins_last_kernel_ms(lins_ctx* ctx, float* ms);
/* HIP-event times (ms) of the update kernels of the last n lins_batch_run() calls, oldest first (n <= 64 and <= the
 * number of runs so far); waits for the newest of them.  lins_last_kernel_ms() is the n = 1 case.                 */
int lins_kernel_ms_history(lins_ctx* ctx, int n, float* ms);

/* --- pipelined staged mode: a stream of batches without a host wait per batch ---------------------------------- */
/* Off by default.  With it on, lins_pose_allgather() leaves the exchange of a run's pose records on the context's
 * communication stream, so that it travels beside the kernels of the NEXT lins_batch_run().  The caller alternates two
 * pose-record buffers (run k -> buffer k & 1) and may enqueue any number of runs before ONE lins_sync(), which waits
 * for everything.  While it is on, use only the staged calls (upload / run / pose_allgather / sync / download /
 * total_iters / kernel-time queries); each of them re-joins the streams where it must.                            */
int lins_set_pipelined(lins_ctx* ctx, int on);

/* Launch queues of lins_batch_run() for a batch beyond the device's workgroup slots (2 per CU: 512 on an MI355X).
 *   2 (default)  a run issued while the run before it is still in flight is launched as launches of at most that many
 *                scans — whole updates — dealt alternately to two HIP streams inside the context: the slots a launch
 *                leaves idle while its slowest updates finish are taken by the other queue's workgroups, of this run or
 *                of the next (successive runs are not joined; lins_sync / lins_batch_download and every other call wait
 *                for both queues): the higher throughput for runs queued back to back.  A run issued into an idle
 *                context is ONE launch whose updates are cut into parts that hand over inside the launch (rounds 3-5):
 *                the shorter latency for a run that is waited for.
 *   1            always the one-launch form.
 * A scan's results are the same bits either way.  (LINS_E_ARG for other values.)                                   */
int lins_set_launch_queues(lins_ctx* ctx, int queues);
/* GPU time (ms) the last n lins_batch_run() calls took together (first launch's start to last launch's end, both queues)
 * and the duration of each of their launches by its own queue's events (2 n values; 0 for the launch a one-launch run
 * does not have). */
int lins_runs_span_ms(lins_ctx* ctx, int n, float* ms);
int lins_launch_ms_history(lins_ctx* ctx, int n, float* ms);

/* --- multi-GPU: the one exchange step of the path --------------------------------------- */
/* Scan pairs are independent, so a batch shards over GPUs with no data-path collective; the results meet in ONE
 * all-gather of the fixed-size 192-byte pose records over RCCL (xGMI).  One lins_ctx (= one GPU) per rank.
 * librccl is dlopen()ed on first use (the copy already in the process when there is one), never linked:
 * LINS_E_UNSUPPORTED when it is not installed.
 *   lins_rccl_unique_id   one rank makes the id (LINS_RCCL_ID_BYTES bytes) and hands it to the others by whatever
 *                         bootstrap the application has (MPI, a file, torch.distributed's store ...)
 *   lins_rccl_init        collective over all ranks: ncclCommInitRank
 *   lins_pose_allgather   every rank contributes n_records records at d_local (DEVICE pointer), d_all (DEVICE pointer,
 *                         world x n_records records) receives them in rank order.  Stream-ordered after the last
 *                         lins_batch_run() (pipelined mode: on the context's communication stream, beside the next
 *                         run); no host wait — lins_sync() covers it.  Ragged shards: pad to the largest.
 *   lins_rccl_destroy     ncclCommDestroy (lins_destroy() does it too)                                              */
#define LINS_RCCL_ID_BYTES 128
int lins_rccl_unique_id(lins_ctx* ctx, void* id);
int lins_rccl_init(lins_ctx* ctx, const void* id, int rank, int world);
int lins_pose_allgather(lins_ctx* ctx, const void* d_local, int n_records, void* d_all);
int lins_rccl_destroy(lins_ctx* ctx);

/* The kernel family the last batch / pass actually ran ("mr", "lds", "lds1", "binned", "brute"): the requested
 * mode after "auto" and the eligibility fall-backs; "" before the first run.  Static storage.               */
const char* lins_last_search(const lins_ctx* ctx);
/* Algorithmic bytes of one iteration summed over the uploaded batch
 * (16*(Nsharp+Nflat+Nls+Nlf) + 8*19 + 8*28 per scan).          */
int lins_batch_bytes_per_iter(lins_ctx* ctx, uint64_t* bytes);
/* Total iterations executed by the last run (sum over scans).                 */
int lins_batch_total_iters(lins_ctx* ctx, uint64_t* iters);

/* --- one correspondence + residual/Jacobian pass (A2+A3) ------------------ */
/* Replaces findCorrespondingSurfFeatures / findCorrespondingCornerFeatures for
 * a given linearisation state (19 f64) and iteration counter. `surf` has
 * in->n_surf_flat entries, `corner` in->n_corner_sharp. Used by the host-side
 * solve of BASELINE.json configs[1] and by the ICP fallback (SE:1163-1196).
 * iter only selects the robust weight (iter >= icp_freq) — a fresh search is
 * always performed.                                                            */
int lins_correspondences(lins_ctx* ctx, const lins_scan_pair* in,
                         const double* lin_state, int iter, lins_corr* surf,
                         lins_corr* corner);
/* Same pass followed by the on-device reduction: 28 f64 sums over the accepted
 * rows h = (c0 c1 c2 a0 a1 a2), a = Rinvleft(-phi)^T (p x R^T c)  (SE:526-531):
 *   [0..20]  upper triangle of H^T H, row-major (00 01 .. 05 11 12 .. 55)
 *   [21..26] H^T r        [27] r^T r
 * i.e. the non-zero 6x6 block (columns pos 0-2, att 6-8) of the 18x18 normal
 * equations, and the accepted-row counts.                    */
int lins_reduce_pass(lins_ctx* ctx, const lins_scan_pair* in,
                     const double* lin_state, int iter, double* sums28,
                     int32_t* m_surf, int32_t* m_corner);

/* --- next row: the ICP fallback on the device --------------------------- */
/* estimateTransform / calculateTransformation (SE:1163-1320), the fallback performIESKF takes
 * when the filter diverges (SE:585-592): up to NUM_ITER rounds of {correspondences at the current
 * pose, 6-DoF Gauss-Newton step with the degeneracy projection of round 0, stop at 0.1 deg /
 * 0.1 cm}, all inside one kernel per scan (same grid and searches as the IESKF kernel).
 * in[].state = the pose to start from (the filter's); out[].state = that state with position and
 * attitude replaced, out[].cov = in[].cov, out[].iters / converged = rounds run / stop rule hit.
 * Returns LINS_E_UNSUPPORTED, with nothing run, when a scan cannot take the grid kernels
 * (unsorted rings, ring ids >= 16, > 12288 target points; with ICP_FREQ > 1 also > 512 queries) —
 * lins_host_perform_ieskf() then runs the same Gauss-Newton step on the host over
 * lins_correspondences() (which searches on every round).                                  */
int lins_icp_update_batch(lins_ctx* ctx, int n, const lins_scan_pair* in, lins_result* out);

/* --- next row after the update: updatePointCloud's re-projection --------- */
/* transformToEnd (SE:1083-1101) of whole clouds with the scan's final relative pose
 * (t = linState_.rn_, q = linState_.qbn_ as w,x,y,z): what turns the new scan's less-sharp /
 * less-flat clouds into the NEXT update's targets (SE:1122-1139), plus, optionally, the
 * YZX axis-swapped copy published to the mapping node (point.x = y, .y = z, .z = x,
 * SE:1125-1129).  Host buffers in and out; in == out_xyz is allowed (the reference
 * transforms in place).                                                                    */
typedef struct lins_reproject_job {
  const lins_point* in;
  lins_point* out_xyz;   /* re-projected cloud, XYZ axes                                   */
  lins_point* out_yzx;   /* NULL or the axis-swapped copy                                   */
  int32_t n;
  int32_t reserved;
  double t[3];
  double q[4];
} lins_reproject_job;
int lins_transform_to_end_batch(lins_ctx* ctx, int n_jobs, const lins_reproject_job* jobs);
/* HIP-event time (ms) of the re-projection kernel of the last call, and the bytes it moved
 * (16 B read + 16 or 32 B written per point).                                              */
int lins_last_reproject_stats(lins_ctx* ctx, float* kernel_ms, uint64_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* LINS_IESKF_H_ */
