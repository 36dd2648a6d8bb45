// ref_driver.cpp — extern "C" driver around the REFERENCE'S OWN SOURCES (oracle/_ref/liblins_ref.so).
//
// TEST INFRASTRUCTURE ONLY.  This translation unit #includes /root/reference/lins/include/StateEstimator.hpp —
// the reference's text, compiled from where it lies (oracle/Makefile, target _ref; never copied into this
// repository) — against the stand-in third-party headers of oracle/ref_shim/.  Every number this library returns
// is computed by the reference's own statements: fusion::StateEstimator::performIESKF (SE:465-600),
// findCorrespondingSurfFeatures / ...CornerFeatures (SE:829-1063), transformToStart / transformToEnd
// (SE:1066-1101), estimateTransform / calculateTransformation (SE:1163-1320), undistortPcl ... extractFeatures
// (SE:619-827), filter::GlobalState::boxPlus / boxMinus (KF:71-94), filter::StatePredictor::predict / reset
// (KF:125-186, 320-352), math_utils (MU).  The driver only moves data in and out of the reference's public members.
//
// What the driver has to infer, because performIESKF keeps it in locals (SE:471-474):
//   diverged   from the ROS_WARN the reference emits (SE:560 "...NaN...", SE:567 "System diverges...")
//   iters      from the number of kd-tree queries the reference issued before that point
//              (ICP_FREQ == 1: every iteration queries each feature once; otherwise by re-running with
//              NUM_ITER = 1, 2, ... until the result stops changing)
//   converged  updateVecNorm_ <= 1e-2 of the last completed iteration (SE:575-578), member of the class
// The reference's two out-of-bounds reads are refused, not executed: an empty target cloud
// (pointSearchSqDis[0] of an empty vector, SE:851 / SE:977) and more query features than target points (the
// `j < surfPointsFlatNum` walk of SE:859 / SE:983 then indexes past the target cloud).
#include <StateEstimator.hpp>

#include <chrono>
#include <cstring>
#include <thread>

#include "../include/lins_host.h"
#include "../include/lins_ieskf.h"
#include <lins_ref_shim/events.h>

#include "ref_params.inc"

namespace {

using fusion::StateEstimator;

void set_params(const lins_params* p) {
  parameter::NUM_ITER = p->num_iter;
  parameter::ICP_FREQ = p->icp_freq;
  parameter::LIDAR_STD = p->lidar_std;
  parameter::LIDAR_SCALE = p->lidar_scale;
  parameter::NEAREST_FEATURE_SEARCH_SQ_DIST = p->nearest_sq_dist;
  parameter::SCAN_PERIOD = p->scan_period;
}

void to_state(const double* s, filter::GlobalState& g) {
  g.rn_ = V3D(s[0], s[1], s[2]);
  g.vn_ = V3D(s[3], s[4], s[5]);
  g.qbn_ = Q4D(s[6], s[7], s[8], s[9]);
  g.ba_ = V3D(s[10], s[11], s[12]);
  g.bw_ = V3D(s[13], s[14], s[15]);
  g.gn_ = V3D(s[16], s[17], s[18]);
}
void from_state(const filter::GlobalState& g, double* s) {
  for (int k = 0; k < 3; ++k) {
    s[k] = g.rn_(k);
    s[3 + k] = g.vn_(k);
    s[10 + k] = g.ba_(k);
    s[13 + k] = g.bw_(k);
    s[16 + k] = g.gn_(k);
  }
  s[6] = g.qbn_.w();
  s[7] = g.qbn_.x();
  s[8] = g.qbn_.y();
  s[9] = g.qbn_.z();
}

void fill(pcl::PointCloud<PointType>::Ptr cloud, const lins_point* p, int n) {
  cloud->clear();
  for (int i = 0; i < n; ++i) {
    PointType q;
    q.x = p[i].x;
    q.y = p[i].y;
    q.z = p[i].z;
    q.intensity = p[i].intensity;
    cloud->push_back(q);
  }
}
void dump(const pcl::PointCloud<PointType>& cloud, lins_point* out, int cap, int32_t* n) {
  *n = static_cast<int32_t>(cloud.points.size());
  for (int i = 0; i < *n && i < cap; ++i) {
    out[i].x = cloud.points[i].x;
    out[i].y = cloud.points[i].y;
    out[i].z = cloud.points[i].z;
    out[i].intensity = cloud.points[i].intensity;
  }
}

// inputs on which the reference itself reads out of bounds (see the header comment)
bool reference_reads_oob(const lins_scan_pair* in) {
  if (in->point_stride_bytes == 32) return true;  // (refused like them: this checker takes packed points)
  if (in->n_surf_flat > 0 && (in->n_surf_last == 0 || in->n_surf_flat > in->n_surf_last)) return true;
  if (in->n_corner_sharp > 0 && (in->n_corner_last == 0 || in->n_corner_sharp > in->n_corner_last)) return true;
  return false;
}

// A StateEstimator in the state processScan() (SE:436-463) hands to performIESKF(): feature clouds of the new and
// the last scan in place, kd-trees built over the last scan's clouds (SE:1156-1160), filter state and covariance.
struct Rig {
  StateEstimator est;
  explicit Rig(const lins_scan_pair* in) {
    est.preintegration_ = new integration::IntegrationBase(V3D(0, 0, 0), V3D(0, 0, 0), parameter::INIT_BA,
                                                           parameter::INIT_BW);  // ~StateEstimator deletes it
    est.status_ = StateEstimator::STATUS_RUNNING;
    fill(est.scan_new_->surfPointsFlat_, in->surf_flat, in->n_surf_flat);
    fill(est.scan_new_->cornerPointsSharp_, in->corner_sharp, in->n_corner_sharp);
    fill(est.scan_last_->surfPointsLessFlat_, in->surf_less_flat_last, in->n_surf_last);
    fill(est.scan_last_->cornerPointsLessSharp_, in->corner_less_sharp_last, in->n_corner_last);
    est.kdtreeCorner_->setInputCloud(est.scan_last_->cornerPointsLessSharp_);
    est.kdtreeSurf_->setInputCloud(est.scan_last_->surfPointsLessFlat_);
    to_state(in->state, est.filter_->state_);
    for (int i = 0; i < 18; ++i)
      for (int j = 0; j < 18; ++j) est.filter_->covariance_(i, j) = in->cov[i * 18 + j];
    est.linState_.setIdentity();
  }
};

struct RunInfo {
  int diverged;       // 0, 1 = residual blow-up (SE:566), 2 = NaN (SE:552)
  long queries_main;  // kd-tree queries issued by the IESKF loop itself (before any fallback)
  int icp_iters;      // iteration at which the fallback reported convergence, -1 = not reported
};

RunInfo run_perform(StateEstimator& est) {
  lins_ref_shim::reset_events();
  est.performIESKF();
  RunInfo r{0, lins_ref_shim::kdtree_queries(), -1};
  for (const lins_ref_shim::Event& e : lins_ref_shim::events()) {
    if (e.text.find("Because of NaN") != std::string::npos) {
      r.diverged = 2;
      r.queries_main = e.queries;
    } else if (e.text.find("System diverges") != std::string::npos) {
      r.diverged = 1;
      r.queries_main = e.queries;
    } else if (e.text.find("System Converges after") != std::string::npos) {
      r.icp_iters = std::atoi(e.text.c_str() + std::strlen("System Converges after "));
    }
  }
  return r;
}

void collect(StateEstimator& est, const RunInfo& info, const lins_scan_pair* in, lins_result* out, double* dx18) {
  std::memset(out, 0, sizeof(*out));
  from_state(est.filter_->state_, out->state);
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) out->cov[i * 18 + j] = est.filter_->covariance_(i, j);
  out->diverged = info.diverged;
  out->update_norm = est.updateVecNorm_;
  out->converged = (!info.diverged && est.updateVecNorm_ <= 1e-2) ? 1 : 0;
  const int per_iter = in->n_surf_flat + in->n_corner_sharp;
  out->iters = -1;  // unknown (the caller falls back to replaying with growing NUM_ITER)
  if (parameter::ICP_FREQ == 1 && per_iter > 0) out->iters = static_cast<int32_t>(info.queries_main / per_iter);
  if (!info.diverged) {  // members of the last executed iteration (the fallback overwrites them)
    out->residual_norm = est.residual_.norm();
    out->m_surf = static_cast<int32_t>(est.keypointSurfs_->points.size());
    out->m_corner = static_cast<int32_t>(est.keypointCorns_->points.size());
  }
  if (dx18)
    for (int k = 0; k < 18; ++k) dx18[k] = est.updateVec_[k];
}

bool same_bits(const PointType& a, const lins_point& b) {
  return std::memcmp(&a.x, &b.x, 4) == 0 && std::memcmp(&a.y, &b.y, 4) == 0 && std::memcmp(&a.z, &b.z, 4) == 0 &&
         std::memcmp(&a.intensity, &b.intensity, 4) == 0;
}

}  // namespace

extern "C" {

const char* ref_describe() {
  return "reference sources compiled verbatim: lins/include/StateEstimator.hpp + KalmanFilter.hpp + math_utils.h + "
         "integrationBase.h + sensor_utils.hpp + parameters.h against oracle/ref_shim";
}

// performIESKF() (SE:465-600) on one scan pair, divergence branch (SE:585-592) included.
// dx18 (optional): updateVec_ of the last executed iteration.
// returns 0, -1 bad argument, -2 the reference would read out of bounds on this input.
int ref_perform_ieskf(const lins_params* prm, const lins_scan_pair* in, lins_result* out, double* dx18) {
  if (!prm || !in || !out) return -1;
  if (reference_reads_oob(in)) return -2;
  set_params(prm);
  Rig rig(in);
  RunInfo info = run_perform(rig.est);
  collect(rig.est, info, in, out, dx18);
  return 0;
}

// The same over n independent pairs on `threads` threads (results in order); rc[i] per pair as above.
int ref_perform_ieskf_batch(const lins_params* prm, int n, const lins_scan_pair* in, lins_result* out, int32_t* rc,
                            int threads) {
  if (!prm || n < 0 || (n > 0 && (!in || !out || !rc))) return -1;
  set_params(prm);
  if (threads < 1) threads = 1;
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([=]() {
      for (int i = t; i < n; i += threads) {
        if (reference_reads_oob(&in[i])) {
          rc[i] = -2;
          continue;
        }
        Rig rig(&in[i]);
        RunInfo info = run_perform(rig.est);
        collect(rig.est, info, &in[i], &out[i], nullptr);
        rc[i] = 0;
      }
    });
  for (auto& th : pool) th.join();
  return 0;
}

// findCorrespondingSurfFeatures + findCorrespondingCornerFeatures (SE:829-1063) at one linearisation state.
// Fills the reference's pointSearch*Ind arrays, the pushed coefficient rows and transformToStart(query).
// returns -3 when iter is not a search iteration (iter % ICP_FREQ != 0: the reference would reuse stale indices).
int ref_correspondences(const lins_params* prm, const lins_scan_pair* in, const double* lin_state, int iter,
                        lins_corr* surf, lins_corr* corner) {
  if (!prm || !in || !lin_state) return -1;
  if (reference_reads_oob(in)) return -2;
  set_params(prm);
  if (iter % parameter::ICP_FREQ != 0) return -3;
  Rig rig(in);
  StateEstimator& est = rig.est;
  to_state(lin_state, est.linState_);
  est.keypointSurfs_->clear();
  est.jacobianCoffSurfs->clear();
  est.keypointCorns_->clear();
  est.jacobianCoffCorns->clear();
  est.findCorrespondingSurfFeatures(est.scan_last_, est.scan_new_, est.keypointSurfs_, est.jacobianCoffSurfs, iter);
  est.findCorrespondingCornerFeatures(est.scan_last_, est.scan_new_, est.keypointCorns_, est.jacobianCoffCorns, iter);
  for (int pass = 0; pass < 2; ++pass) {
    const bool is_surf = pass == 0;
    const int n = is_surf ? in->n_surf_flat : in->n_corner_sharp;
    const lins_point* q = is_surf ? in->surf_flat : in->corner_sharp;
    lins_corr* o = is_surf ? surf : corner;
    if (!o) continue;
    const pcl::PointCloud<PointType>& keys = is_surf ? *est.keypointSurfs_ : *est.keypointCorns_;
    const pcl::PointCloud<PointType>& rows = is_surf ? *est.jacobianCoffSurfs : *est.jacobianCoffCorns;
    size_t r = 0;
    for (int i = 0; i < n; ++i) {
      std::memset(&o[i], 0, sizeof(lins_corr));
      o[i].ind1 = static_cast<int32_t>(is_surf ? est.pointSearchSurfInd1[i] : est.pointSearchCornerInd1[i]);
      o[i].ind2 = static_cast<int32_t>(is_surf ? est.pointSearchSurfInd2[i] : est.pointSearchCornerInd2[i]);
      o[i].ind3 = is_surf ? static_cast<int32_t>(est.pointSearchSurfInd3[i]) : -1;
      PointType raw, sel;
      raw.x = q[i].x;
      raw.y = q[i].y;
      raw.z = q[i].z;
      raw.intensity = q[i].intensity;
      est.transformToStart(&raw, &sel);
      o[i].sel[0] = sel.x;
      o[i].sel[1] = sel.y;
      o[i].sel[2] = sel.z;
      o[i].sel[3] = sel.intensity;
      // rows are pushed in query order together with the raw query point (SE:948-949, 1058-1059): a query owns
      // the next row iff it could have pushed one and the next key is this very point
      const bool could = is_surf ? (o[i].ind2 >= 0 && o[i].ind3 >= 0) : (o[i].ind2 >= 0);
      if (could && r < keys.points.size() && same_bits(keys.points[r], q[i])) {
        o[i].accepted = 1;
        o[i].coeff[0] = rows.points[r].x;
        o[i].coeff[1] = rows.points[r].y;
        o[i].coeff[2] = rows.points[r].z;
        o[i].coeff[3] = rows.points[r].intensity;
        ++r;
      }
    }
    if (r != keys.points.size()) return -4;  // the row attribution above failed
  }
  return 0;
}

// estimateTransform (SE:1163-1196) with calculateTransformation (SE:1198-1320): t[3], q[4] = (w,x,y,z) in / out.
// iters_run: the iteration index the reference reported convergence at + 1, or NUM_ITER.
int ref_icp(const lins_params* prm, const lins_scan_pair* in, double* t, double* q, int32_t* iters_run) {
  if (!prm || !in || !t || !q) return -1;
  if (reference_reads_oob(in)) return -2;
  set_params(prm);
  Rig rig(in);
  V3D tt(t[0], t[1], t[2]);
  Q4D qq(q[0], q[1], q[2], q[3]);
  lins_ref_shim::reset_events();
  rig.est.estimateTransform(rig.est.scan_last_, rig.est.scan_new_, tt, qq);
  int it = parameter::NUM_ITER;
  for (const lins_ref_shim::Event& e : lins_ref_shim::events())
    if (e.text.find("System Converges after") != std::string::npos)
      it = std::atoi(e.text.c_str() + std::strlen("System Converges after ")) + 1;
  if (iters_run) *iters_run = it;
  for (int k = 0; k < 3; ++k) t[k] = tt(k);
  q[0] = qq.w();
  q[1] = qq.x();
  q[2] = qq.y();
  q[3] = qq.z();
  return 0;
}

// transformToStart (SE:1066-1080) / transformToEnd (SE:1083-1101) of n points at linState_ = lin_state.
int ref_transform(const lins_params* prm, const double* lin_state, int to_end, int n, const lins_point* in,
                  lins_point* out) {
  if (!prm || !lin_state || (n > 0 && (!in || !out))) return -1;
  set_params(prm);
  static thread_local StateEstimator* est = nullptr;
  if (!est) {
    est = new StateEstimator();
    est->preintegration_ = nullptr;
  }
  to_state(lin_state, est->linState_);
  for (int i = 0; i < n; ++i) {
    PointType a, b;
    a.x = in[i].x;
    a.y = in[i].y;
    a.z = in[i].z;
    a.intensity = in[i].intensity;
    if (to_end)
      est->transformToEnd(&a, &b);
    else
      est->transformToStart(&a, &b);
    out[i].x = b.x;
    out[i].y = b.y;
    out[i].z = b.z;
    out[i].intensity = b.intensity;
  }
  return 0;
}

// processPCL's feature stage (SE:289-292): undistortPcl, calculateSmoothness, markOccludedPoints, extractFeatures
// on one segmented scan (what image_projection_node publishes).  Output arrays have lins_features' capacities;
// undistorted (optional) has in->n entries.
int ref_extract_features(const lins_params* prm, const lins_segmented_scan* in, lins_features* out,
                         lins_point* undistorted) {
  if (!prm || !in || !out) return -1;
  set_params(prm);
  StateEstimator est;
  est.preintegration_ = nullptr;
  pcl::PointCloud<PointType>::Ptr cloud(new pcl::PointCloud<PointType>());
  pcl::PointCloud<PointType>::Ptr outlier(new pcl::PointCloud<PointType>());
  fill(cloud, in->cloud, in->n);
  cloud_msgs::cloud_info info;
  info.startRingIndex.assign(in->start_ring, in->start_ring + LINS_LINE_NUM);
  info.endRingIndex.assign(in->end_ring, in->end_ring + LINS_LINE_NUM);
  info.startOrientation = in->start_ori;
  info.endOrientation = in->end_ori;
  info.orientationDiff = in->ori_diff;
  // the node allocates the three arrays at LINE_NUM * SCAN_NUM and fills the first n (IP:125-130)
  info.segmentedCloudGroundFlag.assign(LINS_CLOUD_MAX, 0);
  info.segmentedCloudColInd.assign(LINS_CLOUD_MAX, 0);
  info.segmentedCloudRange.assign(LINS_CLOUD_MAX, 0.f);
  for (int i = 0; i < in->n; ++i) {
    info.segmentedCloudGroundFlag[i] = in->ground[i];
    info.segmentedCloudColInd[i] = in->col[i];
    info.segmentedCloudRange[i] = in->range[i];
  }
  est.scan_new_->setPointCloud(0.0, cloud, info, outlier);
  // calculateSmoothness initialises cloudNeighborPicked_ / cloudLabel_ only on [5, n - 5) (SE:671-672) but
  // extractFeatures reads both outside that range (position 0 through the default Smooth at index 4, labels up to
  // ep): the reference sees whatever `new Scan()` returned — zero pages while glibc still mmaps the 0.5 MB object.
  // Start from that state instead of this process's heap garbage.
  std::memset(est.scan_new_->cloudNeighborPicked_, 0, sizeof(est.scan_new_->cloudNeighborPicked_));
  std::memset(est.scan_new_->cloudLabel_, 0, sizeof(est.scan_new_->cloudLabel_));
  est.undistortPcl(est.scan_new_);
  est.calculateSmoothness(est.scan_new_);
  est.markOccludedPoints(est.scan_new_);
  est.extractFeatures(est.scan_new_);
  dump(*est.scan_new_->cornerPointsSharp_, out->corner_sharp, 192, &out->n_corner_sharp);
  dump(*est.scan_new_->cornerPointsLessSharp_, out->corner_less_sharp, 1920, &out->n_corner_less_sharp);
  dump(*est.scan_new_->surfPointsFlat_, out->surf_flat, 1024, &out->n_surf_flat);
  dump(*est.scan_new_->surfPointsLessFlat_, out->surf_less_flat, LINS_CLOUD_MAX, &out->n_surf_less_flat);
  out->n_segmented = in->n;
  out->n_outlier = in->n_outlier;
  if (undistorted) {
    int32_t n = 0;
    dump(*est.scan_new_->undistPointCloud_, undistorted, in->n, &n);
  }
  return 0;
}

// ---- KalmanFilter.hpp / math_utils.h -----------------------------------------------------------------------
void ref_quat2axis(const double* q, double* axis3) {
  V3D a = math_utils::Quat2axis(Q4D(q[0], q[1], q[2], q[3]));
  for (int k = 0; k < 3; ++k) axis3[k] = a(k);
}
void ref_axis2quat(const double* axis3, double* q) {
  Q4D r = math_utils::axis2Quat(V3D(axis3[0], axis3[1], axis3[2]));
  q[0] = r.w();
  q[1] = r.x();
  q[2] = r.y();
  q[3] = r.z();
}
void ref_rinvleft(const double* axis3, double* m9) {
  M3D r = math_utils::Rinvleft(V3D(axis3[0], axis3[1], axis3[2]));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m9[i * 3 + j] = r(i, j);
}
void ref_rpy2quat(const double* rpy, double* q) {
  Q4D r = math_utils::rpy2Quat(V3D(rpy[0], rpy[1], rpy[2]));
  q[0] = r.w();
  q[1] = r.x();
  q[2] = r.y();
  q[3] = r.z();
}
void ref_box_plus(const double* s19, const double* dx18, double* out19) {
  filter::GlobalState a, b;
  to_state(s19, a);
  Eigen::Matrix<double, 18, 1> dx;
  for (int k = 0; k < 18; ++k) dx(k) = dx18[k];
  a.boxPlus(dx, b);
  from_state(b, out19);
}
void ref_box_minus(const double* a19, const double* b19, double* out18) {
  filter::GlobalState a, b;
  to_state(a19, a);
  to_state(b19, b);
  Eigen::Matrix<double, 18, 1> dx;
  a.boxMinus(b, dx);
  for (int k = 0; k < 18; ++k) out18[k] = dx(k);
}

// StatePredictor (KF:118-380) driven the way lins_filter_* of include/lins_host.h is: op 0 = initialization(time 0,
// rn 0, vn, ba, bw) [KF:225-234: identity attitude, covariance type 0], then n predict() calls (KF:125-186), then
// optionally reset(1) (KF:320-352).  imu = n rows of (dt, acc xyz, gyr xyz).
int ref_filter_run(const lins_filter_params* fp, const double* vn, const double* ba, const double* bw, int n,
                   const double* imu, int reset1, double* state19, double* cov324) {
  if (!fp || !vn || !ba || !bw || (n > 0 && !imu) || !state19 || !cov324) return -1;
  parameter::ACC_N = fp->acc_n;
  parameter::GYR_N = fp->gyr_n;
  parameter::ACC_W = fp->acc_w;
  parameter::GYR_W = fp->gyr_w;
  parameter::INIT_POS_STD = V3D(fp->init_pos_std[0], fp->init_pos_std[1], fp->init_pos_std[2]);
  parameter::INIT_VEL_STD = V3D(fp->init_vel_std[0], fp->init_vel_std[1], fp->init_vel_std[2]);
  parameter::INIT_ATT_STD = V3D(fp->init_att_std[0], fp->init_att_std[1], fp->init_att_std[2]);
  parameter::INIT_ACC_STD = V3D(fp->init_acc_std[0], fp->init_acc_std[1], fp->init_acc_std[2]);
  parameter::INIT_GYR_STD = V3D(fp->init_gyr_std[0], fp->init_gyr_std[1], fp->init_gyr_std[2]);
  filter::StatePredictor f;
  f.flag_init_imu_ = false;  // the reference leaves both flags uninitialised until initialization() (KF:121, 378-379)
  f.flag_init_state_ = false;
  f.initialization(0.0, V3D(0, 0, 0), V3D(vn[0], vn[1], vn[2]), V3D(ba[0], ba[1], ba[2]), V3D(bw[0], bw[1], bw[2]));
  for (int i = 0; i < n; ++i) {
    const double* r = imu + 7 * i;
    f.predict(r[0], V3D(r[1], r[2], r[3]), V3D(r[4], r[5], r[6]), true);
  }
  if (reset1) f.reset(1);
  from_state(f.state_, state19);
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) cov324[i * 18 + j] = f.covariance_(i, j);
  return 0;
}

// Timed loop for bench.py's cpu_baseline (kind "reference"): performIESKF over n pairs on `threads` threads.
// Rigs (clouds + kd-trees, what updatePointCloud left behind for the real node) are built outside the timed region.
int ref_bench(const lins_params* prm, int n, const lins_scan_pair* in, int threads, double* seconds,
              uint64_t* iters) {
  if (!prm || !in || n <= 0 || !seconds || !iters) return -1;
  for (int i = 0; i < n; ++i)
    if (reference_reads_oob(&in[i])) return -2;
  set_params(prm);
  if (threads < 1) threads = 1;
  std::vector<Rig*> rigs(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) rigs[i] = new Rig(&in[i]);
  std::vector<uint64_t> its(static_cast<size_t>(threads), 0);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t]() {
      for (int i = t; i < n; i += threads) {
        RunInfo info = run_perform(rigs[i]->est);
        const int per_iter = in[i].n_surf_flat + in[i].n_corner_sharp;
        if (per_iter > 0 && parameter::ICP_FREQ == 1) its[t] += static_cast<uint64_t>(info.queries_main / per_iter);
      }
    });
  for (auto& th : pool) th.join();
  auto t1 = std::chrono::steady_clock::now();
  *seconds = std::chrono::duration<double>(t1 - t0).count();
  *iters = 0;
  for (uint64_t v : its) *iters += v;
  for (Rig* r : rigs) delete r;
  return 0;
}

}  // extern "C"
