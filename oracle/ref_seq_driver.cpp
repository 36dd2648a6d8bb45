// ref_seq_driver.cpp — the drop-in boundary exercised INSIDE the reference's own state machine, over a scan sequence
// (oracle/_ref/liblins_ref_seq.so).
//
// TEST INFRASTRUCTURE ONLY.  Like ref_driver.cpp this translation unit #includes the reference's StateEstimator.hpp from
// where it lies and compiles it against the stand-in headers of oracle/ref_shim/ — every statement of processImu,
// processPCL, processFirstScan / SecondScan, processScan, integrateTransformation, updatePointCloud, the kd-tree
// rebuilds and filter_->reset(1) (SE:242-463, 602-617, 1083-1161; KF:118-380) is the reference's own.  ONE thing is
// swapped: the call `performIESKF();` in processScan (SE:443) reaches a hook.  The swap is a macro around the #include,
// the reference's text is not touched:
//
//     #define performIESKF() performIESKF(); void performIESKF_reference()
//
//   SE:465  `void performIESKF() {` ...   becomes   `void performIESKF(); void performIESKF_reference() {` ...
//           — the reference's body, under another name, plus the declaration of the hook member;
//   SE:443  `performIESKF();`            becomes   `performIESKF(); void performIESKF_reference();`
//           — the call, now to the hook, followed by a (harmless) block-scope function declaration.
//
// The hook, StateEstimator::performIESKF() below, is INTEGRATION.md section 2 written with the reference's real types
// (pcl::PointCloud, GlobalState, Eigen matrices of the stand-in): it packs what performIESKF reads, calls the function
// pointer it was given — lins_host_perform_ieskf of liblins_ieskf.so with its lins_ctx, handed over by the test; this
// library never links the product — and writes linState_, Pk_ and filter_->update() back.  With no function pointer the
// hook runs performIESKF_reference(): the unmodified reference, the run the swapped one is compared with.
#define performIESKF() performIESKF(); void performIESKF_reference()
#include <StateEstimator.hpp>
#undef performIESKF

#include <cstring>
#include <vector>

#include "../include/lins_host.h"
#include "../include/lins_ieskf.h"
#include <lins_ref_shim/events.h>

#include "ref_params.inc"

namespace {

// the boundary's entry point as a pointer: int lins_host_perform_ieskf(lins_ctx*, const lins_params*, const lins_scan_pair*,
// lins_result*, int32_t*) — `user` is the lins_ctx (or whatever a CPU stand-in wants)
typedef int (*perform_fn)(void* user, const lins_params* prm, const lins_scan_pair* in, lins_result* out, int32_t* used_icp);

struct Update {  // what one processScan's update reported (SE:465-600)
  int32_t ran, iters, converged, diverged, used_icp, m_surf, m_corner, rc;
  double update_norm;
};

struct Seq {
  fusion::StateEstimator est;
  lins_params prm;
  perform_fn fn = nullptr;
  void* user = nullptr;
  Update last{};
};
thread_local Seq* g_seq = nullptr;  // the sequence whose processPCL is running on this thread (the hook is a member: no argument)

void pack(const pcl::PointCloud<PointType>& c, std::vector<lins_point>& out) {
  out.resize(c.points.size());
  for (size_t i = 0; i < c.points.size(); ++i) out[i] = lins_point{c.points[i].x, c.points[i].y, c.points[i].z, c.points[i].intensity};
}

}  // namespace

// ---- the hook: INTEGRATION.md section 2 ------------------------------------------------------------------------------
void fusion::StateEstimator::performIESKF() {
  Seq& q = *g_seq;
  q.last = Update{};
  q.last.ran = 1;
  if (!q.fn) {  // the unmodified reference; its flags inferred as ref_driver.cpp does
    lins_ref_shim::reset_events();
    const size_t per_iter = scan_new_->surfPointsFlat_->points.size() + scan_new_->cornerPointsSharp_->points.size();
    performIESKF_reference();
    long queries = lins_ref_shim::kdtree_queries();
    for (const lins_ref_shim::Event& e : lins_ref_shim::events()) {
      if (e.text.find("Because of NaN") != std::string::npos) q.last.diverged = 2, queries = e.queries;
      if (e.text.find("System diverges") != std::string::npos) q.last.diverged = 1, queries = e.queries;
      if (e.text.find("Using ICP Method") != std::string::npos) q.last.used_icp = 1;
    }
    q.last.iters = parameter::ICP_FREQ == 1 && per_iter ? (int32_t)(queries / (long)per_iter) : -1;
    q.last.update_norm = updateVecNorm_;
    q.last.converged = (!q.last.diverged && updateVecNorm_ <= 1e-2) ? 1 : 0;
    if (!q.last.diverged) q.last.m_surf = (int32_t)keypointSurfs_->points.size(), q.last.m_corner = (int32_t)keypointCorns_->points.size();
    return;
  }
  static thread_local std::vector<lins_point> sf, cs, sl, cl;
  pack(*scan_new_->surfPointsFlat_, sf), pack(*scan_new_->cornerPointsSharp_, cs);
  pack(*scan_last_->surfPointsLessFlat_, sl), pack(*scan_last_->cornerPointsLessSharp_, cl);
  lins_scan_pair in;
  in.point_stride_bytes = 0, in.reserved = 0;  // (packed 16-byte points: this checker packs, like the round-4 binding did)
  in.surf_flat = sf.data(), in.n_surf_flat = (int)sf.size();
  in.corner_sharp = cs.data(), in.n_corner_sharp = (int)cs.size();
  in.surf_less_flat_last = sl.data(), in.n_surf_last = (int)sl.size();
  in.corner_less_sharp_last = cl.data(), in.n_corner_last = (int)cl.size();
  const GlobalState& x = filter_->state_;  // KalmanFilter.hpp:35-116
  const double st[19] = {x.rn_[0], x.rn_[1], x.rn_[2], x.vn_[0], x.vn_[1], x.vn_[2], x.qbn_.w(), x.qbn_.x(), x.qbn_.y(), x.qbn_.z(),
                         x.ba_[0], x.ba_[1], x.ba_[2], x.bw_[0], x.bw_[1], x.bw_[2], x.gn_[0], x.gn_[1], x.gn_[2]};
  std::memcpy(in.state, st, sizeof st);
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) in.cov[i * 18 + j] = filter_->covariance_(i, j);  // (Eigen::Map<... RowMajor>(in.cov) = covariance_)
  lins_result out;
  int32_t used_icp = 0;
  q.last.rc = q.fn(q.user, &q.prm, &in, &out, &used_icp);
  if (q.last.rc != LINS_OK) return;  // (the node would ROS_ERROR and keep its prediction)
  linState_.rn_ = V3D(out.state[0], out.state[1], out.state[2]);
  linState_.vn_ = V3D(out.state[3], out.state[4], out.state[5]);
  linState_.qbn_ = Q4D(out.state[6], out.state[7], out.state[8], out.state[9]);  // (w, x, y, z)
  linState_.ba_ = V3D(out.state[10], out.state[11], out.state[12]);
  linState_.bw_ = V3D(out.state[13], out.state[14], out.state[15]);
  linState_.gn_ = V3D(out.state[16], out.state[17], out.state[18]);
  Pk_.resize(18, 18);
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) Pk_(i, j) = out.cov[i * 18 + j];
  filter_->update(linState_, Pk_);  // SE:592 / SE:598
  q.last.iters = out.iters, q.last.converged = out.converged, q.last.diverged = out.diverged, q.last.used_icp = used_icp;
  q.last.m_surf = out.m_surf, q.last.m_corner = out.m_corner, q.last.update_norm = out.update_norm;
}

extern "C" {

// One record per processed scan.
struct ref_seq_record {
  int32_t status;        // StateEstimator::status_ after the scan (0 INIT, 1 FIRST_SCAN, 3 RUNNING: SE:170-176)
  int32_t ran_update;    // processScan reached performIESKF
  int32_t iters, converged, diverged, used_icp, m_surf, m_corner, rc;
  int32_t n_corner_sharp, n_corner_less_sharp, n_surf_flat, n_surf_less_flat;  // the LAST scan's clouds after the slide
  int32_t pad;
  double update_norm;
  double global_state[19];  // globalState_: p, v, q (w x y z), ba, bw, g
  double lin_state[19];     // linState_ (the relative transform of this scan)
  double filter_state[19];  // filter_->state_ after reset(1)
  double cov_trace;         // trace of filter_->covariance_
  double filter_cov[324];   // filter_->covariance_ after reset(1), row-major
  double imu_last[6];       // filter_->acc_last, gyr_last (what the next predict's mid-point rule starts from, KF:136-146)
};

void* ref_seq_create(const lins_params* prm) {
  if (!prm) return nullptr;
  parameter::NUM_ITER = prm->num_iter, parameter::ICP_FREQ = prm->icp_freq;
  parameter::LIDAR_STD = prm->lidar_std, parameter::LIDAR_SCALE = prm->lidar_scale;
  parameter::NEAREST_FEATURE_SEARCH_SQ_DIST = prm->nearest_sq_dist, parameter::SCAN_PERIOD = prm->scan_period;
  Seq* s = new Seq();
  s->prm = *prm;
  s->est.preintegration_ = nullptr;  // (allocated by processFirstScan, SE:356; the destructor deletes it)
  return s;
}
void ref_seq_destroy(void* h) { delete static_cast<Seq*>(h); }
// fn = nullptr: the unmodified reference.  Otherwise every performIESKF of the sequence goes through fn(user, ...).
void ref_seq_set_hook(void* h, void* fn, void* user) {
  Seq* s = static_cast<Seq*>(h);
  s->fn = reinterpret_cast<perform_fn>(fn), s->user = user;
}
// LinsFusion::processPointClouds feeds the IMU samples between two scans one by one (EC:226-234)
int ref_seq_imu(void* h, double dt, const double* acc, const double* gyr) {
  if (!h || !acc || !gyr) return -1;
  static_cast<Seq*>(h)->est.processImu(dt, V3D(acc[0], acc[1], acc[2]), V3D(gyr[0], gyr[1], gyr[2]));
  return 0;
}
// ... and then hands the scan over: processPCL(time, last imu, segmented cloud, cloud_info, outlier cloud) (EC:240-242)
int ref_seq_scan(void* h, double time, const double* acc, const double* gyr, const lins_segmented_scan* in, ref_seq_record* rec) {
  if (!h || !in || !rec || !acc || !gyr) return -1;
  Seq* s = static_cast<Seq*>(h);
  pcl::PointCloud<PointType>::Ptr cloud(new pcl::PointCloud<PointType>()), outlier(new pcl::PointCloud<PointType>());
  for (int i = 0; i < in->n; ++i) {
    PointType p;
    p.x = in->cloud[i].x, p.y = in->cloud[i].y, p.z = in->cloud[i].z, p.intensity = in->cloud[i].intensity;
    cloud->push_back(p);
  }
  cloud_msgs::cloud_info info;
  info.startRingIndex.assign(in->start_ring, in->start_ring + LINS_LINE_NUM);
  info.endRingIndex.assign(in->end_ring, in->end_ring + LINS_LINE_NUM);
  info.startOrientation = in->start_ori, info.endOrientation = in->end_ori, info.orientationDiff = in->ori_diff;
  info.segmentedCloudGroundFlag.assign(LINS_CLOUD_MAX, 0);  // (the node allocates the three arrays at LINE_NUM * SCAN_NUM, IP:125-130)
  info.segmentedCloudColInd.assign(LINS_CLOUD_MAX, 0);
  info.segmentedCloudRange.assign(LINS_CLOUD_MAX, 0.f);
  for (int i = 0; i < in->n; ++i)
    info.segmentedCloudGroundFlag[i] = in->ground[i], info.segmentedCloudColInd[i] = in->col[i], info.segmentedCloudRange[i] = in->range[i];
  // the reference reads cloudNeighborPicked_ / cloudLabel_ of a fresh Scan outside what it initialises (ref_driver.cpp
  // ref_extract_features): start from zeros, what a fresh process's `new Scan()` holds
  std::memset(s->est.scan_new_->cloudNeighborPicked_, 0, sizeof(s->est.scan_new_->cloudNeighborPicked_));
  std::memset(s->est.scan_new_->cloudLabel_, 0, sizeof(s->est.scan_new_->cloudLabel_));
  sensor_utils::Imu imu(time, V3D(acc[0], acc[1], acc[2]), V3D(gyr[0], gyr[1], gyr[2]));
  g_seq = s;
  s->last = Update{};
  s->est.processPCL(time, imu, cloud, info, outlier);
  g_seq = nullptr;
  std::memset(rec, 0, sizeof *rec);
  rec->status = (int32_t)s->est.status_;
  rec->ran_update = s->last.ran, rec->iters = s->last.iters, rec->converged = s->last.converged, rec->diverged = s->last.diverged;
  rec->used_icp = s->last.used_icp, rec->m_surf = s->last.m_surf, rec->m_corner = s->last.m_corner, rec->rc = s->last.rc;
  rec->update_norm = s->last.update_norm;
  rec->n_corner_sharp = (int32_t)s->est.scan_last_->cornerPointsSharp_->points.size();
  rec->n_corner_less_sharp = (int32_t)s->est.scan_last_->cornerPointsLessSharp_->points.size();
  rec->n_surf_flat = (int32_t)s->est.scan_last_->surfPointsFlat_->points.size();
  rec->n_surf_less_flat = (int32_t)s->est.scan_last_->surfPointsLessFlat_->points.size();
  auto put = [](const filter::GlobalState& g, double* o) {
    for (int k = 0; k < 3; ++k) o[k] = g.rn_(k), o[3 + k] = g.vn_(k), o[10 + k] = g.ba_(k), o[13 + k] = g.bw_(k), o[16 + k] = g.gn_(k);
    o[6] = g.qbn_.w(), o[7] = g.qbn_.x(), o[8] = g.qbn_.y(), o[9] = g.qbn_.z();
  };
  put(s->est.globalState_, rec->global_state);
  put(s->est.linState_, rec->lin_state);
  put(s->est.filter_->state_, rec->filter_state);
  if (s->est.filter_->covariance_.rows() == 18)
    for (int i = 0; i < 18; ++i) {
      rec->cov_trace += s->est.filter_->covariance_(i, i);
      for (int j = 0; j < 18; ++j) rec->filter_cov[i * 18 + j] = s->est.filter_->covariance_(i, j);
    }
  for (int k = 0; k < 3; ++k) rec->imu_last[k] = s->est.filter_->acc_last(k), rec->imu_last[3 + k] = s->est.filter_->gyr_last(k);
  return 0;
}

}  // extern "C"
