// fusion_node_stub.cpp — a C++11 consumer of the drop-in boundary: INTEGRATION.md §2's binding, compiled as a real
// translation unit.  No Eigen / PCL / ROS in this image, so the reference's types are stubbed with the same members
// and layouts the binding touches (GlobalState KF:35-116, pcl::PointXYZI = 32 bytes, an 18 x 18 row-major
// covariance); the body of performIESKF() below is the INTEGRATION.md snippet (tests/test_consumer.py checks the
// two texts against each other).  The program reads one scan pair, runs StateEstimator::performIESKF() through
// lins_host_perform_ieskf and writes the posterior.  Built and run by tests/test_consumer.py on the GPU box:
//   g++ -std=c++11 -I include fusion_node_stub.cpp -L <pkg> -llins_ieskf -o fusion_node_stub
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lins_host.h"  // pulls in lins_ieskf.h

// ---- stubs of what the reference provides ---------------------------------------------------------------------
#define ROS_FATAL(...) (std::fprintf(stderr, __VA_ARGS__), std::fputc('\n', stderr), std::exit(2))
#define ROS_ERROR(...) (std::fprintf(stderr, __VA_ARGS__), std::fputc('\n', stderr))
#define ROS_WARN(...) (std::fprintf(stderr, __VA_ARGS__), std::fputc('\n', stderr))
struct V3D {
  double v[3];
  V3D() : v{0, 0, 0} {}
  V3D(double a, double b, double c) : v{a, b, c} {}
  double operator[](int i) const { return v[i]; }
};
struct Q4D {  // Eigen::Quaterniond: constructor order (w, x, y, z)
  double w_, x_, y_, z_;
  Q4D() : w_(1), x_(0), y_(0), z_(0) {}
  Q4D(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}
  double w() const { return w_; }
  double x() const { return x_; }
  double y() const { return y_; }
  double z() const { return z_; }
};
struct GlobalState {  // KalmanFilter.hpp:108-114
  V3D rn_, vn_;
  Q4D qbn_;
  V3D ba_, bw_, gn_;
};
struct Cov18 {  // Eigen::Matrix<double, 18, 18> stand-in, row-major storage
  double m[18 * 18];
};
struct StatePredictor {
  GlobalState state_;
  Cov18 covariance_;
  void update(const GlobalState& s, const Cov18& p) { state_ = s, covariance_ = p; }  // KF:354-357
};
struct alignas(16) PointType {  // pcl::PointXYZI: xyz + pad, intensity + pad = 32 bytes
  float x, y, z, pad0, intensity, pad1[3];
};
static_assert(sizeof(PointType) == 32, "pcl::PointXYZI layout");
struct Cloud {
  std::vector<PointType> points;
  size_t size() const { return points.size(); }
  const PointType& operator[](size_t i) const { return points[i]; }
};
struct Scan {
  Cloud *surfPointsFlat_, *cornerPointsSharp_, *surfPointsLessFlat_, *cornerPointsLessSharp_;
};
static const int NUM_ITER = 30, ICP_FREQ = 1, LINE_NUM = 16, SCAN_NUM = 1800;  // exp_port.yaml:9-20
static const double LIDAR_STD = 0.01, LIDAR_SCALE = 1.0, NEAREST_FEATURE_SEARCH_SQ_DIST = 25.0, SCAN_PERIOD = 0.1;

struct StateEstimator {
  StatePredictor* filter_;
  Scan *scan_new_, *scan_last_;
  GlobalState linState_;
  Cov18 Pk_;

  // BEGIN INTEGRATION.md section 2
  lins_ctx* gpu_ = nullptr;
  lins_params gpu_prm_;

  void initGpu() {
    gpu_prm_.num_iter = NUM_ITER;            // parameters.h / exp_port.yaml:11-20
    gpu_prm_.icp_freq = ICP_FREQ;
    gpu_prm_.fixed_iters = 0;                // reference stop rule (SE:575-578)
    gpu_prm_.reserved = 0;
    gpu_prm_.lidar_std = LIDAR_STD;
    gpu_prm_.lidar_scale = LIDAR_SCALE;
    gpu_prm_.nearest_sq_dist = NEAREST_FEATURE_SEARCH_SQ_DIST;
    gpu_prm_.scan_period = SCAN_PERIOD;
    if (lins_create(&gpu_prm_, /*device*/0, /*max_batch*/1, /*max_targets*/LINE_NUM * SCAN_NUM, &gpu_) != LINS_OK)
      ROS_FATAL("lins_create failed");
  }

  // pcl::PointXYZI is 32 bytes (xyz + pad, intensity + pad): the clouds are passed where they lie, the library reads
  // the 16 payload bytes of every point (point_stride_bytes = 32) — no repacking loop here.
  static const lins_point* points_of(const Cloud& c) { return reinterpret_cast<const lins_point*>(c.points.data()); }

  void performIESKF() {
    lins_scan_pair in;
    in.point_stride_bytes = (int32_t)sizeof(PointType);  in.reserved = 0;
    in.surf_flat = points_of(*scan_new_->surfPointsFlat_);                  in.n_surf_flat = (int)scan_new_->surfPointsFlat_->size();
    in.corner_sharp = points_of(*scan_new_->cornerPointsSharp_);            in.n_corner_sharp = (int)scan_new_->cornerPointsSharp_->size();
    in.surf_less_flat_last = points_of(*scan_last_->surfPointsLessFlat_);   in.n_surf_last = (int)scan_last_->surfPointsLessFlat_->size();
    in.corner_less_sharp_last = points_of(*scan_last_->cornerPointsLessSharp_);  in.n_corner_last = (int)scan_last_->cornerPointsLessSharp_->size();
    const GlobalState& x = filter_->state_;                               // KalmanFilter.hpp:35-116
    const double st[19] = {x.rn_[0], x.rn_[1], x.rn_[2], x.vn_[0], x.vn_[1], x.vn_[2],
                           x.qbn_.w(), x.qbn_.x(), x.qbn_.y(), x.qbn_.z(),
                           x.ba_[0], x.ba_[1], x.ba_[2], x.bw_[0], x.bw_[1], x.bw_[2], x.gn_[0], x.gn_[1], x.gn_[2]};
    std::copy(st, st + 19, in.state);
    copy_covariance_row_major(filter_->covariance_, in.cov);             // Eigen: Eigen::Map<Eigen::Matrix<double, 18, 18, Eigen::RowMajor>>(in.cov) = filter_->covariance_

    lins_result out;
    int32_t used_icp = 0;
    // GPU IESKF loop; on divergence the ICP fallback of SE:585-592, also one device kernel (lins_icp_update_batch)
    if (lins_host_perform_ieskf(gpu_, &gpu_prm_, &in, &out, &used_icp) != LINS_OK) {
      ROS_ERROR("IESKF update failed: %s", lins_last_hip_error(gpu_));
      return;
    }
    if (used_icp) ROS_WARN("======Using ICP Method======");              // SE:586

    linState_.rn_ = V3D(out.state[0], out.state[1], out.state[2]);
    linState_.vn_ = V3D(out.state[3], out.state[4], out.state[5]);
    linState_.qbn_ = Q4D(out.state[6], out.state[7], out.state[8], out.state[9]);   // (w, x, y, z)
    linState_.ba_ = V3D(out.state[10], out.state[11], out.state[12]);
    linState_.bw_ = V3D(out.state[13], out.state[14], out.state[15]);
    linState_.gn_ = V3D(out.state[16], out.state[17], out.state[18]);
    copy_covariance_row_major(out.cov, Pk_);                             // Eigen: Pk_ = Eigen::Map<const Eigen::Matrix<double, 18, 18, Eigen::RowMajor>>(out.cov)
    filter_->update(linState_, Pk_);                                     // SE:592 / SE:598
  }
  // END INTEGRATION.md section 2

  static void copy_covariance_row_major(const Cov18& c, double* out) { std::memcpy(out, c.m, sizeof c.m); }
  static void copy_covariance_row_major(const double* in, Cov18& c) { std::memcpy(c.m, in, sizeof c.m); }
};

// ---- driver: pair.bin (int32 counts[4], clouds as 16-byte points, state[19], cov[324]) -> posterior.bin ----
static bool read_cloud(std::FILE* f, int n, Cloud& c) {
  c.points.resize(n);
  for (int i = 0; i < n; ++i) {
    float v[4];
    if (std::fread(v, sizeof v, 1, f) != 1) return false;
    PointType p{};
    p.x = v[0], p.y = v[1], p.z = v[2], p.intensity = v[3];
    c.points[i] = p;
  }
  return true;
}

int main(int argc, char** argv) {
  if (argc < 3) return 64;
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 65;
  int32_t cnt[4];
  Cloud sf, cs, sl, cl;
  StatePredictor filter;
  double st[19];
  bool ok = std::fread(cnt, sizeof cnt, 1, f) == 1 && read_cloud(f, cnt[0], sf) && read_cloud(f, cnt[1], cs) &&
            read_cloud(f, cnt[2], sl) && read_cloud(f, cnt[3], cl) && std::fread(st, sizeof st, 1, f) == 1 &&
            std::fread(filter.covariance_.m, sizeof filter.covariance_.m, 1, f) == 1;
  std::fclose(f);
  if (!ok) return 66;
  filter.state_.rn_ = V3D(st[0], st[1], st[2]), filter.state_.vn_ = V3D(st[3], st[4], st[5]);
  filter.state_.qbn_ = Q4D(st[6], st[7], st[8], st[9]);
  filter.state_.ba_ = V3D(st[10], st[11], st[12]), filter.state_.bw_ = V3D(st[13], st[14], st[15]);
  filter.state_.gn_ = V3D(st[16], st[17], st[18]);
  Scan scan_new{&sf, &cs, nullptr, nullptr}, scan_last{nullptr, nullptr, &sl, &cl};
  StateEstimator se{};
  se.filter_ = &filter, se.scan_new_ = &scan_new, se.scan_last_ = &scan_last;
  se.initGpu();
  se.performIESKF();
  const GlobalState& x = filter.state_;
  const double o[19] = {x.rn_[0], x.rn_[1], x.rn_[2], x.vn_[0], x.vn_[1], x.vn_[2], x.qbn_.w(), x.qbn_.x(), x.qbn_.y(), x.qbn_.z(),
                        x.ba_[0], x.ba_[1], x.ba_[2], x.bw_[0], x.bw_[1], x.bw_[2], x.gn_[0], x.gn_[1], x.gn_[2]};
  std::FILE* g = std::fopen(argv[2], "wb");
  if (!g) return 67;
  std::fwrite(o, sizeof o, 1, g);
  std::fwrite(filter.covariance_.m, sizeof filter.covariance_.m, 1, g);
  std::fclose(g);
  lins_destroy(se.gpu_);
  return 0;
}
