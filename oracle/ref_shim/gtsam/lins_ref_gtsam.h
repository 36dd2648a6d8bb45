// gtsam — STAND-IN (oracle/ref_shim/README.md).  lidar_mapping_node.cpp is ONE class; its scan-to-map optimisation
// (LM:1351-1652), which the _ref driver runs, never touches GTSAM, but the class's key-frame / loop-closure methods do,
// and they must parse.  These are names with the right shapes and no behaviour: nothing here is ever executed.
#ifndef LINS_REF_SHIM_GTSAM_
#define LINS_REF_SHIM_GTSAM_
#include <boost/shared_ptr.hpp>
#include <eigen3/Eigen/Dense>

#include <cstddef>
namespace gtsam {
typedef Eigen::VectorXd Vector;
struct Point3 {
  Point3() {}
  Point3(double, double, double) {}
  double x() const { return 0; }
  double y() const { return 0; }
  double z() const { return 0; }
};
struct Rot3 {
  static Rot3 RzRyRx(double, double, double) { return Rot3(); }
  double roll() const { return 0; }
  double pitch() const { return 0; }
  double yaw() const { return 0; }
};
struct Pose3 {
  Pose3() {}
  Pose3(const Rot3&, const Point3&) {}
  Pose3 between(const Pose3&) const { return Pose3(); }
  Point3 translation() const { return Point3(); }
  Rot3 rotation() const { return Rot3(); }
};
namespace noiseModel {
struct Diagonal {
  typedef boost::shared_ptr<Diagonal> shared_ptr;
  static shared_ptr Variances(const Vector&) { return shared_ptr(); }
};
}  // namespace noiseModel
template <class T>
struct PriorFactor {
  PriorFactor(std::size_t, const T&, const noiseModel::Diagonal::shared_ptr&) {}
};
template <class T>
struct BetweenFactor {
  BetweenFactor(std::size_t, std::size_t, const T&, const noiseModel::Diagonal::shared_ptr&) {}
};
struct NonlinearFactorGraph {
  template <class F>
  void add(const F&) {}
  void resize(std::size_t) {}
};
struct Values {
  template <class T>
  void insert(std::size_t, const T&) {}
  void clear() {}
  std::size_t size() const { return 0; }
  template <class T>
  T at(std::size_t) const { return T(); }
};
struct ISAM2Params {
  double relinearizeThreshold;
  int relinearizeSkip;
  ISAM2Params() : relinearizeThreshold(0), relinearizeSkip(0) {}
};
struct ISAM2 {
  explicit ISAM2(const ISAM2Params&) {}
  void update() {}
  void update(const NonlinearFactorGraph&) {}
  void update(const NonlinearFactorGraph&, const Values&) {}
  Values calculateEstimate() const { return Values(); }
};
}  // namespace gtsam
#endif
