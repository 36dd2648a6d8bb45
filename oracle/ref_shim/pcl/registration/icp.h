// <pcl/registration/icp.h> — STAND-IN (oracle/ref_shim/README.md): the loop-closure method of lidar_mapping_node.cpp names
// pcl::IterativeClosestPoint; it is never run by the _ref driver.  A name with the right shape, no behaviour.
#ifndef LINS_REF_SHIM_PCL_ICP_
#define LINS_REF_SHIM_PCL_ICP_
#include <pcl/common/common.h>
namespace pcl {
template <class S, class T>
struct IterativeClosestPoint {
  void setMaxCorrespondenceDistance(double) {}
  void setMaximumIterations(int) {}
  void setTransformationEpsilon(double) {}
  void setEuclideanFitnessEpsilon(double) {}
  void setRANSACIterations(int) {}
  template <class P>
  void setInputSource(const P&) {}
  template <class P>
  void setInputTarget(const P&) {}
  template <class C>
  void align(C&) {}
  bool hasConverged() const { return false; }
  double getFitnessScore() const { return 0.0; }
  Eigen::Matrix4f getFinalTransformation() const { return Eigen::Matrix4f(); }
};
}  // namespace pcl
#endif
