// <pcl/common/common.h> — STAND-IN (oracle/ref_shim/README.md).  StateEstimator.hpp uses nothing of it.
// lidar_mapping_node.cpp's key-frame / loop-closure methods (never executed by the _ref driver, but they must parse) name
// pcl::rad2deg, copyPointCloud, transformPointCloud, getTransformation, getTranslationAndEulerAngles and Eigen::Affine3f;
// rad2deg is also used by the LM step's stop rule (LM:1622-1627) and is the real conversion.
#ifndef LINS_REF_SHIM_PCL_COMMON_
#define LINS_REF_SHIM_PCL_COMMON_
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include <cmath>
namespace Eigen {
struct Affine3f {  // (a name only: loop closure is never run here)
  Affine3f() {}
  template <class T>
  Affine3f(const T&) {}
  template <class T>
  Affine3f& operator=(const T&) {
    return *this;
  }
  Affine3f operator*(const Affine3f&) const { return Affine3f(); }
};
struct Matrix4f {};
}  // namespace Eigen
namespace pcl {
inline float rad2deg(float alpha) { return alpha * 57.29578f; }   // pcl/common/impl/angles.hpp:46-50
inline double rad2deg(double alpha) { return alpha * 57.29578; }  // (the same literal, as PCL has it)
template <class P>
void copyPointCloud(const PointCloud<P>& in, PointCloud<P>& out) {
  out = in;
}
template <class P, class T>
void transformPointCloud(const PointCloud<P>& in, PointCloud<P>& out, const T&) {
  out = in;
}
inline Eigen::Affine3f getTransformation(float, float, float, float, float, float) { return Eigen::Affine3f(); }
inline void getTranslationAndEulerAngles(const Eigen::Affine3f&, float& x, float& y, float& z, float& roll, float& pitch, float& yaw) {
  x = y = z = roll = pitch = yaw = 0.f;
}
}  // namespace pcl
#endif
