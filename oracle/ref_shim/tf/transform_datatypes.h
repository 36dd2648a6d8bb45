// <tf/transform_datatypes.h> — STAND-IN (oracle/ref_shim/README.md): the names lidar_mapping_node.cpp's publishing and
// message-handling methods use.  Never executed by the _ref driver (it runs the scan-to-map optimisation only); the
// conversions return identities.
#ifndef LINS_REF_SHIM_TF_DATATYPES_
#define LINS_REF_SHIM_TF_DATATYPES_
#include <geometry_msgs/Quaternion.h>
#include <std_msgs/Header.h>

#include <string>
namespace tf {
struct Quaternion {
  double x_, y_, z_, w_;
  Quaternion() : x_(0), y_(0), z_(0), w_(1) {}
  Quaternion(double x, double y, double z, double w) : x_(x), y_(y), z_(z), w_(w) {}
};
struct Vector3 {
  double x_, y_, z_;
  Vector3() : x_(0), y_(0), z_(0) {}
  Vector3(double x, double y, double z) : x_(x), y_(y), z_(z) {}
};
struct Matrix3x3 {
  Matrix3x3() {}
  explicit Matrix3x3(const Quaternion&) {}
  void getRPY(double& r, double& p, double& y) const { r = p = y = 0; }
};
inline geometry_msgs::Quaternion createQuaternionMsgFromRollPitchYaw(double, double, double) { return geometry_msgs::Quaternion(); }
inline geometry_msgs::Quaternion createQuaternionMsgFromYaw(double) { return geometry_msgs::Quaternion(); }
inline void quaternionMsgToTF(const geometry_msgs::Quaternion&, Quaternion&) {}
struct StampedTransform {
  ros::Time stamp_;
  std::string frame_id_, child_frame_id_;
  void setRotation(const Quaternion&) {}
  void setOrigin(const Vector3&) {}
};
}  // namespace tf
#endif
