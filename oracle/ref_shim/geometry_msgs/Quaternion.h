// <geometry_msgs/Quaternion.h> and friends — STAND-IN (oracle/ref_shim/README.md): plain fields.
#ifndef LINS_REF_SHIM_GEOMETRY_MSGS_
#define LINS_REF_SHIM_GEOMETRY_MSGS_
namespace geometry_msgs {
struct Quaternion {
  double x, y, z, w;
  Quaternion() : x(0), y(0), z(0), w(1) {}
};
struct Vector3 {
  double x, y, z;
  Vector3() : x(0), y(0), z(0) {}
};
struct Point {
  double x, y, z;
  Point() : x(0), y(0), z(0) {}
};
struct Pose {
  Point position;
  Quaternion orientation;
};
struct PoseWithCovariance {
  Pose pose;
};
struct Twist {
  Vector3 linear, angular;
};
struct TwistWithCovariance {
  Twist twist;
};
}  // namespace geometry_msgs
#endif
