// <pcl/point_types.h> — STAND-IN (oracle/ref_shim/README.md).
#ifndef LINS_REF_SHIM_PCL_POINT_TYPES_
#define LINS_REF_SHIM_PCL_POINT_TYPES_
namespace pcl {
struct PointXYZI {
  float x, y, z, intensity;
  PointXYZI() : x(0.f), y(0.f), z(0.f), intensity(0.f) {}
};
}  // namespace pcl
// what a user-defined point type is written with (lidar_mapping_node.cpp:57-73 defines PointXYZIRPYT): the data members,
// no alignment requirement, no registration
#define PCL_ADD_POINT4D float x, y, z;
#define PCL_ADD_INTENSITY float intensity
#define EIGEN_ALIGN16
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, fields)
#endif
