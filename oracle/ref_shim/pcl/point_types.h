// <pcl/point_types.h> — STAND-IN (oracle/ref_shim/README.md).
#ifndef LINS_REF_SHIM_PCL_POINT_TYPES_
#define LINS_REF_SHIM_PCL_POINT_TYPES_
namespace pcl {
struct PointXYZI {
  float x, y, z, intensity;
  PointXYZI() : x(0.f), y(0.f), z(0.f), intensity(0.f) {}
};
}  // namespace pcl
#endif
