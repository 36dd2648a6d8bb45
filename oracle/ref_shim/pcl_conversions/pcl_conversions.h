// <pcl_conversions/pcl_conversions.h> — STAND-IN (oracle/ref_shim/README.md): fromROSMsg / toROSMsg between the stand-in
// PointCloud2 (which carries the points as they are) and pcl::PointCloud<PointXYZI>.
#ifndef LINS_REF_SHIM_PCL_CONVERSIONS_
#define LINS_REF_SHIM_PCL_CONVERSIONS_
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl {
inline void fromROSMsg(const sensor_msgs::PointCloud2& msg, PointCloud<PointXYZI>& cloud) {
  cloud.points = msg.lins_ref_points;
  cloud.width = static_cast<std::uint32_t>(cloud.points.size());
  cloud.height = 1;
  cloud.is_dense = msg.is_dense;
}
inline void toROSMsg(const PointCloud<PointXYZI>& cloud, sensor_msgs::PointCloud2& msg) {
  msg.lins_ref_points = cloud.points;
  msg.is_dense = cloud.is_dense;
}
}  // namespace pcl
#endif
