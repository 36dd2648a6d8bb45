// <sensor_msgs/PointCloud2.h> — STAND-IN (oracle/ref_shim/README.md).  image_projection_node.cpp only moves clouds in
// and out of this message through pcl::fromROSMsg / pcl::toROSMsg (pcl_conversions stand-in), so the serialised
// byte buffer of the real message is replaced by the points themselves.
#ifndef LINS_REF_SHIM_SENSOR_MSGS_POINTCLOUD2_
#define LINS_REF_SHIM_SENSOR_MSGS_POINTCLOUD2_
#include <boost/shared_ptr.hpp>
#include <pcl/point_types.h>
#include <std_msgs/Header.h>

#include <vector>
namespace sensor_msgs {
struct PointCloud2 {
  std_msgs::Header header;
  std::vector<pcl::PointXYZI> lins_ref_points;
  bool is_dense;
  PointCloud2() : is_dense(true) {}
};
typedef boost::shared_ptr<PointCloud2> PointCloud2Ptr;
typedef boost::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
}  // namespace sensor_msgs
#endif
