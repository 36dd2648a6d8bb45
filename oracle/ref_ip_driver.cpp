// ref_ip_driver.cpp — the REFERENCE'S OWN image_projection_node.cpp (range-image projection, ground removal,
// segmentation: IP:191-415) compiled verbatim into oracle/_ref/liblins_ref.so, next to ref_driver.cpp.
//
// TEST INFRASTRUCTURE ONLY (oracle/): loaded by tests/ through oracle/ref.py; never by the product.
//
// The node is one class around a ros::NodeHandle: cloudHandler(msg) runs the whole stage and publishes the
// segmented cloud, the cloud_info message and the outlier cloud.  With the stand-in <ros/ros.h> of oracle/ref_shim a
// publication stores the message under its topic; this driver builds the node, hands cloudHandler() a raw cloud and
// reads the three messages back.  Not a line of the node is touched: its main() is compiled under another name
// (never called), its text comes from /root/reference/lins/src through the include path.
#include <parameters.h>

#include <cstring>

#include "../include/lins_host.h"

#define main lins_ref_image_projection_node_main
#include <image_projection_node.cpp>
#undef main

namespace parameter {
void readParameters(ros::NodeHandle&) {}  // (lins/src/lib/parameters.cpp: yaml reading — the driver sets the globals itself)
}

extern "C" {

// raw: n points in firing order (x, y, z, intensity).  cloud / range / col / ground: caller-allocated, LINS_CLOUD_MAX
// entries each; out is pointed at them.  -> 0, or -2 for a cloud the node itself would crash on (fewer than two points:
// findStartEndAngle reads points[size - 2]).
int ref_segment(const lins_point* raw, int n_raw, lins_point* cloud, float* range, uint32_t* col, uint8_t* ground,
                lins_segmented_scan* out) {
  if (!raw || !cloud || !range || !col || !ground || !out) return -1;
  if (n_raw < 2) return -2;
  parameter::LINE_NUM = LINS_LINE_NUM;
  parameter::SCAN_NUM = LINS_SCAN_NUM;
  ros::NodeHandle nh, pnh("~");
  ImageProjection node(nh, pnh);
  boost::shared_ptr<sensor_msgs::PointCloud2> msg(new sensor_msgs::PointCloud2());
  msg->lins_ref_points.resize(n_raw);
  for (int i = 0; i < n_raw; ++i) {
    msg->lins_ref_points[i].x = raw[i].x, msg->lins_ref_points[i].y = raw[i].y, msg->lins_ref_points[i].z = raw[i].z;
    msg->lins_ref_points[i].intensity = raw[i].intensity;
  }
  lins_ref_shim::Published<sensor_msgs::PointCloud2>::by_topic().clear();
  lins_ref_shim::Published<cloud_msgs::cloud_info>::by_topic().clear();
  node.cloudHandler(msg);
  const sensor_msgs::PointCloud2& seg = lins_ref_shim::Published<sensor_msgs::PointCloud2>::by_topic()["/segmented_cloud"];
  const sensor_msgs::PointCloud2& outl = lins_ref_shim::Published<sensor_msgs::PointCloud2>::by_topic()["/outlier_cloud"];
  const cloud_msgs::cloud_info& info = lins_ref_shim::Published<cloud_msgs::cloud_info>::by_topic()["/segmented_cloud_info"];
  const int n = (int)seg.lins_ref_points.size();
  if (n > LINS_CLOUD_MAX) return -3;
  for (int i = 0; i < n; ++i) {
    const pcl::PointXYZI& p = seg.lins_ref_points[i];
    cloud[i].x = p.x, cloud[i].y = p.y, cloud[i].z = p.z, cloud[i].intensity = p.intensity;
    range[i] = info.segmentedCloudRange[i];
    col[i] = info.segmentedCloudColInd[i];
    ground[i] = info.segmentedCloudGroundFlag[i] ? 1 : 0;
  }
  std::memset(out, 0, sizeof *out);
  out->cloud = cloud, out->range = range, out->col = col, out->ground = ground;
  out->n = n;
  for (int r = 0; r < LINS_LINE_NUM; ++r) out->start_ring[r] = info.startRingIndex[r], out->end_ring[r] = info.endRingIndex[r];
  out->start_ori = info.startOrientation, out->end_ori = info.endOrientation, out->ori_diff = info.orientationDiff;
  out->n_outlier = (int)outl.lins_ref_points.size();
  return 0;
}

}  // extern "C"
