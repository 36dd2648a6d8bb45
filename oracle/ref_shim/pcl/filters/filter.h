// <pcl/filters/filter.h> — STAND-IN (oracle/ref_shim/README.md): pcl::removeNaNFromPointCloud as PCL 1.8 defines it
// (common/impl/filter.hpp:46-97): a dense cloud is copied as it is, otherwise the points with a non-finite x, y or z
// are dropped and the cloud is marked dense.
#ifndef LINS_REF_SHIM_PCL_FILTER_
#define LINS_REF_SHIM_PCL_FILTER_
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include <cmath>
#include <vector>
namespace pcl {
template <typename PointT>
void removeNaNFromPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, std::vector<int>& index) {
  if (&in != &out) {
    out.points.resize(in.points.size());
    out.is_dense = in.is_dense;
  }
  index.resize(in.points.size());
  std::size_t j = 0;
  if (in.is_dense) {
    if (&in != &out) out.points = in.points;
    for (j = 0; j < out.points.size(); ++j) index[j] = static_cast<int>(j);
  } else {
    for (std::size_t i = 0; i < in.points.size(); ++i) {
      if (!std::isfinite(in.points[i].x) || !std::isfinite(in.points[i].y) || !std::isfinite(in.points[i].z)) continue;
      out.points[j] = in.points[i];
      index[j] = static_cast<int>(i);
      ++j;
    }
    if (j != in.points.size()) {
      out.points.resize(j);
      index.resize(j);
    }
    out.height = 1;
    out.width = static_cast<std::uint32_t>(j);
    out.is_dense = true;
  }
}
}  // namespace pcl
#endif
