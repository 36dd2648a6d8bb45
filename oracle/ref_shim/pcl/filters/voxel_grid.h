// <pcl/filters/voxel_grid.h> — STAND-IN (oracle/ref_shim/README.md).
//
// pcl::VoxelGrid<PointXYZI> as the reference uses it (StateEstimator.hpp:189, 822-824: leaf 0.2, defaults
// otherwise — every field downsampled, no minimum count).  Follows the documented algorithm of applyFilter: f32
// bounding box of the finite points, min_b = floor(min * inverse_leaf), cell index
// sum_k (floor(x_k * inverse_leaf_k) - min_b_k) * mul_k, points sorted by cell index with std::sort (so the order
// of the points inside one voxel — and with it the last bit of an f32 centroid — is whatever this libstdc++'s
// introsort leaves; the same would be true of a real PCL build), one output point per occupied cell in ascending
// cell order: the f32 sum of every field divided by the count.
#ifndef LINS_REF_SHIM_PCL_VOXEL_GRID_
#define LINS_REF_SHIM_PCL_VOXEL_GRID_
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace pcl {
template <typename PointT>
class VoxelGrid {
 public:
  VoxelGrid() {
    leaf_[0] = leaf_[1] = leaf_[2] = 0.f;
    inv_[0] = inv_[1] = inv_[2] = 0.f;
  }
  void setLeafSize(float lx, float ly, float lz) {
    leaf_[0] = lx;
    leaf_[1] = ly;
    leaf_[2] = lz;
    for (int k = 0; k < 3; ++k) inv_[k] = 1.0f / leaf_[k];
  }
  void setInputCloud(const typename PointCloud<PointT>::Ptr& cloud) { input_ = cloud; }
  void filter(PointCloud<PointT>& output) {
    output.clear();
    output.height = 1;
    output.is_dense = true;
    const std::vector<PointT>& in = input_->points;
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::numeric_limits<float>::max();
      mx[k] = -std::numeric_limits<float>::max();
    }
    bool any = false;
    for (size_t i = 0; i < in.size(); ++i) {
      if (!std::isfinite(in[i].x) || !std::isfinite(in[i].y) || !std::isfinite(in[i].z)) continue;
      const float v[3] = {in[i].x, in[i].y, in[i].z};
      for (int k = 0; k < 3; ++k) {
        mn[k] = std::min(mn[k], v[k]);
        mx[k] = std::max(mx[k], v[k]);
      }
      any = true;
    }
    if (!any) return;
    int min_b[3], max_b[3], div_b[3];
    for (int k = 0; k < 3; ++k) {
      min_b[k] = static_cast<int>(std::floor(mn[k] * inv_[k]));
      max_b[k] = static_cast<int>(std::floor(mx[k] * inv_[k]));
      div_b[k] = max_b[k] - min_b[k] + 1;
    }
    const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
    std::vector<Entry> idx;
    idx.reserve(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
      if (!std::isfinite(in[i].x) || !std::isfinite(in[i].y) || !std::isfinite(in[i].z)) continue;
      const int ijk0 = static_cast<int>(std::floor(in[i].x * inv_[0]) - static_cast<float>(min_b[0]));
      const int ijk1 = static_cast<int>(std::floor(in[i].y * inv_[1]) - static_cast<float>(min_b[1]));
      const int ijk2 = static_cast<int>(std::floor(in[i].z * inv_[2]) - static_cast<float>(min_b[2]));
      Entry e;
      e.idx = static_cast<unsigned int>(ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2]);
      e.cloud_point_index = static_cast<unsigned int>(i);
      idx.push_back(e);
    }
    std::sort(idx.begin(), idx.end());
    size_t first = 0;
    while (first < idx.size()) {
      size_t last = first + 1;
      while (last < idx.size() && idx[last].idx == idx[first].idx) ++last;
      float c[4] = {0.f, 0.f, 0.f, 0.f};
      for (size_t i = first; i < last; ++i) {
        const PointT& p = in[idx[i].cloud_point_index];
        c[0] += p.x;
        c[1] += p.y;
        c[2] += p.z;
        c[3] += p.intensity;
      }
      const float n = static_cast<float>(last - first);
      PointT o;
      o.x = c[0] / n;
      o.y = c[1] / n;
      o.z = c[2] / n;
      o.intensity = c[3] / n;
      output.push_back(o);
      first = last;
    }
  }

 private:
  struct Entry {
    unsigned int idx;
    unsigned int cloud_point_index;
    bool operator<(const Entry& o) const { return idx < o.idx; }
  };
  float leaf_[3], inv_[3];
  typename PointCloud<PointT>::Ptr input_;
};
}  // namespace pcl
#endif
