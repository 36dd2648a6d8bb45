// <pcl/point_cloud.h> — STAND-IN (oracle/ref_shim/README.md): the members of pcl::PointCloud the reference uses.
#ifndef LINS_REF_SHIM_PCL_POINT_CLOUD_
#define LINS_REF_SHIM_PCL_POINT_CLOUD_
#include <boost/shared_ptr.hpp>
#include <cstddef>
#include <cstdint>
#include <vector>
namespace pcl {
template <typename PointT>
class PointCloud {
 public:
  typedef boost::shared_ptr<PointCloud<PointT> > Ptr;
  typedef boost::shared_ptr<const PointCloud<PointT> > ConstPtr;
  PointCloud() : width(0), height(1), is_dense(true) {}
  void push_back(const PointT& p) {
    points.push_back(p);
    width = static_cast<std::uint32_t>(points.size());
    height = 1;
  }
  void clear() {
    points.clear();
    width = 0;
    height = 0;
  }
  void resize(std::size_t n) {
    points.resize(n);
    width = static_cast<std::uint32_t>(n);
    height = 1;
  }
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  PointCloud& operator+=(const PointCloud& rhs) {
    points.insert(points.end(), rhs.points.begin(), rhs.points.end());
    width = static_cast<std::uint32_t>(points.size());
    height = 1;
    return *this;
  }
  std::vector<PointT> points;
  std::uint32_t width, height;
  bool is_dense;
};
}  // namespace pcl
#endif
