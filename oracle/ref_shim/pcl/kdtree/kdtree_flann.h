// <pcl/kdtree/kdtree_flann.h> — STAND-IN (oracle/ref_shim/README.md).
//
// pcl::KdTreeFLANN<PointXYZI>::nearestKSearch as the reference calls it (StateEstimator.hpp:847, 973; k = 1): an
// EXACT nearest-neighbour query over (x, y, z) with FLANN's L2_Simple<float> distance — the squared differences are
// accumulated in f32 in dimension order, ((dx*dx + dy*dy) + dz*dz).  The tree snapshots the cloud at
// setInputCloud() time, like PCL (it converts the cloud into its own float array).  Implemented as an exact k-d
// tree with a leaf scan; equal distances resolve to the LOWEST index (which leaf FLANN's own tree visits first
// depends on its build and is not a documented property — stated, not pinned).
// Also counts its queries (thread-local) so a driver can tell how many search rounds a call ran.
#ifndef LINS_REF_SHIM_PCL_KDTREE_FLANN_
#define LINS_REF_SHIM_PCL_KDTREE_FLANN_
#include <lins_ref_shim/events.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include <algorithm>
#include <limits>
#include <utility>
#include <vector>

namespace pcl {
template <typename PointT>
class KdTreeFLANN {
 public:
  typedef boost::shared_ptr<KdTreeFLANN<PointT> > Ptr;
  typedef typename PointCloud<PointT>::Ptr CloudPtr;
  KdTreeFLANN() {}

  void setInputCloud(const CloudPtr& cloud) {
    const int n = static_cast<int>(cloud->points.size());
    pts_.resize(static_cast<size_t>(n) * 3);
    order_.resize(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
      pts_[3 * i + 0] = cloud->points[i].x;
      pts_[3 * i + 1] = cloud->points[i].y;
      pts_[3 * i + 2] = cloud->points[i].z;
      order_[i] = i;
    }
    nodes_.clear();
    if (n > 0) build(0, n);
  }

  int nearestKSearch(const PointT& p, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const {
    ++lins_ref_shim::kdtree_queries();
    const float q[3] = {p.x, p.y, p.z};
    std::vector<std::pair<float, int> > best;  // ascending (distance, index), at most k
    if (!nodes_.empty() && k > 0) search(0, q, k, best);
    k_indices.resize(best.size());
    k_sqr_distances.resize(best.size());
    for (size_t i = 0; i < best.size(); ++i) {
      k_sqr_distances[i] = best[i].first;
      k_indices[i] = best[i].second;
    }
    return static_cast<int>(best.size());
  }

  // every point within `radius` (by the same f32 distance), ascending (distance, index); brute force — the reference
  // only calls it from its key-frame bookkeeping, which the _ref drivers never run
  int radiusSearch(const PointT& p, double radius, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances, unsigned max_nn = 0) const {
    std::vector<std::pair<float, int> > hits;
    const float r2 = static_cast<float>(radius * radius);
    for (std::size_t i = 0; i < order_.size(); ++i) {
      const float dx = p.x - pts_[3 * i], dy = p.y - pts_[3 * i + 1], dz = p.z - pts_[3 * i + 2];
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (d <= r2) hits.push_back(std::make_pair(d, static_cast<int>(i)));
    }
    std::sort(hits.begin(), hits.end());
    if (max_nn && hits.size() > max_nn) hits.resize(max_nn);
    k_indices.resize(hits.size()), k_sqr_distances.resize(hits.size());
    for (std::size_t i = 0; i < hits.size(); ++i) k_sqr_distances[i] = hits[i].first, k_indices[i] = hits[i].second;
    return static_cast<int>(hits.size());
  }

 private:
  struct Node {
    int lo, hi;       // range in order_
    int left, right;  // children, -1 = leaf
    int dim;
    float split;
    float bmin[3], bmax[3];
  };
  enum { kLeaf = 12 };

  int build(int lo, int hi) {
    Node nd;
    nd.lo = lo;
    nd.hi = hi;
    nd.left = nd.right = -1;
    nd.dim = 0;
    nd.split = 0.f;
    for (int d = 0; d < 3; ++d) {
      nd.bmin[d] = std::numeric_limits<float>::infinity();
      nd.bmax[d] = -std::numeric_limits<float>::infinity();
    }
    for (int i = lo; i < hi; ++i)
      for (int d = 0; d < 3; ++d) {
        const float v = pts_[3 * order_[i] + d];
        nd.bmin[d] = std::min(nd.bmin[d], v);
        nd.bmax[d] = std::max(nd.bmax[d], v);
      }
    const int id = static_cast<int>(nodes_.size());
    nodes_.push_back(nd);
    if (hi - lo > kLeaf) {
      int dim = 0;
      for (int d = 1; d < 3; ++d)
        if (nd.bmax[d] - nd.bmin[d] > nd.bmax[dim] - nd.bmin[dim]) dim = d;
      const int mid = (lo + hi) / 2;
      const std::vector<float>& P = pts_;
      std::nth_element(order_.begin() + lo, order_.begin() + mid, order_.begin() + hi, [&P, dim](int a, int b) {
        const float va = P[3 * a + dim], vb = P[3 * b + dim];
        return va < vb || (va == vb && a < b);
      });
      const int l = build(lo, mid);
      const int r = build(mid, hi);
      nodes_[id].left = l;
      nodes_[id].right = r;
      nodes_[id].dim = dim;
    }
    return id;
  }

  // a lower bound of the L2_Simple<float> distance from q to anything inside the node's box, as a double that is
  // strictly below any f32-rounded distance to a point of the box (scaled down by a few ulps of slack)
  static double box_lower_bound(const Node& nd, const float* q) {
    double s = 0.0;
    for (int d = 0; d < 3; ++d) {
      double g = 0.0;
      if (q[d] < nd.bmin[d]) g = double(nd.bmin[d]) - double(q[d]);
      if (q[d] > nd.bmax[d]) g = double(q[d]) - double(nd.bmax[d]);
      s += g * g;
    }
    return s * (1.0 - 1e-6);
  }

  void search(int id, const float* q, int k, std::vector<std::pair<float, int> >& best) const {
    const Node& nd = nodes_[id];
    if (static_cast<int>(best.size()) == k && box_lower_bound(nd, q) > double(best.back().first)) return;
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; ++i) {
        const int j = order_[i];
        float dist = 0.f;  // L2_Simple<float>: result += diff * diff, dimension by dimension
        for (int d = 0; d < 3; ++d) {
          const float diff = q[d] - pts_[3 * j + d];
          dist += diff * diff;
        }
        const std::pair<float, int> cand(dist, j);
        if (static_cast<int>(best.size()) < k) {
          best.insert(std::upper_bound(best.begin(), best.end(), cand), cand);
        } else if (cand < best.back()) {
          best.pop_back();
          best.insert(std::upper_bound(best.begin(), best.end(), cand), cand);
        }
      }
      return;
    }
    const double dl = box_lower_bound(nodes_[nd.left], q), dr = box_lower_bound(nodes_[nd.right], q);
    if (dl <= dr) {
      search(nd.left, q, k, best);
      search(nd.right, q, k, best);
    } else {
      search(nd.right, q, k, best);
      search(nd.left, q, k, best);
    }
  }

  std::vector<float> pts_;
  std::vector<int> order_;
  std::vector<Node> nodes_;
};
}  // namespace pcl
#endif
