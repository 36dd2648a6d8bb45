// "cloud_msgs/cloud_info.h" — STAND-IN for the header catkin generates from cloud_msgs/msg/cloud_info.msg
// (oracle/ref_shim/README.md): the message's fields with the C++ types roscpp's generator gives them
// (int32[] -> std::vector<int32_t>, bool[] -> std::vector<uint8_t>, uint32[] -> std::vector<uint32_t>,
// float32[] -> std::vector<float>).
//
// One departure, for defined behaviour: segmentedCloudColInd is a vector whose operator[] tolerates the index -1.
// extractFeatures' neighbour-masking loop (StateEstimator.hpp:763-777, 795-811) reads
// segmentedCloudColInd[ind + l] with ind + l == -1 whenever the point at position 0 is picked — which the
// reference does on every scan whose first point is a ground point, because ring 0's first sector starts at
// index 4 (IP:296: startRingIndex = -1 + 5) where cloudSmoothness_ still holds its default (value 0, ind 0;
// SE:656 starts at 5).  On glibc that read lands in the heap chunk header in front of the array and returns 0
// or the chunk size; here it returns a value that makes the column gap exceed 10, so the loop stops there —
// the same picks, minus the reference's stray write to cloudNeighborPicked_[-1].
#ifndef LINS_REF_SHIM_CLOUD_INFO_
#define LINS_REF_SHIM_CLOUD_INFO_
#include <boost/shared_ptr.hpp>
#include <std_msgs/Header.h>

#include <cstddef>
#include <cstdint>
#include <vector>
namespace lins_ref_shim {
template <typename T>
class VectorWithMinusOne {
 public:
  VectorWithMinusOne() : before_(static_cast<T>(0x7fffffff)) {}
  void assign(std::size_t n, const T& v) { d_.assign(n, v); }
  std::size_t size() const { return d_.size(); }
  T& operator[](std::ptrdiff_t i) { return i < 0 ? before_ : d_[static_cast<std::size_t>(i)]; }
  const T& operator[](std::ptrdiff_t i) const { return i < 0 ? before_ : d_[static_cast<std::size_t>(i)]; }

 private:
  std::vector<T> d_;
  T before_;
};
}  // namespace lins_ref_shim

namespace cloud_msgs {
struct cloud_info {
  typedef boost::shared_ptr<cloud_info> Ptr;
  typedef boost::shared_ptr<const cloud_info> ConstPtr;
  cloud_info() : startOrientation(0.f), endOrientation(0.f), orientationDiff(0.f) {}
  std_msgs::Header header;
  std::vector<int32_t> startRingIndex;
  std::vector<int32_t> endRingIndex;
  float startOrientation;
  float endOrientation;
  float orientationDiff;
  std::vector<uint8_t> segmentedCloudGroundFlag;
  lins_ref_shim::VectorWithMinusOne<uint32_t> segmentedCloudColInd;
  std::vector<float> segmentedCloudRange;
};
}  // namespace cloud_msgs
#endif
