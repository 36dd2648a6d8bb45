// <opencv2/opencv.hpp> — STAND-IN (oracle/ref_shim/README.md).  parameters.h names cv::FileStorage in two declarations
// (readV3D / readQ4D), which the _ref build never defines or calls; image_projection_node.cpp keeps its range / label /
// ground images in cv::Mat and uses exactly: the (rows, cols, type, Scalar) constructor with CV_32F / CV_8S / CV_32S,
// Scalar::all, assignment, and at<T>(row, col).
#ifndef LINS_REF_SHIM_OPENCV_
#define LINS_REF_SHIM_OPENCV_
#include <float.h>  // (opencv2/core/cvdef.h includes it: the node uses FLT_MAX without including it itself)

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#define CV_8S 1
#define CV_32S 4
#define CV_32F 5
namespace cv {
class FileStorage;
struct Scalar {
  double val[4];
  static Scalar all(double v) {
    Scalar s;
    s.val[0] = s.val[1] = s.val[2] = s.val[3] = v;
    return s;
  }
};
class Mat {
 public:
  Mat() : rows(0), cols(0), esz_(0) {}
  Mat(int r, int c, int type, const Scalar& s) : rows(r), cols(c), esz_(type == CV_8S ? 1 : 4) {
    data_.resize(static_cast<std::size_t>(r) * c * esz_);
    for (std::size_t k = 0; k < static_cast<std::size_t>(r) * c; ++k) {
      if (type == CV_32F) {
        const float v = static_cast<float>(s.val[0]);
        std::memcpy(&data_[k * 4], &v, 4);
      } else if (type == CV_32S) {
        const std::int32_t v = static_cast<std::int32_t>(s.val[0]);
        std::memcpy(&data_[k * 4], &v, 4);
      } else {
        data_[k] = static_cast<unsigned char>(static_cast<std::int8_t>(s.val[0]));
      }
    }
  }
  template <typename T>
  T& at(int i, int j) {
    return *reinterpret_cast<T*>(&data_[(static_cast<std::size_t>(i) * cols + j) * sizeof(T)]);
  }
  int rows, cols;

 private:
  std::vector<unsigned char> data_;
  std::size_t esz_;
};
}  // namespace cv
#endif
