// <opencv2/opencv.hpp> — STAND-IN (oracle/ref_shim/README.md).  parameters.h names cv::FileStorage in two declarations
// (readV3D / readQ4D), which the _ref build never defines or calls; image_projection_node.cpp keeps its range / label /
// ground images in cv::Mat and uses exactly: the (rows, cols, type, Scalar) constructor with CV_32F / CV_8S / CV_32S,
// Scalar::all, assignment, and at<T>(row, col); lidar_mapping_node.cpp's scan-to-map optimisation in addition: cv::eigen,
// cv::solve(DECOMP_QR), cv::transpose, Mat * Mat, Mat::inv, Mat::copyTo — their numerics in lins_ref_shim/cv_restated.h.
#ifndef LINS_REF_SHIM_OPENCV_
#define LINS_REF_SHIM_OPENCV_
#include <float.h>  // (opencv2/core/cvdef.h includes it: the node uses FLT_MAX without including it itself)

#include <lins_ref_shim/cv_restated.h>

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
enum { DECOMP_LU = 0, DECOMP_QR = 4 };
class Mat {
 public:
  Mat() : rows(0), cols(0), type_(CV_32F) {}
  Mat(int r, int c, int type, const Scalar& s) : rows(r), cols(c), type_(type) {
    const std::size_t esz = type == CV_8S ? 1 : 4;
    data_.resize(static_cast<std::size_t>(r) * c * esz);
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
  template <typename T>
  const T& at(int i, int j) const {
    return *reinterpret_cast<const T*>(&data_[(static_cast<std::size_t>(i) * cols + j) * sizeof(T)]);
  }
  float* f32() { return reinterpret_cast<float*>(data_.data()); }
  const float* f32() const { return reinterpret_cast<const float*>(data_.data()); }
  void copyTo(Mat& dst) const { dst = *this; }
  Mat inv(int = DECOMP_LU) const {  // (square CV_32F; lins_cvr::inv)
    Mat r(rows, cols, CV_32F, Scalar::all(0));
    lins_cvr::inv(f32(), rows, r.f32());
    return r;
  }
  int rows, cols;

 private:
  std::vector<unsigned char> data_;
  int type_;
};
inline Mat operator*(const Mat& a, const Mat& b) {  // CV_32F; lins_cvr::matmul (f64 accumulation)
  Mat r(a.rows, b.cols, CV_32F, Scalar::all(0));
  lins_cvr::matmul(a.f32(), a.rows, a.cols, b.f32(), b.cols, r.f32());
  return r;
}
inline void transpose(const Mat& src, Mat& dst) {
  Mat r(src.cols, src.rows, CV_32F, Scalar::all(0));
  for (int i = 0; i < src.rows; ++i)
    for (int j = 0; j < src.cols; ++j) r.at<float>(j, i) = src.at<float>(i, j);
  dst = r;
}
// eigenvalues (1 x n or n x 1, descending) and eigenvectors (rows) of a symmetric CV_32F matrix; src is left untouched
inline bool eigen(const Mat& src, Mat& eigenvalues, Mat& eigenvectors) {
  const int n = src.rows;
  std::vector<float> a(src.f32(), src.f32() + static_cast<std::size_t>(n) * n), w(n), v(static_cast<std::size_t>(n) * n);
  lins_cvr::jacobi_eig(a.data(), n, w.data(), v.data());
  if (eigenvalues.rows * eigenvalues.cols != n) eigenvalues = Mat(n, 1, CV_32F, Scalar::all(0));
  if (eigenvectors.rows != n || eigenvectors.cols != n) eigenvectors = Mat(n, n, CV_32F, Scalar::all(0));
  std::memcpy(eigenvalues.f32(), w.data(), sizeof(float) * n);
  std::memcpy(eigenvectors.f32(), v.data(), sizeof(float) * n * n);
  return true;
}
// least squares / linear solve, DECOMP_QR, one right-hand side; the operands are left untouched
inline bool solve(const Mat& A, const Mat& B, Mat& X, int /*flags: DECOMP_QR on this path*/) {
  std::vector<float> a(A.f32(), A.f32() + static_cast<std::size_t>(A.rows) * A.cols), b(B.f32(), B.f32() + B.rows), x(A.cols);
  lins_cvr::qr_solve(a.data(), A.rows, A.cols, b.data(), x.data());
  if (X.rows != A.cols || X.cols != 1) X = Mat(A.cols, 1, CV_32F, Scalar::all(0));
  std::memcpy(X.f32(), x.data(), sizeof(float) * A.cols);
  return true;
}
}  // namespace cv
#endif
