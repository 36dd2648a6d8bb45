// <opencv2/opencv.hpp> — STAND-IN (oracle/ref_shim/README.md): parameters.h only names cv::FileStorage in two
// declarations (readV3D / readQ4D), which the _ref build never defines or calls.
#ifndef LINS_REF_SHIM_OPENCV_
#define LINS_REF_SHIM_OPENCV_
namespace cv {
class FileStorage;
}
#endif
