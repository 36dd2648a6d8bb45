// <opencv2/core/eigen.hpp> — STAND-IN (oracle/ref_shim/README.md): unused on the compiled path.
#include <opencv2/opencv.hpp>
