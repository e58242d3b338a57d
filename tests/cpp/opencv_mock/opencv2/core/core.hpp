// Declaration-only stand-in for the handful of cv::Mat members include/rslf_hip.hpp touches, so that the
// RSLFX_HAVE_OPENCV block of OUR wrapper can be compiled (syntax and types) in an image without OpenCV
// (tests/test_abi.py::test_opencv_block_compiles).  It defines no behaviour and is never linked or run;
// it is not, and must not be used as, a way to build the reference.
#pragma once
#include <cstddef>

#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC(n) CV_MAKETYPE(CV_32F, (n))

namespace cv {
struct MatStep {
    size_t operator[](int i) const;
};
class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type, void* data);
    Mat clone() const;
    int channels() const;
    int type() const;
    int depth() const;
    int rows, cols;
    unsigned char* data;
    MatStep step;
};
}  // namespace cv
