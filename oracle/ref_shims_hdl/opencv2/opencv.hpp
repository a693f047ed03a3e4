// stand-in for <opencv2/opencv.hpp>: the reference's mapping_types.h constructs a few cv::Mat members (camera parameters) that the
// FastLIO path never reads
#pragma once
#define CV_32F 5
#ifndef CV_8UC1
#define CV_8UC1 0
#endif
namespace cv {
template <typename T> struct DataType { static const int type = 5; };
struct Mat {
    Mat() {}
    Mat(int, int, int) {}
    int type() const { return -1; }
    template <typename T> T& at(int) { static T v; return v; }
};
enum { COLOR_YUV2BGR_I420 = 101 };
inline void cvtColor(const Mat&, Mat&, int) {}
}  // namespace cv
