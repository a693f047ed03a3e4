// stand-in for <opencv2/opencv.hpp>: slam_utils.h / slam_utils.cpp name cv::Mat in one image helper that is not on the path
#pragma once
namespace cv {
struct Mat {
    int type() const { return -1; }
};
enum { COLOR_YUV2BGR_I420 = 101 };
inline void cvtColor(const Mat&, Mat&, int) {}
}  // namespace cv
#ifndef CV_8UC1
#define CV_8UC1 0
#endif
