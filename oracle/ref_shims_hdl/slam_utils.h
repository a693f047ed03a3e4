// stand-in for slam/common/slam_utils.h (UTM projection, image helpers): what fastlio.cpp and the Option 0 binding use
#pragma once
#include <chrono>
#include <Eigen/Geometry>
#include "mapping_types.h"
Eigen::Matrix4d getTransformFromRPYT(double x, double y, double z, double yaw, double pitch, double roll);
template <class result_t = std::chrono::milliseconds, class clock_t = std::chrono::steady_clock, class duration_t = std::chrono::milliseconds>
auto since(std::chrono::time_point<clock_t, duration_t> const& start) { return std::chrono::duration_cast<result_t>(clock_t::now() - start); }
