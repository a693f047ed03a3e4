// stand-in for sensor_driver/common_lib/cpp_utils/SystemUtils.h (its directory also holds the spdlog Logger.h): the three
// GPS-time helpers slam_utils.cpp calls in its NMEA functions, which are not on the path.  Defined (as traps) in ref_slam_utils.cpp.
#pragma once
#include <cstdint>
#include "Logger.h"
uint64_t gps2Utc(int gps_week, double gps_time);
int getGPSweek(const uint64_t& stamp);
double getGPSsecond(const uint64_t& stamp);
