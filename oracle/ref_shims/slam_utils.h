// stand-in for slam/common/slam_utils.h (which includes UTM / system helpers that are not on the path): the one function
// laserMapping.cpp calls.  Defined in oracle/ref_fastlio.cpp.
#pragma once
#include <Eigen/Geometry>
#include "mapping_types.h"
Eigen::Matrix4d getTransformFromRPYT(double x, double y, double z, double yaw, double pitch, double roll);
