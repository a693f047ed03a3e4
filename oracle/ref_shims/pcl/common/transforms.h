// stand-in for <pcl/common/transforms.h> (included by IMU_Processing.hpp, nothing from it is used on the path)
#pragma once
#include <pcl/point_cloud.h>
