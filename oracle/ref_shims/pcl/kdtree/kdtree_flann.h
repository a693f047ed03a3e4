// stand-in for <pcl/kdtree/kdtree_flann.h> (included by IMU_Processing.hpp, unused)
#pragma once
