// stand-in for <pcl/common/centroid.h> (ivox3d_node.hpp includes it but uses nothing from it)
#pragma once
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
