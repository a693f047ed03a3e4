// <pcl/point_types.h> for the reference's own mapping_types.h: the stand-in of oracle/ref_shims + the colour point type it names
#pragma once
#include "../../ref_shims/pcl/point_types.h"
namespace pcl {
struct PointXYZRGB {
    float x = 0, y = 0, z = 0;
    uint8_t r = 0, g = 0, b = 0;
};
}  // namespace pcl
