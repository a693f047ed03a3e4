// stand-in for <pcl/io/pcd_io.h>: the map-saving helper of laserMapping.cpp names PCDWriter; it is never called here
#pragma once
#include <string>
#include <pcl/point_cloud.h>
namespace pcl {
struct PCDWriter {
    template <typename C> int writeBinary(const std::string&, const C&) { return 0; }
};
}  // namespace pcl
