// Minimal stand-in for <pcl/point_types.h> (PCL is not installed here).  Only what the reference headers that
// oracle/ref_harness.cpp includes actually touch: the three point structs with x/y/z(/intensity/normal/curvature)
// members and getVector3fMap() / getVector4fMap() (x, y, z, 1: PCL's data[3] is 1).  Test infrastructure; not part of the product.
#pragma once
#include <Eigen/Core>
namespace pcl {
struct alignas(16) PointXYZ {
    float x = 0, y = 0, z = 0, pad_ = 1.f;
    Eigen::Map<Eigen::Vector3f> getVector3fMap() { return Eigen::Map<Eigen::Vector3f>(&x); }
    Eigen::Map<const Eigen::Vector3f> getVector3fMap() const { return Eigen::Map<const Eigen::Vector3f>(&x); }
    Eigen::Map<Eigen::Vector4f> getVector4fMap() { return Eigen::Map<Eigen::Vector4f>(&x); }
    Eigen::Map<const Eigen::Vector4f> getVector4fMap() const { return Eigen::Map<const Eigen::Vector4f>(&x); }
};
struct alignas(16) PointXYZI {
    float x = 0, y = 0, z = 0, pad_ = 1.f;
    float intensity = 0, pad2_[3] = {0, 0, 0};
    Eigen::Map<Eigen::Vector3f> getVector3fMap() { return Eigen::Map<Eigen::Vector3f>(&x); }
    Eigen::Map<const Eigen::Vector3f> getVector3fMap() const { return Eigen::Map<const Eigen::Vector3f>(&x); }
    Eigen::Map<Eigen::Vector4f> getVector4fMap() { return Eigen::Map<Eigen::Vector4f>(&x); }
    Eigen::Map<const Eigen::Vector4f> getVector4fMap() const { return Eigen::Map<const Eigen::Vector4f>(&x); }
};
struct alignas(16) PointXYZINormal {
    float x = 0, y = 0, z = 0, pad_ = 1.f;
    float normal_x = 0, normal_y = 0, normal_z = 0, pad2_ = 0;
    float intensity = 0, curvature = 0, pad3_[2] = {0, 0};
    Eigen::Map<Eigen::Vector3f> getVector3fMap() { return Eigen::Map<Eigen::Vector3f>(&x); }
    Eigen::Map<const Eigen::Vector3f> getVector3fMap() const { return Eigen::Map<const Eigen::Vector3f>(&x); }
    Eigen::Map<Eigen::Vector4f> getVector4fMap() { return Eigen::Map<Eigen::Vector4f>(&x); }
    Eigen::Map<const Eigen::Vector4f> getVector4fMap() const { return Eigen::Map<const Eigen::Vector4f>(&x); }
};
}  // namespace pcl
