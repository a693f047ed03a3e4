// stand-in for the reference's slam/common/mapping_types.h, which pulls in OpenCV, g2o and a lock-free queue that are not
// installed.  Only the few plain data types the FastLIO sources name are declared, with the members those sources touch
// (mapping_types.h:20-38, 86-148): same member names, types and defaults.
#pragma once
#include <Eigen/Geometry>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include <map>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <opencv2/opencv.hpp>

typedef pcl::PointXYZI Point;
typedef pcl::PointCloud<Point> PointCloud;

struct PointAttr {
    int id;
    uint32_t stamp;  // us, relative to the scan's header stamp
};

struct PointCloudAttr {
    PointCloudAttr() { cloud = PointCloud::Ptr(new PointCloud()); }
    PointCloud::Ptr cloud;
    std::vector<PointAttr> attr;
    Eigen::Matrix4d T;
};
typedef std::shared_ptr<PointCloudAttr> PointCloudAttrPtr;

struct RTKType {
    uint64_t timestamp = 0;
    double heading = 0, pitch = 0, roll = 0;
    double gyro_x = 0, gyro_y = 0, gyro_z = 0, acc_x = 0, acc_y = 0, acc_z = 0;
    double latitude = 0, longitude = 0, altitude = 0;
    double Ve = 0, Vn = 0, Vu = 0;
    int status = 0;
    std::string sensor, state;
    int dimension = 2;
    double precision = 100.0;
    Eigen::Matrix4d T = Eigen::Matrix4d::Identity();
    Eigen::VectorXf mean;
};

struct PoseType {
    double latitude = 0, longitude = 0, altitude = 0, heading = 0, pitch = 0, roll = 0;
    int status = 0;
    std::string state;
    uint64_t timestamp = 0;
    Eigen::Matrix4d T = Eigen::Matrix4d::Identity();
};

struct ImuType {
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    double stamp = 0;
    Eigen::Vector3d acc = Eigen::Vector3d::Zero();
    Eigen::Vector3d gyr = Eigen::Vector3d::Zero();
    Eigen::Quaterniond rot = Eigen::Quaterniond::Identity();
};
