// oracle/ref_slam_utils.cpp -- the reference's OWN slam/common/slam_utils.cpp compiled whole from where it lies (nothing is copied),
// behind a C ABI: undistortPoints (the constant-velocity motion compensation of the localisation mode, :163-191, and the pose-list
// variant :193-228), getTransformFromRPYT, interpolateTransform, pointsDistanceFilter.  Shimmed: PCL containers, pcl::transformPoint
// (PCL 1.9.1's published one-liner), logging, cv::Mat and the plain structs of mapping_types.h; the UTM / GPS-time helpers its NMEA
// functions link against are traps.  TEST INFRASTRUCTURE ONLY: oracle/_ref/libref_slam_utils.so by `make -C oracle ref`.
#define __MAPPING_TYPES_H  // slam_utils.h sits beside the real mapping_types.h (OpenCV, g2o, a lock-free queue): keep it out, use the shim
#include "ref_shims/mapping_types.h"
#include <common/slam_utils.cpp>  // -I$(REF)/slam

#include <cstdlib>

uint64_t gps2Utc(int, double) { abort(); }
int getGPSweek(const uint64_t&) { abort(); }
double getGPSsecond(const uint64_t&) { abort(); }
UTMProjector::UTMProjector(int) { abort(); }
UTMProjector::~UTMProjector() {}
int UTMProjector::FromGlobalToLocal(double, double, double&, double&) { abort(); }
int UTMProjector::FromLocalToGlobal(double, double, double&, double&) { abort(); }
double UTMProjector::GetLongitude0() { abort(); }
double get_grid_convergence(const double&, const double&, const double&) { abort(); }

static PointCloudAttrPtr make_cloud(const float* xyzi, const uint32_t* stamp_us, int n, uint64_t header_us) {
    PointCloudAttrPtr f(new PointCloudAttr());
    f->cloud->points.resize(n);
    f->attr.resize(n);
    for (int i = 0; i < n; i++) {
        Point& p = f->cloud->points[i];
        p.x = xyzi[4 * i]; p.y = xyzi[4 * i + 1]; p.z = xyzi[4 * i + 2]; p.intensity = xyzi[4 * i + 3];
        f->attr[i].id = 0;
        f->attr[i].stamp = stamp_us[i];
    }
    f->cloud->header.stamp = header_us;
    return f;
}
static void cloud_back(const PointCloudAttrPtr& f, float* out) {
    for (size_t i = 0; i < f->cloud->points.size(); i++) {
        const Point& p = f->cloud->points[i];
        out[4 * i] = p.x; out[4 * i + 1] = p.y; out[4 * i + 2] = p.z; out[4 * i + 3] = p.intensity;
    }
}

extern "C" {
// undistortPoints(const Eigen::Matrix4f& delta_pose, PointCloudAttrPtr&, double scan_period); delta row-major
void ref_undistort_delta(const float* delta16, const float* xyzi, const uint32_t* stamp_us, int n, double scan_period, float* out_xyzi) {
    Eigen::Matrix4f D;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) D(r, c) = delta16[4 * r + c];
    PointCloudAttrPtr f = make_cloud(xyzi, stamp_us, n, 0);
    undistortPoints(D, f, scan_period);
    cloud_back(f, out_xyzi);
}
// undistortPoints(std::vector<PoseType>& poses, PointCloudAttrPtr&): poses[i] = (timestamp us, T row-major), i = 0 is the scan start
void ref_undistort_poses(const uint64_t* pose_stamp_us, const double* pose_T16, int n_poses, const float* xyzi, const uint32_t* stamp_us, int n,
                         uint64_t header_us, float* out_xyzi) {
    std::vector<PoseType> poses(n_poses);
    for (int k = 0; k < n_poses; k++) {
        poses[k].timestamp = pose_stamp_us[k];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) poses[k].T(r, c) = pose_T16[16 * k + 4 * r + c];
    }
    PointCloudAttrPtr f = make_cloud(xyzi, stamp_us, n, header_us);
    undistortPoints(poses, f);
    cloud_back(f, out_xyzi);
}
void ref_transform_from_rpyt(double x, double y, double z, double yaw, double pitch, double roll, double* T16) {
    Eigen::Matrix4d T = getTransformFromRPYT(x, y, z, yaw, pitch, roll);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T16[4 * r + c] = T(r, c);
}
void ref_interpolate_transform(const double* A16, const double* B16, double ratio, double* T16) {
    Eigen::Matrix4d A, B;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { A(r, c) = A16[4 * r + c]; B(r, c) = B16[4 * r + c]; }
    Eigen::Matrix4d T = interpolateTransform(A, B, ratio);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T16[4 * r + c] = T(r, c);
}
int ref_distance_filter(const float* xyzi, int n, double min_range, double max_range, float* out_xyzi) {
    PointCloud::Ptr in(new PointCloud()), out(new PointCloud());
    in->points.resize(n);
    for (int i = 0; i < n; i++) { Point& p = in->points[i]; p.x = xyzi[4 * i]; p.y = xyzi[4 * i + 1]; p.z = xyzi[4 * i + 2]; p.intensity = xyzi[4 * i + 3]; }
    PointCloud::ConstPtr cin = in;
    pointsDistanceFilter(cin, out, min_range, max_range);
    for (size_t i = 0; i < out->points.size(); i++) { const Point& p = out->points[i]; out_xyzi[4 * i] = p.x; out_xyzi[4 * i + 1] = p.y; out_xyzi[4 * i + 2] = p.z; out_xyzi[4 * i + 3] = p.intensity; }
    return (int)out->points.size();
}
}
