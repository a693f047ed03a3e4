// ref_hdl_fastlio.cpp -- the reference's OWN class Mapping::HDL_FastLIO (/root/reference/slam/mapping/fastlio/src/fastlio.cpp, compiled
// whole from where it lies) LINKED against liblio_hip.so through the Option 0 binding that INTEGRATION.md shows a maintainer (the text is
// extracted from the document at build time: _ref/obj/option0.inc), and driven through its public interface -- setSensors / setStaticTransform /
// init / feedImuData / feedPointData / getPose, with its runLio thread polling fastlio_main() -- on the GPU box.
// tests/test_hdl_fastlio_gpu.py holds its odometry against libref_fastlio.so (the reference's laserMapping.cpp on the CPU) fed the same data.
// The graph back end / floor detection / prefilter nodelets HDL_FastLIO starts are out of scope: defined here as no-ops.
// Test infrastructure only.
#include <sys/prctl.h>
#include <unistd.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "fastlio.h"  // the reference's (slam/mapping/fastlio/include), over its own slam_base.h / mapping_types.h
#include "Logger.h"

// ---- the back end the class starts: no-ops ----
void init_floor_node() {}
void deinit_floor_node() {}
FloorCoeffs enqueue_floor(PointCloud::Ptr&) { return FloorCoeffs(); }
void init_filter_node(InitParameter&) {}
void deinit_filter_node() {}
PointCloud::Ptr enqueue_filter(PointCloud::Ptr& p) { return p; }
void init_graph_node(InitParameter&) {}
void deinit_graph_node() {}
void graph_set_origin(RTKType&) {}
void enqueue_graph_floor(FloorCoeffs&) {}
void enqueue_graph_gps(bool, std::shared_ptr<RTKType>&) {}
Eigen::Isometry3d get_odom2map() { return Eigen::Isometry3d::Identity(); }
void graph_optimization(bool&) {}

Eigen::Matrix4d getTransformFromRPYT(double x, double y, double z, double yaw, double pitch, double roll) {  // slam_utils.cpp:89-96
    const double Ang2Rad = 0.01745329251994;
    Eigen::AngleAxisd rollAngle(roll * Ang2Rad, Eigen::Vector3d::UnitY());
    Eigen::AngleAxisd pitchAngle(pitch * Ang2Rad, Eigen::Vector3d::UnitX());
    Eigen::AngleAxisd yawAngle(yaw * Ang2Rad, Eigen::Vector3d::UnitZ());
    Eigen::Translation3d trans(x, y, z);
    return (trans * yawAngle * pitchAngle * rollAngle).matrix();
}

// ---- the reference's translation unit, then the binding of INTEGRATION.md in place of laserMapping.cpp ----
#include "src/fastlio.cpp"
static Eigen::Matrix3d Lidar_R_wrt_IMU = Eigen::Matrix3d::Identity();  // laserMapping.cpp:139 (read by the binding's fastlio_ins_enqueue)
#include "_ref/obj/option0.inc"

// ---- C entry points for the test ----
static std::unique_ptr<Mapping::HDL_FastLIO> H;
extern "C" {
int hdl_create(const char* lidar_name, const double* T_static16, const double* T_imu16, double scan_period) {
    H.reset(new Mapping::HDL_FastLIO());
    std::vector<std::string> sensors = {std::string(lidar_name), "IMU"};
    std::vector<std::string> used = H->setSensors(sensors);
    Eigen::Matrix4d Ts = Eigen::Map<const Eigen::Matrix<double, 4, 4, Eigen::RowMajor>>(T_static16);
    Eigen::Matrix4d Ti = Eigen::Map<const Eigen::Matrix<double, 4, 4, Eigen::RowMajor>>(T_imu16);
    H->setStaticTransform(Ts);
    H->setImuStaticTransform(Ti);
    InitParameter p;
    p.map_path = "";
    p.resolution = 0.2;
    p.key_frame_distance = 1.0;
    p.key_frame_degree = 10.0;
    p.key_frame_range = 100.0;
    p.scan_period = scan_period;
    return H->init(p) ? (int)used.size() : -1;
}
void hdl_destroy() { H.reset(nullptr); }
void hdl_feed_imu(double stamp, const double* gyr, const double* acc_ms2) {
    ImuType imu;
    imu.stamp = stamp;
    imu.gyr = Eigen::Vector3d(gyr[0], gyr[1], gyr[2]);
    imu.acc = Eigen::Vector3d(acc_ms2[0], acc_ms2[1], acc_ms2[2]);
    H->feedImuData(imu);
}
// one frame: feedPointData (lidar-frame cloud, the class applies the static transform) then getPose (blocks on the LIO thread's odometry)
int hdl_frame(const char* lidar_name, const float* xyzi, const uint32_t* stamp_us, int n, uint64_t header_stamp_us, double* pose16, double* delta16) {
    PointCloudAttrPtr c(new PointCloudAttr());
    c->cloud->points.resize(n);
    c->attr.resize(n);
    for (int i = 0; i < n; i++) {
        c->cloud->points[i].x = xyzi[4 * i]; c->cloud->points[i].y = xyzi[4 * i + 1]; c->cloud->points[i].z = xyzi[4 * i + 2];
        c->cloud->points[i].intensity = xyzi[4 * i + 3];
        c->attr[i].stamp = stamp_us[i];
        c->attr[i].id = 0;
    }
    c->cloud->width = n;
    c->cloud->height = 1;
    c->cloud->header.stamp = header_stamp_us;
    std::map<std::string, PointCloudAttrPtr> pts;
    pts[lidar_name] = c;
    H->feedPointData(header_stamp_us, pts);
    PointCloudAttrImagePose frame;
    const Eigen::Matrix4d T = H->getPose(frame);
    for (int r = 0; r < 4; r++)
        for (int col = 0; col < 4; col++) { pose16[r * 4 + col] = T(r, col); delta16[r * 4 + col] = frame.points->T(r, col); }
    return (int)frame.imu_poses.size();
}
int hdl_is_init() { return fastlio_is_init() ? 1 : 0; }
}
