// ref_hdl_localization.cpp -- the reference's OWN localisation loop, compiled whole from where it lies and LINKED against liblio_hip.so:
//   slam/localization/hdl_localization/apps/hdl_localization_nodelet.cpp  (IMU mean, predict, undistort, downsample, match, correct, the ping-pong
//                                                                          target hand-over: frame_callback :166-275, globalmap_callback :281-297)
//   slam/localization/hdl_localization/src/pose_estimator.cpp             (the UKF around the matcher)
// with hdl_graph_slam::select_registration_method("NDT_CUDA", t) returning
//   * NdtHip -- the pcl::Registration subclass INTEGRATION.md section 3a shows a maintainer, extracted from the document at build time
//     (_ref/obj/ndt_hip.inc) -- or
//   * RefNdtAdapter -- an adapter over the reference's own fast_gicp::NDTCuda object compiled for gfx950 (oracle/_ref/libref_ndt_cuda.so, loaded at
//     run time: its translation units need hipcc), the baseline the same nodelet is driven over.
// undistortPoints / interpolateTransform are the reference's slam_utils.cpp (oracle/_ref/libref_slam_utils.so, loaded at run time).
// tests/test_localization_boundary.py drives both variants through a localisation sequence on the GPU.  Test infrastructure only.
#include <dlfcn.h>
#include <unistd.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "slam_base.h"       // ref_shims_hdl (the reference's own mapping_types.h underneath)
#include "Logger.h"
#include <pcl/search/kdtree.h>
#include <pcl/registration/registration.h>
#include "../include/lio_hip.h"

// ---- slam_utils.cpp of the reference, through the harness that compiles it whole (oracle/ref_slam_utils.cpp) ----
static void* sym(const char* so, const char* name) {
    static std::map<std::string, void*> libs;
    void*& h = libs[so];
    if (!h) {
        Dl_info info;
        dladdr(reinterpret_cast<void*>(&sym), &info);
        std::string dir = info.dli_fname;
        dir = dir.substr(0, dir.find_last_of('/') + 1);
        h = dlopen((dir + so).c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "ref_hdl_localization: cannot load %s: %s\n", so, dlerror()); abort(); }
    }
    void* f = dlsym(h, name);
    if (!f) { fprintf(stderr, "ref_hdl_localization: %s lacks %s\n", so, name); abort(); }
    return f;
}
Eigen::Matrix4d interpolateTransform(Eigen::Matrix4d& A, Eigen::Matrix4d& B, double ratio) {
    static auto f = reinterpret_cast<void (*)(const double*, const double*, double, double*)>(sym("libref_slam_utils.so", "ref_interpolate_transform"));
    double a[16], b[16], t[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { a[4 * r + c] = A(r, c); b[4 * r + c] = B(r, c); }
    f(a, b, ratio, t);
    Eigen::Matrix4d T;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T(r, c) = t[4 * r + c];
    return T;
}
void undistortPoints(const Eigen::Matrix4f& delta_pose, PointCloudAttrPtr& points, double scan_period) {
    static auto f = reinterpret_cast<void (*)(const float*, const float*, const uint32_t*, int, double, float*)>(sym("libref_slam_utils.so", "ref_undistort_delta"));
    const int n = (int)points->cloud->points.size();
    if (n == 0 || (int)points->attr.size() != n) return;
    float d[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) d[4 * r + c] = delta_pose(r, c);
    std::vector<float> in(4 * n), out(4 * n);
    std::vector<uint32_t> st(n);
    for (int i = 0; i < n; i++) {
        const Point& p = points->cloud->points[i];
        in[4 * i] = p.x; in[4 * i + 1] = p.y; in[4 * i + 2] = p.z; in[4 * i + 3] = p.intensity;
        st[i] = points->attr[i].stamp;
    }
    f(d, in.data(), st.data(), n, scan_period, out.data());
    for (int i = 0; i < n; i++) {
        Point& p = points->cloud->points[i];
        p.x = out[4 * i]; p.y = out[4 * i + 1]; p.z = out[4 * i + 2];
    }
}

// ---- the two matchers behind select_registration_method ----
#include "_ref/obj/ndt_hip.inc"

class RefNdtAdapter : public pcl::Registration<pcl::PointXYZI, pcl::PointXYZI, float> {
    using PointT = pcl::PointXYZI;
    void* h_;
    static std::vector<float> flat(const pcl::PointCloud<PointT>& c) {
        std::vector<float> b(4 * c.points.size() + 4);
        for (size_t i = 0; i < c.points.size(); i++) { b[4 * i] = c.points[i].x; b[4 * i + 1] = c.points[i].y; b[4 * i + 2] = c.points[i].z; b[4 * i + 3] = c.points[i].intensity; }
        return b;
    }

   public:
    explicit RefNdtAdapter(int64_t max_process_time) {
        static auto mk = reinterpret_cast<void* (*)(double, int, double)>(sym("libref_ndt_cuda.so", "ref_ndtreg_create"));
        h_ = mk(1.0, 7, (double)max_process_time);
        reg_name_ = "fast_gicp::NDTCuda (reference, gfx950 build)";
    }
    ~RefNdtAdapter() override {
        static auto rm = reinterpret_cast<void (*)(void*)>(sym("libref_ndt_cuda.so", "ref_ndtreg_destroy"));
        rm(h_);
    }
    void setInputTarget(const PointCloudTargetConstPtr& c) override {
        static auto f = reinterpret_cast<void (*)(void*, const float*, int)>(sym("libref_ndt_cuda.so", "ref_ndtreg_set_target"));
        pcl::Registration<PointT, PointT, float>::setInputTarget(c);
        tree_.reset();
        f(h_, flat(*c).data(), (int)c->points.size());
    }
    void setInputSource(const PointCloudSourceConstPtr& c) override {
        static auto f = reinterpret_cast<void (*)(void*, const float*, int)>(sym("libref_ndt_cuda.so", "ref_ndtreg_set_source"));
        pcl::Registration<PointT, PointT, float>::setInputSource(c);
        f(h_, flat(*c).data(), (int)c->points.size());
    }
    // pcl::Registration::getFitnessScore (PCL 1.9.1 registration.hpp): nearest target point of every transformed source point, mean squared
    // distance over those within max_range -- on the exact kd-tree stand-in
    double getFitnessScore(double max_range) override {
        if (!tree_) { tree_.reset(new pcl::search::KdTree<PointT>()); tree_->setInputCloud(target_); }
        pcl::PointCloud<PointT> moved;
        pcl::transformPointCloud(*input_, moved, final_transformation_);
        std::vector<int> idx(1);
        std::vector<float> d2(1);
        double s = 0.0;
        int nr = 0;
        for (const PointT& p : moved.points) {
            tree_->nearestKSearch(p, 1, idx, d2);
            if (!d2.empty() && d2[0] <= max_range) { s += d2[0]; nr++; }
        }
        return nr > 0 ? s / nr : std::numeric_limits<double>::max();
    }

   protected:
    void computeTransformation(PointCloudSource& out, const Eigen::Matrix4f& guess) override {
        static auto f = reinterpret_cast<int (*)(void*, const float*, float*, int*)>(sym("libref_ndt_cuda.so", "ref_ndtreg_align"));
        float g[16], T[16];
        int it = 0;
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) g[4 * r + c] = guess(r, c);
        converged_ = f(h_, g, T, &it) != 0;
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) final_transformation_(r, c) = T[4 * r + c];
        nr_iterations_ = it;
        pcl::transformPointCloud(*input_, out, final_transformation_);
    }
    std::shared_ptr<pcl::search::KdTree<PointT>> tree_;
};

static int g_use_reference_matcher = 0;
namespace hdl_graph_slam {
pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>::Ptr select_registration_method(std::string, int64_t max_process_time) {
    if (g_use_reference_matcher) return pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>::Ptr(new RefNdtAdapter(max_process_time));
    return pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>::Ptr(new NdtHip(max_process_time));
}
}  // namespace hdl_graph_slam

// ---- the reference's translation units ----
#define HAVE_CUDA_ENABLE 1  // hdl_localization_nodelet.cpp:46-50: the "NDT_CUDA" branch
#define usleep(x) ((void)0)  // the GNSS-only match sleeps 100 ms when it has nothing to fuse (pose_estimator.cpp:310)
#include <localization/hdl_localization/src/pose_estimator.cpp>
#undef usleep
#include <localization/hdl_localization/apps/hdl_localization_nodelet.cpp>

// ---- C entry points for the test ----
extern "C" {
// the matcher objects are made once per process by onInit (file-scope `registration[2]`): one variant per process
int hloc_init(int use_reference_matcher, double resolution, double scan_period, const double* imu_ext16) {
    g_use_reference_matcher = use_reference_matcher;
    InitParameter p;
    p.map_path = "";
    p.resolution = resolution;
    p.key_frame_distance = 1.0;
    p.key_frame_degree = 10.0;
    p.key_frame_range = 100.0;
    p.scan_period = scan_period;
    init_hdl_localization_node(p);
    Eigen::Matrix4d E;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) E(r, c) = imu_ext16[4 * r + c];
    set_imu_extrinic_hdl_localization(E);
    return 0;
}
void hloc_deinit() { deinit_hdl_localization_node(); }
void hloc_set_initpose(uint64_t stamp_us, const double* T16) {
    Eigen::Matrix4d T;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T(r, c) = T16[4 * r + c];
    set_initpose_hdl_localization(stamp_us, T);
}
void hloc_set_map(const float* xyzi, int n) {  // Localization::updateLocalMap -> globalmap_callback: the next local map (NULL / 0: none)
    PointCloud::Ptr c;
    if (xyzi && n > 0) {
        c.reset(new PointCloud());
        c->points.resize(n);
        for (int i = 0; i < n; i++) { c->points[i].x = xyzi[4 * i]; c->points[i].y = xyzi[4 * i + 1]; c->points[i].z = xyzi[4 * i + 2]; c->points[i].intensity = xyzi[4 * i + 3]; }
        c->width = n; c->height = 1;
    }
    set_map_hdl_localization(c);
}
void hloc_imu(double stamp_s, const double* acc, const double* gyr) {
    ImuType imu;
    imu.stamp = stamp_s;
    imu.acc = Eigen::Vector3d(acc[0], acc[1], acc[2]);
    imu.gyr = Eigen::Vector3d(gyr[0], gyr[1], gyr[2]);
    enqueue_imu_hdl_localization(imu);
}
void hloc_ins(uint64_t stamp_us, const double* T16, double precision, int dimension) {
    std::shared_ptr<RTKType> ins(new RTKType());
    ins->timestamp = stamp_us;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) ins->T(r, c) = T16[4 * r + c];
    ins->precision = precision;
    ins->dimension = dimension;
    enqueue_ins_hdl_localization(ins);
}
// one frame through frame_callback: returns LocType (0 OK, 1 ERROR, 2 OTHER), the pose (row-major 4 x 4)
int hloc_frame(const float* xyzi, const uint32_t* stamp_us, int n, uint64_t header_stamp_us, double* pose16) {
    PointCloudAttrPtr c(new PointCloudAttr());
    c->cloud->points.resize(n);
    c->attr.resize(n);
    for (int i = 0; i < n; i++) {
        c->cloud->points[i].x = xyzi[4 * i]; c->cloud->points[i].y = xyzi[4 * i + 1]; c->cloud->points[i].z = xyzi[4 * i + 2];
        c->cloud->points[i].intensity = xyzi[4 * i + 3];
        c->attr[i].stamp = stamp_us[i];
        c->attr[i].id = 0;
    }
    c->cloud->width = n; c->cloud->height = 1;
    c->cloud->header.stamp = header_stamp_us;
    ImageType image;
    Eigen::Isometry3d pose = Eigen::Isometry3d::Identity();
    const LocType r = enqueue_hdl_localization(c, image, pose);
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) pose16[4 * a + b] = pose.matrix()(a, b);
    return (int)r;
}
int hloc_timed_pose(uint64_t stamp_us, double* pose16) {
    Eigen::Matrix4d T = Eigen::Matrix4d::Identity();
    const bool ok = get_timed_pose_hdl_localization(stamp_us, T);
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) pose16[4 * a + b] = T(a, b);
    return ok ? 1 : 0;
}
}
