// slam_wrapper.cpp -- the reference's OUTER boundary: the pybind11 module `slam_wrapper` that slam/slam.py and slam/map_manager.py
// import (/root/reference/slam/src/slam_wrapper.cpp:192-324: same function names, argument names and order, return types), in C++
// over the C ABI of liblio_hip.so.
//
// On the hot path (mapping mode, FastLIO): init_slam / set_ins_external_param / set_imu_external_param / setup_slam / process /
// deinit_slam do what the reference does between Python and the filter --
//   numpy_to_imu            slam/src/py_utils.cpp:244-258   deg/s -> rad/s, g -> m/s^2, us -> s
//   pydict_to_cloud         slam/src/py_utils.cpp:149-181   N x 4 f32 + N x 2 attr (col 0 = per-point offset in us)
//   preprocessPoints        slam/common/slam_base.h:83-85   lidar -> INS static transform (pcl::transformPointCloud with a Matrix4d)
//   HDL_FastLIO::init / setSensors / feedImuData / feedPointData / runLio / getPose   slam/mapping/fastlio/src/fastlio.cpp:119-277
//   SLAM::run               slam/src/slam.cpp:273-367       pose fields, heading / pitch / roll from the odometry
// -- except that the cloud goes numpy -> pinned staging buffer -> HBM in ONE pass (lio_fastlio_pcl_stage / _commit): the static
// transform and the stamp conversion are applied while copying, the 48-byte PointXYZINormal inflation and the two intermediate
// PCL clouds of the reference never exist.
// Off the hot path (graph back end, GNSS, map export, colouration: SURVEY.md section 2 OUT-OF-SCOPE, Appendix B): type-correct minimal
// implementations -- empty dict / list, identity 4 x 4, stored-and-returned settings -- so that slam.py and map_manager.py run
// unchanged.  They are listed in INTEGRATION.md.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <iomanip>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lio_hip.h"

namespace py = pybind11;

namespace {

constexpr double kAng2Rad = 0.01745329251994;  // the reference's truncated constant (slam/common/slam_utils.cpp:88)

struct Mat4 {
    double m[16];
    static Mat4 identity() {
        Mat4 r;
        for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
        return r;
    }
    double& operator()(int r, int c) { return m[r * 4 + c]; }
    double operator()(int r, int c) const { return m[r * 4 + c]; }
};
Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += a(i, k) * b(k, j);
            r(i, j) = s;
        }
    return r;
}
Mat4 rigid_inverse(const Mat4& a) {  // [R t; 0 1]^-1 = [R^T  -R^T t; 0 1]
    Mat4 r = Mat4::identity();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r(i, j) = a(j, i);
    for (int i = 0; i < 3; i++) r(i, 3) = -(r(i, 0) * a(0, 3) + r(i, 1) * a(1, 3) + r(i, 2) * a(2, 3));
    return r;
}
// getTransformFromRPYT (slam_utils.cpp:89-96): translation * Rz(yaw) * Rx(pitch) * Ry(roll), angles in degrees
Mat4 transform_from_rpyt(double x, double y, double z, double yaw, double pitch, double roll) {
    const double cy = std::cos(yaw * kAng2Rad), sy = std::sin(yaw * kAng2Rad);
    const double cp = std::cos(pitch * kAng2Rad), sp = std::sin(pitch * kAng2Rad);
    const double cr = std::cos(roll * kAng2Rad), sr = std::sin(roll * kAng2Rad);
    Mat4 Rz = Mat4::identity(), Rx = Mat4::identity(), Ry = Mat4::identity(), T = Mat4::identity();
    Rz(0, 0) = cy; Rz(0, 1) = -sy; Rz(1, 0) = sy; Rz(1, 1) = cy;
    Rx(1, 1) = cp; Rx(1, 2) = -sp; Rx(2, 1) = sp; Rx(2, 2) = cp;
    Ry(0, 0) = cr; Ry(0, 2) = sr; Ry(2, 0) = -sr; Ry(2, 2) = cr;
    T(0, 3) = x; T(1, 3) = y; T(2, 3) = z;
    return mul(mul(mul(T, Rz), Rx), Ry);
}
// getRPYTfromTransformFrom (slam_utils.cpp:98-110): Eigen's MatrixBase::eulerAngles(2, 0, 1) of the rotation block (Geometry/EulerAngles.h,
// the Tait-Bryan branch with a0 = 2, a1 = 0, a2 = 1: even permutation, i = 2, j = 0, k = 1, result negated), in degrees
void rpy_from_transform(const Mat4& T, double& yaw, double& pitch, double& roll) {
    const int i = 2, j = 0, k = 1;
    double r0 = std::atan2(T(j, k), T(k, k));
    const double c2 = std::sqrt(T(i, i) * T(i, i) + T(i, j) * T(i, j));
    double r1;
    if (r0 > 0.0) {  // (!odd && res[0] > 0)
        r0 -= M_PI;
        r1 = std::atan2(-T(i, k), -c2);
    } else {
        r1 = std::atan2(-T(i, k), c2);
    }
    const double s1 = std::sin(r0), c1 = std::cos(r0);
    const double r2 = std::atan2(s1 * T(k, i) - c1 * T(j, i), c1 * T(j, j) - s1 * T(k, j));
    yaw = -r0 / kAng2Rad;
    pitch = -r1 / kAng2Rad;
    roll = -r2 / kAng2Rad;
}

// ---- localisation mode (slam/localization): the state of Locate::Localization + the hdl_localization nodelet, over the C ABI ----
struct KeyFrameDisk {  // one directory of <map>/graph: `data` (KeyFrame::save, slam/common/keyframe.cpp:118-133) + `cloud.pcd`
    uint64_t stamp = 0;
    int id = 0;
    Mat4 odom = Mat4::identity();
    std::vector<float> xyzi;   // in the map frame (KeyFrame::mTransfromPoints)
    std::vector<float> local;  // as stored (KeyFrame::mPoints): what get_graph_map hands to map_manager.py
};
struct Loc {
    std::vector<KeyFrameDisk> frames;
    lio_localmap* lm = nullptr;
    lio_ndt* ndt = nullptr;
    lio_scan* scan = nullptr;
    lio_pose_estimator* pe = nullptr;
    lio_ndt_params par;
    double resolution = 0.2, key_frame_distance = 1.0;
    bool have_init_pose = false, initialized = false, have_map = false;
    Mat4 init_pose = Mat4::identity(), last_odom = Mat4::identity();
    uint64_t init_stamp = 0;
    int age = 0, failures = 0;          // mLocalizationAge / mFailureCounter (localization.cpp:235-275)
    std::deque<std::pair<uint64_t, Mat4>> pose_data;  // HdlLocalizationNodelet::pose_data (<= 10)
    struct Imu { double stamp; float acc[3], gyr[3]; };
    std::vector<Imu> imu_data;
    std::vector<uint32_t> stamps;
    std::vector<float> staged;
    ~Loc() {
        if (pe) lio_pose_estimator_destroy(pe);
        if (lm) lio_localmap_destroy(lm);
        if (ndt) lio_ndt_destroy(ndt);
        if (scan) lio_scan_destroy(scan);
    }
};

struct Slam {
    std::unique_ptr<Loc> loc;
    std::string map_path;
    std::string mode, method;
    std::vector<std::string> sensors;
    std::string lidar;
    bool use_imu = false, use_gps = false;
    Mat4 T_static = Mat4::identity();      // lidar -> INS   (set_ins_external_param)
    Mat4 T_imu = Mat4::identity();         // IMU extrinsic  (set_imu_external_param)
    Mat4 T_imu_ins = Mat4::identity(), T_imu_ins_inv = Mat4::identity();
    double scan_period = 0.1;
    lio_engine* engine = nullptr;
    // HDL_FastLIO::runLio + mOdomQueue
    std::thread lio_thread;
    std::atomic<bool> running{false};
    std::mutex mtx;
    std::condition_variable cv;
    std::deque<std::pair<Mat4, Mat4>> odom_queue;
    // settings the off-path calls store and return
    double origin[6] = {0, 0, 0, 0, 0, 0};
    bool origin_set = false;
    bool ground_constraint = false, loop_closure = false, gravity_constraint = false, colouration = false;
    // SLAM::setInsConfig / preprocessInsData (slam.cpp:196-268): the GNSS status configurations in priority order, the state of the status filter
    struct InsCfg { std::string name; int status = 0, priority = -1; double stable_time = 0, precision = 0; };
    std::vector<InsCfg> ins_cfg;
    int last_ins_priority = -1;
    double last_ins_timestamp = 0;
    double init_pose[6] = {0, 0, 0, 0, 0, 0};
    std::string dest;
    int dest_port = 0;
    double export_z[2] = {0, 0};
    std::string export_color;
    uint64_t max_points = 8000000, max_voxels = 1u << 21;
};
std::unique_ptr<Slam> g;  // one global instance per process, like the reference's slam_ptr (slam_wrapper.cpp:4)

void require(bool ok, const char* what) {
    if (!ok) throw std::runtime_error(std::string("slam_wrapper: ") + what + (lio_last_error()[0] ? std::string(": ") + lio_last_error() : std::string()));
}


// ---- the map on disk: <map>/graph/<n>/{data, cloud.pcd} as KeyFrame::save writes them (slam/common/keyframe.cpp:118-133; the cloud through
// pcl::io::savePCDFileBinary for PointXYZI: FIELDS x y z intensity, 16 bytes per point) -------------------------------------------------
bool write_pcd_binary(const std::string& path, const float* xyzi, size_t n) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH " << n
      << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
    f.write(reinterpret_cast<const char*>(xyzi), (std::streamsize)(n * 16));
    return (bool)f;
}
// reads the x, y, z, intensity fields of an ascii or binary PCD (any field order / extra fields of 4-byte scalars)
bool read_pcd(const std::string& path, std::vector<float>& xyzi) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<std::string> fields;
    std::vector<int> sizes, counts;
    size_t points = 0;
    std::string data_kind, line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "FIELDS") { std::string v; while (ls >> v) fields.push_back(v); }
        else if (key == "SIZE") { int v; while (ls >> v) sizes.push_back(v); }
        else if (key == "COUNT") { int v; while (ls >> v) counts.push_back(v); }
        else if (key == "POINTS") ls >> points;
        else if (key == "DATA") { ls >> data_kind; break; }
    }
    if (fields.empty() || sizes.size() != fields.size()) return false;
    if (counts.empty()) counts.assign(fields.size(), 1);
    int off[4] = {-1, -1, -1, -1}, stride = 0, col[4] = {-1, -1, -1, -1}, ncol = 0;
    for (size_t i = 0; i < fields.size(); i++) {
        const int k = fields[i] == "x" ? 0 : fields[i] == "y" ? 1 : fields[i] == "z" ? 2 : fields[i] == "intensity" ? 3 : -1;
        if (k >= 0) { if (sizes[i] != 4) return false; off[k] = stride; col[k] = ncol; }
        stride += sizes[i] * counts[i];
        ncol += counts[i];
    }
    if (off[0] < 0 || off[1] < 0 || off[2] < 0) return false;
    xyzi.assign(points * 4, 0.f);
    if (data_kind == "binary") {
        std::vector<char> rec(stride);
        for (size_t i = 0; i < points; i++) {
            if (!f.read(rec.data(), stride)) return false;
            for (int k = 0; k < 4; k++)
                if (off[k] >= 0) std::memcpy(&xyzi[4 * i + k], rec.data() + off[k], 4);
        }
    } else if (data_kind == "ascii") {
        std::vector<double> row(ncol);
        for (size_t i = 0; i < points; i++) {
            for (int c = 0; c < ncol; c++) if (!(f >> row[c])) return false;
            for (int k = 0; k < 4; k++) if (col[k] >= 0) xyzi[4 * i + k] = (float)row[col[k]];
        }
    } else {
        return false;  // (binary_compressed: not written by the reference)
    }
    return true;
}
bool write_keyframe_data(const std::string& dir, uint64_t stamp_us, int id, const Mat4& odom) {  // KeyFrame::save (keyframe.cpp:122-132)
    std::ofstream ofs(dir + "/data");
    if (!ofs) return false;
    auto mat = [&](std::ostream& o) {
        for (int r = 0; r < 4; r++) { for (int c = 0; c < 4; c++) o << (c ? " " : "") << odom(r, c); o << "\n"; }  // (Eigen's operator<<: 6 significant digits)
    };
    ofs << "stamp " << stamp_us / 1000000ULL << " " << stamp_us % 1000000ULL * 1000 << std::endl;
    ofs << "estimate" << std::endl; mat(ofs);
    ofs << "odom " << std::endl; mat(ofs);
    ofs << "id " << id << std::endl;
    return (bool)ofs;
}
bool read_keyframe_data(const std::string& dir, KeyFrameDisk& kf) {  // KeyFrame::loadOdom, graph form (keyframe.cpp:41-64)
    std::ifstream ifs(dir + "/data");
    if (!ifs) return false;
    bool have = false;
    while (!ifs.eof()) {
        std::string token;
        ifs >> token;
        if (token == "stamp") { uint64_t sec = 0, nsec = 0; ifs >> sec >> nsec; kf.stamp = sec * 1000000ULL + nsec / 1000ULL; }
        else if (token == "estimate") { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) ifs >> kf.odom(i, j); have = true; }
        else if (token == "id") ifs >> kf.id;
    }
    return have;
}
// MapLoader::getKeyframeFiles + loadGraphMap (map_loader.cpp:255-300): every sub-directory of <map>/graph holding `data` and `cloud.pcd`, sorted by
// name, then by id; clouds moved into the map frame by their pose (KeyFrame::transformPoints: pcl::transformPointCloud with a Matrix4d)
bool load_keyframes(const std::string& map_path, std::vector<KeyFrameDisk>& out) {
    const std::string graph = map_path + "/graph";
    DIR* d = opendir(graph.c_str());
    if (!d) return false;
    std::vector<std::string> dirs;
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        const std::string p = graph + "/" + name;
        struct stat st;
        if (stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && stat((p + "/data").c_str(), &st) == 0 && stat((p + "/cloud.pcd").c_str(), &st) == 0) dirs.push_back(p);
    }
    closedir(d);
    std::sort(dirs.begin(), dirs.end());
    for (const std::string& p : dirs) {
        KeyFrameDisk kf;
        if (!read_keyframe_data(p, kf) || !read_pcd(p + "/cloud.pcd", kf.xyzi)) { out.clear(); return false; }
        kf.local = kf.xyzi;
        const size_t n = kf.xyzi.size() / 4;
        for (size_t i = 0; i < n; i++) {
            const double x = kf.xyzi[4 * i], y = kf.xyzi[4 * i + 1], z = kf.xyzi[4 * i + 2];
            for (int r = 0; r < 3; r++) kf.xyzi[4 * i + r] = (float)(kf.odom(r, 0) * x + kf.odom(r, 1) * y + kf.odom(r, 2) * z + kf.odom(r, 3));
        }
        out.push_back(std::move(kf));
    }
    std::stable_sort(out.begin(), out.end(), [](const KeyFrameDisk& a, const KeyFrameDisk& b) { return a.id < b.id; });
    return !out.empty();
}

// Localization::init (localization.cpp:91-131) without the graph back end and the global locator: the key frames of the map go to HBM once
bool loc_setup(Slam* s) {
    Loc& L = *s->loc;
    if (L.frames.empty()) return false;  // "Map Loader: error to load map" (setup_slam read the key frames from disk)
    uint64_t total = 0;
    uint32_t biggest = 0;
    for (const KeyFrameDisk& kf : L.frames) { total += kf.xyzi.size() / 4; biggest = std::max<uint32_t>(biggest, (uint32_t)(kf.xyzi.size() / 4)); }
    L.lm = lio_localmap_create(0, total + 16, 200000, std::max<uint32_t>(biggest, 1024));
    L.ndt = lio_ndt_create(0, 1.0f, 7, 200000ull + biggest + 1024, 400000, 262144);
    L.scan = lio_scan_create(0, 1u << 18, 1u << 18);
    if (!L.lm || !L.ndt || !L.scan) return false;
    lio_ndt_default_params(&L.par);
    L.par.max_process_time_ms = 50000;  // select_registration_method("NDT_CUDA", 50000) (hdl_localization_nodelet.cpp:47)
    for (const KeyFrameDisk& kf : L.frames) {
        const float pos[3] = {(float)kf.odom(0, 3), (float)kf.odom(1, 3), (float)kf.odom(2, 3)};
        if (lio_localmap_add_keyframe(L.lm, kf.xyzi.data(), (uint32_t)(kf.xyzi.size() / 4), pos) < 0) return false;
    }
    return true;
}

// one loop turn of Localization::runUpdateLocalMap for the pose just located (localization.cpp:303-373) -- synchronous: the key frames are
// resident in HBM, an update is a millisecond of device-to-device copies + VoxelGrid + target build, not a thread's worth of work
void loc_update_local_map(Loc& L, const Mat4& pose) {
    const double p[3] = {pose(0, 3), pose(1, 3), pose(2, 3)};
    int nk = 0;
    uint32_t npts = 0;
    const int rc = lio_localmap_update(L.lm, L.ndt, p, 10.0, 30.0, L.key_frame_distance, (float)std::max(L.resolution, 0.1), &nk, &npts);
    if (rc == 1) L.have_map = true;
    else if (rc == 2 || rc == 3) L.have_map = false;  // out of map / nearest key frame too far: the localizer is handed a null map
}

// HdlLocalizationNodelet::get_timed_pose(stamp) (hdl_localization_nodelet.cpp:108-155) for the two stamps the undistortion asks for
bool loc_timed_pose(Loc& L, uint64_t stamp, Mat4& out) {
    if (L.pose_data.empty()) return false;
    if (stamp < L.pose_data.front().first) return false;
    if (stamp > L.pose_data.back().first) {
        if (stamp - L.pose_data.back().first > 1000000) return false;
        if (!L.pe) return false;
        return lio_pose_estimator_predict_nostate(L.pe, stamp, out.m) >= 0;
    }
    if (L.pose_data.size() < 2) return false;
    out = L.pose_data.back().second;  // (a stamp inside the stored history: not reached by frame_callback, whose stamps are the newest)
    return true;
}

// HdlLocalizationNodelet::frame_callback (hdl_localization_nodelet.cpp:166-275) + Localization::feedPointData's bookkeeping (localization.cpp:235-275).
// The cloud (INS frame) is in L.staged / L.stamps.  Returns true when the pose is a localised one.
bool loc_localize(Slam* s, uint32_t n, uint64_t stamp, Mat4& pose_out) {
    Loc& L = *s->loc;
    if (!L.initialized) {
        if (!L.have_init_pose) return false;  // (the global locator -- GNSS / scan-context initial pose -- is out of scope: set_init_pose starts the filter)
        // Localization::initLocalizer -> initialpose_callback (hdl_localization_nodelet.cpp:304-317): normalised quaternion of the pose, cool time 0.2 s
        float ext[16];
        for (int i = 0; i < 16; i++) ext[i] = (float)s->T_imu.m[i];
        const Mat4& T = L.init_pose;
        double q[4];  // w x y z, Eigen's matrix -> quaternion
        {
            const double t = T(0, 0) + T(1, 1) + T(2, 2);
            if (t > 0) { const double r = std::sqrt(t + 1.0); q[0] = 0.5 * r; const double k = 0.5 / r; q[1] = (T(2, 1) - T(1, 2)) * k; q[2] = (T(0, 2) - T(2, 0)) * k; q[3] = (T(1, 0) - T(0, 1)) * k; }
            else {
                int i = 0;
                if (T(1, 1) > T(0, 0)) i = 1;
                if (T(2, 2) > T(i, i)) i = 2;
                const int j = (i + 1) % 3, k = (j + 1) % 3;
                const double r = std::sqrt(T(i, i) - T(j, j) - T(k, k) + 1.0);
                q[1 + i] = 0.5 * r;
                const double kk = 0.5 / r;
                q[0] = (T(k, j) - T(j, k)) * kk; q[1 + j] = (T(j, i) + T(i, j)) * kk; q[1 + k] = (T(k, i) + T(i, k)) * kk;
            }
            const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            for (double& v : q) v /= nq;
        }
        const float pos[3] = {(float)T(0, 3), (float)T(1, 3), (float)T(2, 3)};
        const float qf[4] = {(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
        if (L.pe) lio_pose_estimator_destroy(L.pe);
        L.pe = lio_pose_estimator_create(ext, stamp, pos, qf, 0.2);
        if (!L.pe) return false;
        L.init_stamp = stamp;
        L.pose_data.clear();
        L.imu_data.clear();
        L.last_odom = T;
        L.initialized = true;
        L.age = 0;
        L.failures = 0;
        lio_localmap_destroy(L.lm);  // a fresh local-map thread state: the first pose always builds a map
        L.lm = nullptr;
        uint64_t total = 0;
        uint32_t biggest = 0;
        for (const KeyFrameDisk& kf : L.frames) { total += kf.xyzi.size() / 4; biggest = std::max<uint32_t>(biggest, (uint32_t)(kf.xyzi.size() / 4)); }
        L.lm = lio_localmap_create(0, total + 16, 200000, std::max<uint32_t>(biggest, 1024));
        if (!L.lm) return false;
        for (const KeyFrameDisk& kf : L.frames) {
            const float kp[3] = {(float)kf.odom(0, 3), (float)kf.odom(1, 3), (float)kf.odom(2, 3)};
            lio_localmap_add_keyframe(L.lm, kf.xyzi.data(), (uint32_t)(kf.xyzi.size() / 4), kp);
        }
        loc_update_local_map(L, T);  // mPoseQueue.enqueue(mLastOdom)
    }
    if (n == 0) {  // "cloud is empty!!": the nodelet returns LocType::OTHER, which Localization::feedPointData counts like any failed frame
                   // (localization.cpp:245-254)
        L.failures++;
        if (L.failures >= 5) { L.initialized = false; L.have_init_pose = false; }
        return false;
    }
    // imu process: mean of the samples up to the frame's stamp
    float acc[3] = {0, 0, 0}, gyr[3] = {0, 0, 0};
    size_t used = 0;
    for (; used < L.imu_data.size(); used++) {
        if (stamp < (uint64_t)(L.imu_data[used].stamp * 1000000.0)) break;
        for (int a = 0; a < 3; a++) { acc[a] = acc[a] + L.imu_data[used].acc[a]; gyr[a] = gyr[a] + L.imu_data[used].gyr[a]; }
    }
    const bool use_imu = used != 0;
    if (use_imu) for (int a = 0; a < 3; a++) { acc[a] = acc[a] / (float)(int)used; gyr[a] = gyr[a] / (float)(int)used; }
    L.imu_data.erase(L.imu_data.begin(), L.imu_data.begin() + used);
    lio_pose_estimator_predict(L.pe, stamp, use_imu ? acc : nullptr, use_imu ? gyr : nullptr);
    // undistortion over the filter's step, then the downsample, on the device
    if (lio_scan_upload(L.scan, L.staged.data(), n) != LIO_OK) return false;
    Mat4 a, b;
    if (loc_timed_pose(L, stamp, a) && loc_timed_pose(L, stamp + lio_pose_estimator_get_dt(L.pe), b)) {
        const Mat4 d = mul(rigid_inverse(a), b);
        float df[16];
        for (int i = 0; i < 16; i++) df[i] = (float)d.m[i];
        lio_scan_undistort_delta(L.scan, L.stamps.data(), 0, df, s->scan_period);
    }
    uint32_t n_ds = 0;
    if (L.resolution >= 0.1) { if (lio_scan_voxel_downsample(L.scan, (float)L.resolution, 1, &n_ds) != LIO_OK) return false; }
    else { std::vector<float> raw(4 * (size_t)n); lio_scan_download_raw(L.scan, raw.data(), n); lio_scan_set_ds(L.scan, raw.data(), n); }
    // correct
    float obs[7], cov[49];
    bool result = false;
    double fitness = 0.0;
    if (L.have_map) {
        int it = 0;
        const int rc = lio_pose_estimator_match_gps(L.pe, L.ndt, L.scan, &L.par, nullptr, obs, cov, &it);
        result = rc == 1;
        if (stamp - L.init_stamp < 10000000) {  // WARM_UP_TIME: getFitnessScore(25) of the alignment (pose_estimator.cpp:254-264)
            double T[16] = {0, 0, 0, obs[0], 0, 0, 0, obs[1], 0, 0, 0, obs[2], 0, 0, 0, 1};
            const double w = obs[3], x = obs[4], y = obs[5], z = obs[6];
            T[0] = 1 - 2 * (y * y + z * z); T[1] = 2 * (x * y - z * w); T[2] = 2 * (x * z + y * w);
            T[4] = 2 * (x * y + z * w); T[5] = 1 - 2 * (x * x + z * z); T[6] = 2 * (y * z - x * w);
            T[8] = 2 * (x * z - y * w); T[9] = 2 * (y * z + x * w); T[10] = 1 - 2 * (x * x + y * y);
            uint32_t n_in = 0;
            lio_ndt_fitness_score(L.ndt, L.scan, T, 25.0, &fitness, &n_in);
        }
    } else {
        result = lio_pose_estimator_match_gps_only(L.pe, nullptr, obs, cov) == 1;
    }
    lio_pose_estimator_correct(L.pe, stamp, obs);
    float M[16];
    lio_pose_estimator_matrix(L.pe, M);
    for (int i = 0; i < 16; i++) pose_out.m[i] = (double)M[i];
    while (L.pose_data.size() >= 10) L.pose_data.pop_front();
    L.pose_data.emplace_back(stamp, pose_out);
    if (fitness > 1.0) result = false;  // "localization fitness score is too large"
    L.last_odom = pose_out;
    // Localization::feedPointData (localization.cpp:252-274)
    if (result) {
        L.age++;
        L.failures = 0;
        loc_update_local_map(L, pose_out);  // mPoseQueue.enqueue(mLastOdom)
    } else {
        L.failures++;
    }
    if (L.failures >= 5) {  // "failed to localize, fallback to initializing": back to waiting for an initial pose
        L.initialized = false;
        L.have_init_pose = false;
    }
    return true;
}

// HDL_FastLIO::runLio (fastlio.cpp:262-276)
void run_lio(Slam* s) {
    while (s->running.load()) {
        const int rc = lio_fastlio_main(s->engine);
        if (rc != LIO_MAIN_IDLE && rc >= 0) {
            double os[16], oe[16];
            lio_fastlio_odometry(s->engine, os, oe);
            Mat4 a, b;
            std::memcpy(a.m, os, sizeof(os));
            std::memcpy(b.m, oe, sizeof(oe));
            a = mul(mul(s->T_imu_ins_inv, a), s->T_imu_ins);
            b = mul(mul(s->T_imu_ins_inv, b), s->T_imu_ins);
            {
                std::lock_guard<std::mutex> lk(s->mtx);
                s->odom_queue.emplace_back(a, b);
            }
            s->cv.notify_all();
            continue;  // more may be queued
        }
        std::this_thread::sleep_for(std::chrono::microseconds(rc == LIO_MAIN_IDLE ? 200 : 2000));
    }
}

py::array_t<float> mat4_to_numpy_f32(const Mat4& T) {  // eigen_to_numpy(Matrix4d) -> float32 4 x 4 (py_utils.cpp:4-11)
    py::array_t<float> a({4, 4});
    auto r = a.mutable_unchecked<2>();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r(i, j) = (float)T(i, j);
    return a;
}

}  // namespace

// ---- hot path ---------------------------------------------------------------------------------------------------------------
py::list init_slam(const std::string mode, const std::string map_path, const std::string method, py::list& sensor_input, double resolution,
                   float dist_threshold, float degree_threshold, float frame_range) {
    (void)degree_threshold; (void)frame_range;
    // slam.py hands in method = "Localization" when the mode is not "mapping" (slam/slam.py:12); SLAM::SLAM then builds Locate::Localization
    const bool localization = mode == "localization" || method == "Localization";
    if (!localization && method != "FastLIO")
        throw std::runtime_error("slam_wrapper: only the FastLIO odometry path and the localisation mode are built on the device (method=" + method + ")");
    g.reset(new Slam());
    g->mode = localization ? "localization" : mode;
    g->method = method;
    g->map_path = map_path;
    std::vector<std::string> in;
    for (auto h : sensor_input) in.push_back(py::cast<std::string>(h));
    if (localization) {
        g->loc.reset(new Loc());
        g->loc->resolution = resolution;
        g->loc->key_frame_distance = dist_threshold;
        // Localization::setSensors (localization.cpp:154-182): RTK / IMU as they come, the first "n-" name is the lidar, the first other one the camera
        bool cam = false;
        for (auto& s : in) {
            g->sensors.push_back(s);
            if (s == "RTK") g->use_gps = true;
            else if (s == "IMU") g->use_imu = true;
            else if (s.length() < 2 || s[1] != '-') cam = cam || true;
            else if (g->lidar.empty()) g->lidar = s;
        }
        return py::cast(g->sensors);
    }
    // HDL_FastLIO::setSensors (fastlio.cpp:119-151): RTK and IMU first; without an IMU nothing else; the first "n-" name is the lidar
    bool imu = false;
    for (auto& s : in) {
        if (s == "RTK") { g->sensors.push_back(s); g->use_gps = true; }
        else if (s == "IMU") { g->sensors.push_back(s); imu = true; }
    }
    g->use_imu = imu;
    if (imu)
        for (auto& s : in) {
            if (s == "RTK" || s == "IMU") continue;
            g->sensors.push_back(s);
            if (!(s.length() < 2 || s[1] != '-') && g->lidar.empty()) g->lidar = s;
        }
    return py::cast(g->sensors);
}

void set_ins_external_param(double x, double y, double z, double yaw, double pitch, double roll) {
    require((bool)g, "init_slam first");
    g->T_static = transform_from_rpyt(x, y, z, yaw, pitch, roll);
}
void set_imu_external_param(double x, double y, double z, double yaw, double pitch, double roll) {
    require((bool)g, "init_slam first");
    g->T_imu = transform_from_rpyt(x, y, z, yaw, pitch, roll);
}

bool setup_slam() {
    require((bool)g, "init_slam first");
    if (g->loc) load_keyframes(g->map_path, g->loc->frames);  // host side of MapLoader: get_graph_map serves the key frames whatever follows
    if (lio_device_count() < 1) return false;  // the reference logs and returns false from setup() when the back end cannot start
    if (g->loc) return loc_setup(g.get());
    // HDL_FastLIO::init (fastlio.cpp:153-171): T_imu_ins = T_imu * T_static^-1 is the (INS-frame cloud) -> IMU extrinsic
    g->T_imu_ins = mul(g->T_imu, rigid_inverse(g->T_static));
    g->T_imu_ins_inv = rigid_inverse(g->T_imu_ins);
    g->engine = lio_engine_create(0, 0.5f, 75, g->max_points, g->max_voxels, 1u << 18, 100000);
    if (!g->engine) return false;
    const double extT[3] = {g->T_imu_ins(0, 3), g->T_imu_ins(1, 3), g->T_imu_ins(2, 3)};
    double extR[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) extR[i * 3 + j] = g->T_imu_ins(i, j);
    if (lio_fastlio_init(g->engine, extT, extR, 1, -1, g->scan_period, 1) != LIO_OK) return false;
    g->running.store(true);
    g->lio_thread = std::thread(run_lio, g.get());
    return true;
}

void deinit_slam() {
    if (!g) return;
    if (g->running.exchange(false) && g->lio_thread.joinable()) g->lio_thread.join();
    if (g->engine) lio_engine_destroy(g->engine);
    g->loc.reset(nullptr);
    g.reset(nullptr);
}

// SLAM::run, localisation branch (slam.cpp:273-367) over Localization::feedImuData / feedPointData / getPose
py::dict process_localization(Slam* s, py::dict& points, py::dict& points_attr, py::dict& rtk_dict, py::array_t<double>& imu_list, uint64_t timestamp) {
    Loc& L = *s->loc;
    int status = 0;
    if (rtk_dict.contains("Status")) status = py::cast<int>(rtk_dict["Status"]);
    std::string name = s->lidar;
    if (name.empty() || !points.contains(name.c_str())) {
        require(py::len(points) > 0, "process: no point cloud");
        name = py::cast<std::string>((*points.begin()).first);
    }
    py::array_t<float, py::array::c_style | py::array::forcecast> cloud = py::cast<py::array>(points[name.c_str()]);
    py::dict attr = py::cast<py::dict>(points_attr[name.c_str()]);
    py::array_t<float, py::array::c_style | py::array::forcecast> pattr = py::cast<py::array>(attr["points_attr"]);
    const uint64_t header_stamp = py::cast<uint64_t>(attr["timestamp"]);
    require(cloud.ndim() == 2 && cloud.shape(1) >= 4 && pattr.ndim() == 2 && pattr.shape(1) >= 1 && pattr.shape(0) == cloud.shape(0),
            "process: points must be N x 4 float32 with an N x 2 attribute array");
    const uint32_t n = (uint32_t)cloud.shape(0);
    const float* src = cloud.data();
    const float* asrc = pattr.data();
    const py::ssize_t cs = cloud.shape(1), as = pattr.shape(1);
    std::vector<Loc::Imu> imu_in;
    if (s->use_imu && imu_list.ndim() == 2 && imu_list.shape(1) >= 7) {  // numpy_to_imu (py_utils.cpp:244-258)
        auto ref = imu_list.unchecked<2>();
        for (py::ssize_t i = 0; i < ref.shape(0); i++) {
            Loc::Imu m;
            m.stamp = ref(i, 0) / 1000000.0;
            for (int a = 0; a < 3; a++) { m.gyr[a] = (float)(ref(i, 1 + a) / 180.0 * M_PI); m.acc[a] = (float)(ref(i, 4 + a) * 9.81); }
            imu_in.push_back(m);
        }
    }
    Mat4 out_pose = Mat4::identity();
    bool located = false, inited = false;
    {
        py::gil_scoped_release release;
        std::lock_guard<std::mutex> lk(s->mtx);
        if (L.initialized) L.imu_data.insert(L.imu_data.end(), imu_in.begin(), imu_in.end());  // Localization::feedImuData: dropped while not initialised
        // preprocessPoints (slam_base.h:83-85) straight into the staging vector
        L.staged.resize(4 * (size_t)n + 4);
        L.stamps.resize((size_t)n + 1);
        const Mat4& M = s->T_static;
        for (uint32_t i = 0; i < n; i++) {
            const double x = src[i * cs], y = src[i * cs + 1], z = src[i * cs + 2];
            L.staged[4 * i + 0] = (float)(M(0, 0) * x + M(0, 1) * y + M(0, 2) * z + M(0, 3));
            L.staged[4 * i + 1] = (float)(M(1, 0) * x + M(1, 1) * y + M(1, 2) * z + M(1, 3));
            L.staged[4 * i + 2] = (float)(M(2, 0) * x + M(2, 1) * y + M(2, 2) * z + M(2, 3));
            L.staged[4 * i + 3] = src[i * cs + 3];
            L.stamps[i] = (uint32_t)asrc[i * as];
        }
        located = loc_localize(s, n, header_stamp, out_pose);
        if (!located) out_pose = L.last_odom;
        inited = L.initialized && L.age >= 10;  // Localization::isInited: mInitialized && isStable()
    }
    double heading, pitch, roll;
    rpy_from_transform(out_pose, heading, pitch, roll);
    if (std::fabs(roll) >= 90.0 || std::fabs(pitch) >= 90.0) {
        const Mat4 o2 = transform_from_rpyt(out_pose(0, 3), out_pose(1, 3), out_pose(2, 3), -heading, pitch, roll);
        rpy_from_transform(o2, heading, pitch, roll);
    } else {
        heading = -heading;
    }
    if (heading < 0) heading += 360;
    py::dict pose;
    pose["latitude"] = 0.0;   // (no map origin / UTM projection without the GNSS side: slam.cpp:338-342)
    pose["longitude"] = 0.0;
    pose["altitude"] = 0.0;
    pose["heading"] = heading;
    pose["pitch"] = pitch;
    pose["roll"] = roll;
    pose["Ve"] = 0;
    pose["Vn"] = 0;
    pose["Vu"] = 0;
    pose["Status"] = status;
    pose["state"] = inited ? "Localizing(L)" : "Initializing";
    pose["timestamp"] = header_stamp;
    pose["odom_matrix"] = mat4_to_numpy_f32(out_pose);
    py::dict data;
    data["frame_start_timestamp"] = timestamp;
    data["pose"] = pose;
    data["slam_valid"] = true;
    return data;
}

py::dict process(py::dict& points, py::dict& points_attr, py::dict& image_dict, py::dict& image_stream_dict, py::dict& image_param,
                 py::dict& rtk_dict, py::array_t<double>& imu_list, uint64_t timestamp) {
    (void)image_dict; (void)image_stream_dict; (void)image_param;
    require(g && (g->engine || (g->loc && g->loc->scan)), "init_slam / setup_slam first");
    Slam* s = g.get();
    if (s->loc) return process_localization(s, points, points_attr, rtk_dict, imu_list, timestamp);
    // pydict_to_rtk: only the fields SLAM::run copies into the pose in mapping mode (slam.cpp:344-348)
    double lat = 0, lon = 0, alt = 0;
    int status = 0;
    if (rtk_dict.contains("latitude")) lat = py::cast<double>(rtk_dict["latitude"]);
    if (rtk_dict.contains("longitude")) lon = py::cast<double>(rtk_dict["longitude"]);
    if (rtk_dict.contains("altitude")) alt = py::cast<double>(rtk_dict["altitude"]);
    if (rtk_dict.contains("Status")) status = py::cast<int>(rtk_dict["Status"]);
    // SLAM::run (slam.cpp:283-296): preprocessInsData's validity test, then HDL_FastLIO::feedInsData -> fastlio_ins_enqueue (fastlio.cpp:185-187,
    // laserMapping.cpp:417-441): the INS velocity, ENU -> ego -> IMU, third component zeroed -- IMU initialisation seeds the state's velocity
    // from it (IMU_Processing.hpp:201-204), the wheel-speed rows read it
    if (s->use_gps && rtk_dict.contains("Ve") && rtk_dict.contains("timestamp")) {
        const uint64_t rtk_stamp = py::cast<uint64_t>(rtk_dict["timestamp"]);
        const std::string sensor = rtk_dict.contains("Sensor") ? py::cast<std::string>(rtk_dict["Sensor"]) : std::string();
        bool valid;
        if (status == 0 && std::fabs(lon) < 1e-4 && std::fabs(lat) < 1e-4) {
            const double lost = timestamp / 1000000.0 - s->last_ins_timestamp;
            if (s->last_ins_priority != -1 && lost >= 1.0) s->last_ins_priority = -1;
            valid = false;
        } else {
            const Slam::InsCfg* match = nullptr;
            for (const auto& c : s->ins_cfg) if (c.status == status) { match = &c; break; }
            if (!match) for (const auto& c : s->ins_cfg) if (c.status == -1) { match = &c; break; }
            const int mp = match ? match->priority : -1;
            int priority = -1;
            if (mp == s->last_ins_priority) { priority = mp; s->last_ins_timestamp = rtk_stamp / 1000000.0; }
            else if (mp < s->last_ins_priority) { priority = mp; s->last_ins_priority = mp; s->last_ins_timestamp = rtk_stamp / 1000000.0; }
            else {
                const double keep = rtk_stamp / 1000000.0 - s->last_ins_timestamp;
                if (keep >= match->stable_time) { priority = mp; s->last_ins_priority = mp; s->last_ins_timestamp = rtk_stamp / 1000000.0; }
                else priority = s->last_ins_priority;
            }
            valid = priority >= 0;
        }
        if (valid || sensor.find("Wheel") != std::string::npos) {
            const double hd = rtk_dict.contains("heading") ? py::cast<double>(rtk_dict["heading"]) : 0.0;
            const double pt = rtk_dict.contains("pitch") ? py::cast<double>(rtk_dict["pitch"]) : 0.0;
            const double rl = rtk_dict.contains("roll") ? py::cast<double>(rtk_dict["roll"]) : 0.0;
            const double ve[3] = {py::cast<double>(rtk_dict["Ve"]), rtk_dict.contains("Vn") ? py::cast<double>(rtk_dict["Vn"]) : 0.0,
                                  rtk_dict.contains("Vu") ? py::cast<double>(rtk_dict["Vu"]) : 0.0};
            const Mat4 Tve = rigid_inverse(transform_from_rpyt(0, 0, 0, -hd, pt, rl));
            double ego[3], vi[3];
            for (int r = 0; r < 3; r++) ego[r] = Tve(r, 0) * ve[0] + Tve(r, 1) * ve[1] + Tve(r, 2) * ve[2];
            for (int r = 0; r < 3; r++) vi[r] = s->T_imu_ins(r, 0) * ego[0] + s->T_imu_ins(r, 1) * ego[1] + s->T_imu_ins(r, 2) * ego[2];
            vi[2] = 0.0;  // "body up speed of INS is not accurate"
            lio_fastlio_ins_enqueue(s->engine, rtk_stamp / 1000000.0, vi);
        }
    }
    // numpy_to_imu + HDL_FastLIO::feedImuData
    if (s->use_imu && imu_list.ndim() == 2 && imu_list.shape(1) >= 7) {
        auto ref = imu_list.unchecked<2>();
        for (py::ssize_t i = 0; i < ref.shape(0); i++) {
            const double gyr[3] = {ref(i, 1) / 180.0 * M_PI, ref(i, 2) / 180.0 * M_PI, ref(i, 3) / 180.0 * M_PI};
            const double acc[3] = {ref(i, 4) * 9.81, ref(i, 5) * 9.81, ref(i, 6) * 9.81};
            lio_fastlio_imu_enqueue(s->engine, ref(i, 0) / 1000000.0, gyr, acc);
        }
    }
    // pydict_to_cloud + feedPointData: the lidar's cloud (fastlio.cpp:204-210 takes points[mLidarName])
    std::string name = s->lidar;
    if (name.empty() || !points.contains(name.c_str())) {
        require(py::len(points) > 0, "process: no point cloud");
        name = py::cast<std::string>((*points.begin()).first);
    }
    py::array_t<float, py::array::c_style | py::array::forcecast> cloud = py::cast<py::array>(points[name.c_str()]);
    py::dict attr = py::cast<py::dict>(points_attr[name.c_str()]);
    py::array_t<float, py::array::c_style | py::array::forcecast> pattr = py::cast<py::array>(attr["points_attr"]);
    const uint64_t header_stamp = py::cast<uint64_t>(attr["timestamp"]);
    require(cloud.ndim() == 2 && cloud.shape(1) >= 4 && pattr.ndim() == 2 && pattr.shape(1) >= 1 && pattr.shape(0) == cloud.shape(0),
            "process: points must be N x 4 float32 with an N x 2 attribute array");
    const uint32_t n = (uint32_t)cloud.shape(0);
    const float* src = cloud.data();
    const float* asrc = pattr.data();
    const py::ssize_t cs = cloud.shape(1), as = pattr.shape(1);
    Mat4 out_pose = Mat4::identity();
    bool got = false;
    {
        py::gil_scoped_release release;
        float* dst = nullptr;
        uint32_t* dstamp = nullptr;
        require(lio_fastlio_pcl_stage(s->engine, n, &dst, &dstamp) == LIO_OK, "process: staging failed");
        // pcl::transformPointCloud(in, out, Matrix4d): per point in double, terms left to right, cast to float (PCL 1.9.1 transforms.hpp)
        const Mat4& M = s->T_static;
        for (uint32_t i = 0; i < n; i++) {
            const double x = src[i * cs], y = src[i * cs + 1], z = src[i * cs + 2];
            dst[4 * i + 0] = (float)(M(0, 0) * x + M(0, 1) * y + M(0, 2) * z + M(0, 3));
            dst[4 * i + 1] = (float)(M(1, 0) * x + M(1, 1) * y + M(1, 2) * z + M(1, 3));
            dst[4 * i + 2] = (float)(M(2, 0) * x + M(2, 1) * y + M(2, 2) * z + M(2, 3));
            dst[4 * i + 3] = src[i * cs + 3];
            dstamp[i] = (uint32_t)asrc[i * as];  // pointcloud_attr[i].stamp = ref_attr(i, 0): float -> uint32_t
        }
        require(lio_fastlio_pcl_commit(s->engine, n, (double)header_stamp / 1000000.0) == LIO_OK, "process: enqueue failed");
        // HDL_FastLIO::getPose: wait for the LIO thread's odometry (10 s), then get_odom2map() (identity without a graph back end) * odom.first
        std::unique_lock<std::mutex> lk(s->mtx);
        got = s->cv.wait_for(lk, std::chrono::seconds(10), [&] { return !s->odom_queue.empty(); });
        if (got) {
            out_pose = s->odom_queue.front().first;
            s->odom_queue.pop_front();
        }
    }
    // SLAM::run, mapping branch (slam.cpp:344-364)
    double heading, pitch, roll;
    rpy_from_transform(out_pose, heading, pitch, roll);
    if (std::fabs(roll) >= 90.0 || std::fabs(pitch) >= 90.0) {
        // (the reference rebuilds a LOCAL copy of the odometry here; pose.T was assigned before and keeps the filter's matrix)
        const Mat4 o2 = transform_from_rpyt(out_pose(0, 3), out_pose(1, 3), out_pose(2, 3), -heading, pitch, roll);
        rpy_from_transform(o2, heading, pitch, roll);
    } else {
        heading = -heading;
    }
    if (heading < 0) heading += 360;
    py::dict pose;
    pose["latitude"] = lat;
    pose["longitude"] = lon;
    pose["altitude"] = alt;
    pose["heading"] = heading;
    pose["pitch"] = pitch;
    pose["roll"] = roll;
    pose["Ve"] = 0;
    pose["Vn"] = 0;
    pose["Vu"] = 0;
    pose["Status"] = status;
    pose["state"] = "Mapping";
    pose["timestamp"] = header_stamp;
    pose["odom_matrix"] = mat4_to_numpy_f32(out_pose);
    py::dict data;
    data["frame_start_timestamp"] = timestamp;
    data["pose"] = pose;
    data["slam_valid"] = true;
    return data;
}

// ---- off the hot path: type-correct minimal implementations -------------------------------------------
void set_camera_param(py::list& cameras) { (void)cameras; }
// pydict_to_ins_config (py_utils.cpp:295-317): ins_normal / ins_float / ins_fix entries with use, status, stable_time, precision
void set_ins_config(py::dict& dict) {
    if (!g) return;
    g->ins_cfg.clear();
    int priority = 0;
    for (const char* name : {"ins_normal", "ins_float", "ins_fix"}) {
        if (!dict.contains(name)) continue;
        py::dict e = py::cast<py::dict>(dict[name]);
        if (!e.contains("use") || !py::cast<bool>(e["use"])) continue;
        Slam::InsCfg c;
        c.name = name;
        c.status = py::cast<int>(e["status"]);
        c.stable_time = py::cast<double>(e["stable_time"]);
        c.precision = py::cast<double>(e["precision"]);
        c.priority = priority++;
        g->ins_cfg.push_back(c);
    }
}
void set_init_pose(double x, double y, double z, double yaw, double pitch, double roll) {
    if (!g) return;
    const double v[6] = {x, y, z, yaw, pitch, roll};
    std::memcpy(g->init_pose, v, sizeof(v));
    if (g->loc) {  // SLAM::setInitPose (slam.cpp): getTransformFromRPYT(x, y, z, yaw, pitch, roll) -> Localization::setInitPose: mInitialized = false
        std::lock_guard<std::mutex> lk(g->mtx);
        g->loc->init_pose = transform_from_rpyt(x, y, z, yaw, pitch, roll);
        g->loc->have_init_pose = true;
        g->loc->initialized = false;
    }
}
py::list get_estimate_pose(double x0, double y0, double x1, double y1) {
    (void)x0; (void)y0; (void)x1; (void)y1;
    return py::cast(std::vector<double>{0, 0, 0, 0, 0, 0, 0});  // x, y, z, roll, pitch, -yaw, result (0 = no estimate)
}
void set_destination(bool enable, std::string dest, int port) { (void)enable; if (g) { g->dest = dest; g->dest_port = port; } }
py::dict update_odom() {
    py::dict d;
    d["odoms"] = py::dict();
    d["keyframes"] = py::list();
    return d;
}
py::dict get_graph_status() {
    py::dict d;
    d["loop_detected"] = false;
    return d;
}
py::array_t<double> get_map_origin() {
    py::array_t<double> a({1, 7});
    auto r = a.mutable_unchecked<2>();
    for (int i = 0; i < 6; i++) r(0, i) = g ? g->origin[i] : 0.0;
    r(0, 6) = 0;
    return a;
}
void set_map_origin(double lat, double lon, double alt, double heading, double pitch, double roll) {
    if (g && !g->origin_set) {  // SLAM::setOrigin: first one wins
        const double v[6] = {lat, lon, alt, heading, pitch, roll};
        std::memcpy(g->origin, v, sizeof(v));
        g->origin_set = true;
    }
}
py::dict merge_map(const std::string& directory) { (void)directory; return py::dict(); }
// get_graph_map (slam_wrapper.cpp:180-184 -> keyframe_to_pydict, py_utils.cpp:272-293): the key frames of the loaded map in localisation mode
// (points N x 4 f32 as stored, pose 4 x 4 f32, stamp); empty in mapping mode (the graph back end is out of scope)
py::dict get_graph_map() {
    py::dict d;
    py::dict points, images, poses, stamps;
    static const std::vector<KeyFrameDisk> none;
    for (const KeyFrameDisk& kf : (g && g->loc) ? g->loc->frames : none) {
        const std::string id = std::to_string(kf.id);
        const py::ssize_t n = (py::ssize_t)(kf.local.size() / 4);
        py::array_t<float> a({n, (py::ssize_t)4});
        auto r = a.mutable_unchecked<2>();
        for (py::ssize_t i = 0; i < n; i++) { r(i, 0) = kf.local[4 * i]; r(i, 1) = kf.local[4 * i + 1]; r(i, 2) = kf.local[4 * i + 2]; r(i, 3) = kf.local[4 * i + 3]; }
        points[id.c_str()] = a;
        poses[id.c_str()] = mat4_to_numpy_f32(kf.odom);
        stamps[id.c_str()] = kf.stamp;
        images[id.c_str()] = py::dict();
    }
    d["points"] = points;
    d["images"] = images;
    d["poses"] = poses;
    d["stamps"] = stamps;
    return d;
}
py::array_t<float> get_color_map() { return py::array_t<float>(std::vector<py::ssize_t>{0, 6}); }
py::dict get_graph_edges() { return py::dict(); }
py::dict get_graph_meta() { return py::dict(); }
// pointcloud_align (graph_utils.cpp:20-46): Generalized-ICP of two clouds from a guess, the guess's translation reset when it is 50 m or
// more.  The reference calls PCL's GeneralizedIterativeClosestPoint (third-party, source not in the tree) with 20 neighbours, 64 iterations,
// transformation epsilon 1e-2; here the same cost on the device (lio_gicp_*, the FastGICP formulation the reference's other GICP call
// sites use) with those settings and PCL's default correspondence distance for that class (5 m).  Clouds too small for 20 neighbours, or
// a failed alignment, return the (sanitised) guess.
py::array_t<float> pointcloud_align(py::array_t<float>& source_point, py::array_t<float>& target_point, py::array_t<float>& guess) {
    py::array_t<float> out({4, 4});
    auto o = out.mutable_unchecked<2>();
    auto gi = guess.unchecked<2>();
    double G[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) G[i * 4 + j] = (double)(float)gi(i, j);
    const double d = std::sqrt(G[3] * G[3] + G[7] * G[7] + G[11] * G[11]);
    if (d >= 50.0) { G[3] = 0; G[7] = 0; G[11] = 0; }
    for (int i = 0; i < 16; i++) o(i / 4, i % 4) = (float)G[i];
    auto src = py::array_t<float, py::array::c_style | py::array::forcecast>::ensure(source_point);
    auto tgt = py::array_t<float, py::array::c_style | py::array::forcecast>::ensure(target_point);
    if (!src || !tgt || src.ndim() != 2 || tgt.ndim() != 2 || src.shape(1) != 4 || tgt.shape(1) != 4 || src.shape(0) < 20 || tgt.shape(0) < 20) return out;
    const uint32_t ns = (uint32_t)src.shape(0), nt = (uint32_t)tgt.shape(0);
    lio_gicp* m = lio_gicp_create(0, 1.0f, std::max(ns, nt), 20);
    if (!m) return out;
    double T[16];
    int it = 0, conv = 0;
    lio_ndt_params prm;
    lio_ndt_default_params(&prm);
    prm.transformation_epsilon = 1e-2;
    prm.rotation_epsilon_deg = 1e-2;
    prm.max_iterations = 64;
    prm.max_process_time_ms = -1;
    int rc;
    {
        py::gil_scoped_release nogil;
        rc = lio_gicp_set_target(m, tgt.data(), nt);
        if (rc == LIO_OK) rc = lio_gicp_set_source(m, src.data(), ns);
        if (rc == LIO_OK) rc = lio_gicp_align(m, G, &prm, 5.0, T, &it, &conv);
        lio_gicp_destroy(m);
    }
    if (rc == LIO_OK)
        for (int i = 0; i < 16; i++) o(i / 4, i % 4) = (float)T[i];
    return out;
}
void set_mapping_ground_constraint(bool enable) { if (g) g->ground_constraint = enable; }
bool get_mapping_ground_constraint() { return g ? g->ground_constraint : false; }
void set_mapping_constraint(bool loop_closure, bool gravity_constraint) { if (g) { g->loop_closure = loop_closure; g->gravity_constraint = gravity_constraint; } }
void set_map_colouration(bool enable) { if (g) g->colouration = enable; }
void del_graph_vertex(int id) { (void)id; }
void add_graph_edge(py::array_t<float>& prev, int prev_id, py::array_t<float>& next, int next_id, py::array_t<float>& relative) {
    (void)prev; (void)prev_id; (void)next; (void)next_id; (void)relative;
}
void del_graph_edge(int id) { (void)id; }
void set_graph_vertex_fix(int id, bool fix) { (void)id; (void)fix; }
py::dict run_graph_optimization() { return py::dict(); }
py::dict run_robust_graph_optimization(std::string mode) { (void)mode; return py::dict(); }
// dump_keyframe (graph_utils.cpp:123-131): KeyFrame(stamp, id, pose, numpy_to_pointcloud(points, 255.0)).save(directory) -- `data` + `cloud.pcd`,
// the files the localisation mode loads its map from
void dump_keyframe(const std::string& directory, uint64_t stamp, int id, py::array_t<float>& points_input, py::array_t<float>& pose_input) {
    auto pts = py::array_t<float, py::array::c_style | py::array::forcecast>::ensure(points_input);
    auto pose = py::array_t<float, py::array::c_style | py::array::forcecast>::ensure(pose_input);
    if (!pts || !pose || pts.ndim() != 2 || pts.shape(1) < 4 || pose.size() != 16) return;
    const size_t n = (size_t)pts.shape(0);
    std::vector<float> xyzi(4 * n + 4);
    const float* src = pts.data();
    const py::ssize_t cs = pts.shape(1);
    for (size_t i = 0; i < n; i++) { xyzi[4 * i] = src[i * cs]; xyzi[4 * i + 1] = src[i * cs + 1]; xyzi[4 * i + 2] = src[i * cs + 2]; xyzi[4 * i + 3] = src[i * cs + 3] * 255.0f; }
    Mat4 T;
    for (int i = 0; i < 16; i++) T.m[i] = (double)pose.data()[i];
    write_pcd_binary(directory + "/cloud.pcd", xyzi.data(), n);
    write_keyframe_data(directory, stamp, id, T);
}
void dump_odometry(const std::string& directory) { (void)directory; }
void set_export_map_config(double z_min, double z_max, std::string color) { if (g) { g->export_z[0] = z_min; g->export_z[1] = z_max; g->export_color = color; } }
void export_points(py::array_t<float>& points_input, py::array_t<float>& odom_input) { (void)points_input; (void)odom_input; }
void dump_map_points(std::string file) { (void)file; }
py::list dump_graph(const std::string& directory) { (void)directory; return py::list(); }
py::list align_pose(double stamp1, double stamp2, py::array_t<double>& estimate1_py, py::array_t<double>& estimate2_py, py::array_t<double>& poses_stamp,
                    py::list& poses_py) {
    (void)stamp1; (void)stamp2; (void)estimate1_py; (void)estimate2_py; (void)poses_stamp;
    return poses_py;
}
void save_undistortion_cloud(std::string file, py::array_t<float>& points, py::dict& points_attr, py::array_t<double>& poses) {
    (void)file; (void)points; (void)points_attr; (void)poses;
}
void accumulate_cloud(py::array_t<float>& points, py::dict& points_attr, py::array_t<double>& poses, std::string odometry_type, bool extract_ground) {
    (void)points; (void)points_attr; (void)poses; (void)odometry_type; (void)extract_ground;
}
void save_accumulate_cloud(std::string file, double resolution) { (void)file; (void)resolution; }
void texture_mesh(std::string mesh_path, std::string cloud_path, std::string output_path) { (void)mesh_path; (void)cloud_path; (void)output_path; }
void set_colouration_config(py::list& cameras) { (void)cameras; }
void set_map_odometrys(py::array_t<double>& poses) { (void)poses; }
void colouration_frame(std::string lidar_name, py::dict& points, py::dict& points_attr, py::dict& image_dict, py::dict& image_stream_dict, py::dict& image_param) {
    (void)lidar_name; (void)points; (void)points_attr; (void)image_dict; (void)image_stream_dict; (void)image_param;
}
void save_render_cloud(std::string file) { (void)file; }

// test visibility (not in the reference's module): the engine behind the module, so that a test can hold the C++ path against the C ABI
uintptr_t _engine_handle() { return g ? reinterpret_cast<uintptr_t>(g->engine) : 0; }
py::array_t<double> _transform_from_rpyt(double x, double y, double z, double yaw, double pitch, double roll) {  // the module's getTransformFromRPYT
    const Mat4 T = transform_from_rpyt(x, y, z, yaw, pitch, roll);
    py::array_t<double> a({4, 4});
    auto r = a.mutable_unchecked<2>();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r(i, j) = T(i, j);
    return a;
}
void _set_capacity(uint64_t max_points, uint64_t max_voxels) { if (g) { g->max_points = max_points; g->max_voxels = max_voxels; } }

PYBIND11_MODULE(slam_wrapper, m) {
    m.doc() = "mapping python interface (MI355X-native LIO core behind the reference's slam_wrapper surface)";
    m.def("init_slam", &init_slam, "init slam", py::arg("mode"), py::arg("map_path"), py::arg("method"), py::arg("sensor_input"), py::arg("resolution"),
          py::arg("dist_threshold"), py::arg("degree_threshold"), py::arg("frame_range"));
    m.def("setup_slam", &setup_slam, py::call_guard<py::gil_scoped_release>());
    m.def("deinit_slam", &deinit_slam, "deinit slam");
    m.def("set_camera_param", &set_camera_param, "set camera parameters", py::arg("cameras"));
    m.def("set_ins_external_param", &set_ins_external_param, "set ins external param", py::arg("x"), py::arg("y"), py::arg("z"), py::arg("yaw"), py::arg("pitch"),
          py::arg("roll"));
    m.def("set_imu_external_param", &set_imu_external_param, "set imu external param", py::arg("x"), py::arg("y"), py::arg("z"), py::arg("yaw"), py::arg("pitch"),
          py::arg("roll"));
    m.def("set_ins_config", &set_ins_config, "set ins config", py::arg("dict"));
    m.def("set_init_pose", &set_init_pose, "set init pose", py::arg("x"), py::arg("y"), py::arg("z"), py::arg("yaw"), py::arg("pitch"), py::arg("roll"));
    m.def("get_estimate_pose", &get_estimate_pose, "get estimate pose", py::arg("x0"), py::arg("y0"), py::arg("x1"), py::arg("y1"));
    m.def("set_destination", &set_destination, "set destination", py::arg("enable"), py::arg("dest"), py::arg("port"));
    m.def("process", &process, "process", py::arg("points"), py::arg("points_attr"), py::arg("image_dict"), py::arg("image_stream_dict"), py::arg("image_param"),
          py::arg("rtk_dict"), py::arg("imu_list"), py::arg("timestamp"));
    m.def("update_odom", &update_odom, "update odom");
    m.def("get_graph_status", &get_graph_status, "get graph status");
    m.def("get_map_origin", &get_map_origin, "get map origin");
    m.def("set_map_origin", &set_map_origin, "set map origin", py::arg("lat"), py::arg("lon"), py::arg("alt"), py::arg("heading"), py::arg("pitch"), py::arg("roll"));
    m.def("get_color_map", &get_color_map, "get color map");
    m.def("merge_map", &merge_map, "merge map", py::arg("directory"));
    m.def("get_graph_map", &get_graph_map, "get graph map");
    m.def("get_graph_edges", &get_graph_edges, "get graph edges");
    m.def("pointcloud_align", &pointcloud_align, "pointcloud align", py::arg("source_point"), py::arg("target_point"), py::arg("guess"));
    m.def("set_mapping_ground_constraint", &set_mapping_ground_constraint, "set mapping ground constraint", py::arg("enable"));
    m.def("get_mapping_ground_constraint", &get_mapping_ground_constraint, "get mapping ground constraint");
    m.def("set_mapping_constraint", &set_mapping_constraint, "set mapping constraint", py::arg("loop_closure"), py::arg("gravity_constraint"));
    m.def("set_map_colouration", &set_map_colouration, "set map colouration", py::arg("enable"));
    m.def("get_graph_meta", &get_graph_meta, "get graph meta");
    m.def("del_graph_vertex", &del_graph_vertex, "del graph vertex", py::arg("id"));
    m.def("add_graph_edge", &add_graph_edge, "add graph edge", py::arg("prev"), py::arg("prev_id"), py::arg("next"), py::arg("next_id"), py::arg("relative"));
    m.def("del_graph_edge", &del_graph_edge, "del graph edge", py::arg("id"));
    m.def("set_graph_vertex_fix", &set_graph_vertex_fix, "set graph vertex fix", py::arg("id"), py::arg("fix"));
    m.def("run_graph_optimization", &run_graph_optimization, "run graph optimization");
    m.def("run_robust_graph_optimization", &run_robust_graph_optimization, "run robust graph optimization", py::arg("mode"));
    m.def("dump_keyframe", &dump_keyframe, "dump keyframe", py::arg("directory"), py::arg("stamp"), py::arg("id"), py::arg("points_input"), py::arg("pose_input"));
    m.def("dump_odometry", &dump_odometry, "dump odometry", py::arg("directory"));
    m.def("set_export_map_config", &set_export_map_config, "set export map config", py::arg("z_min"), py::arg("z_max"), py::arg("color"));
    m.def("export_points", &export_points, "export points", py::arg("points_input"), py::arg("odom_input"));
    m.def("dump_map_points", &dump_map_points, "dump map points", py::arg("file"));
    m.def("dump_graph", &dump_graph, "dump graph", py::arg("directory"));
    m.def("align_pose", &align_pose, "align pose", py::arg("stamp1"), py::arg("stamp2"), py::arg("estimate1_py"), py::arg("estimate2_py"), py::arg("poses_stamp"),
          py::arg("poses_py"));
    m.def("save_undistortion_cloud", &save_undistortion_cloud, "save undistortion cloud", py::arg("file"), py::arg("points"), py::arg("points_attr"), py::arg("poses"));
    m.def("accumulate_cloud", &accumulate_cloud, "accumulate cloud", py::arg("points"), py::arg("points_attr"), py::arg("poses"), py::arg("odometry_type"),
          py::arg("extract_ground"));
    m.def("save_accumulate_cloud", &save_accumulate_cloud, "save accumulate cloud", py::arg("file"), py::arg("resolution"));
    m.def("texture_mesh", &texture_mesh, "texture mesh", py::arg("mesh_path"), py::arg("cloud_path"), py::arg("output_path"));
    m.def("set_colouration_config", &set_colouration_config, "set colouration config", py::arg("cameras"));
    m.def("set_map_odometrys", &set_map_odometrys, "set map odometrys", py::arg("poses"));
    m.def("colouration_frame", &colouration_frame, "colouration frame", py::arg("lidar_name"), py::arg("points"), py::arg("points_attr"), py::arg("image_dict"),
          py::arg("image_stream_dict"), py::arg("image_param"));
    m.def("save_render_cloud", &save_render_cloud, "save render cloud", py::arg("file"));
    m.def("_engine_handle", &_engine_handle);
    m.def("_transform_from_rpyt", &_transform_from_rpyt);
    m.def("_set_capacity", &_set_capacity, py::arg("max_points"), py::arg("max_voxels"));
}
