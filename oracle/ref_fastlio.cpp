// oracle/ref_fastlio.cpp -- the reference's OWN FastLIO translation units, compiled whole from where they lie under
// /root/reference (nothing is copied): slam/mapping/fastlio/src/laserMapping.cpp (fastlio_init / imu_enqueue /
// pcl_enqueue / ins_enqueue / sync_packages / h_share_model / map_incremental / fastlio_main / fastlio_odometry),
// src/IMU_Processing.hpp (IMU_init, UndistortPcl), src/preprocess.cpp (velodyne_handler), include/ikd-Tree/ikd_Tree.cpp,
// the iVox map, IKFoM and MTK.  What is NOT the reference's: the shims under oracle/ref_shims (PCL containers, Boost,
// logging, the plain data types of mapping_types.h) and pcl::VoxelGrid, which is routed to the oracle's restatement so
// that both sides see the same downsampled cloud -- pcl::VoxelGrid is the one stage this harness cannot pin.
// TEST INFRASTRUCTURE ONLY: built into oracle/_ref/libref_fastlio.so by `make -C oracle ref`.
#include "ref_shims/lsd_ikfom_manifolds.h"

#include <src/laserMapping.cpp>  // -I$(REF)/slam/mapping/fastlio; preprocess.cpp and ikd_Tree.cpp are compiled beside it (Makefile)

// slam_utils.cpp:89-96 restated (that file needs UTM / system helpers): translation * Rz(yaw) * Rx(pitch) * Ry(roll), degrees
Eigen::Matrix4d getTransformFromRPYT(double x, double y, double z, double yaw, double pitch, double roll) {
    const double d2r = 0.01745329251994;
    Eigen::Affine3d T = Eigen::Translation3d(x, y, z) * Eigen::AngleAxisd(yaw * d2r, Eigen::Vector3d::UnitZ()) *
                        Eigen::AngleAxisd(pitch * d2r, Eigen::Vector3d::UnitX()) * Eigen::AngleAxisd(roll * d2r, Eigen::Vector3d::UnitY());
    return T.matrix();
}

static void state_to26(const state_ikfom& x, double* s) {  // pos3 rot4(xyzw) ril4 til3 vel3 bg3 ba3 grav3
    for (int i = 0; i < 3; i++) { s[i] = x.pos[i]; s[11 + i] = x.offset_T_L_I[i]; s[14 + i] = x.vel[i]; s[17 + i] = x.bg[i]; s[20 + i] = x.ba[i]; s[23 + i] = x.grav.vec[i]; }
    for (int i = 0; i < 4; i++) { s[3 + i] = x.rot.coeffs()[i]; s[7 + i] = x.offset_R_L_I.coeffs()[i]; }
}
static int cloud_out(const PointCloudXYZI::Ptr& c, float* out, int cap, int stride8) {
    const int n = c ? (int)c->points.size() : 0;
    for (int i = 0; i < n && i < cap; i++) {
        const PointType& p = c->points[i];
        if (stride8) { float* o = out + 8 * (size_t)i; o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = p.intensity; o[4] = p.curvature; o[5] = p.normal_x; o[6] = p.normal_y; o[7] = p.normal_z; }
        else { float* o = out + 4 * (size_t)i; o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = p.intensity; }
    }
    return n;
}

// instrumentation: the filter is given a wrapper that calls the reference's h_share_model and keeps what each call saw and produced
struct HCall {
    double s26[26];
    int converge, valid, n_eff, degenerate;
    double total_residual, HtH[36], Hth[6];
};
static std::vector<HCall> g_calls;
static int g_canonical = 0, g_log = 1;
extern "C" void ref_fl_canonical_neighbours();
// canonical-order mode: a search pass has just refreshed Nearest_Points; order every list canonically, restore the selection flags to
// what the search branch sets (laserMapping.cpp:850) and linearise again WITHOUT a new search -- the reference's own code on the
// neighbour order the oracle and the HIP kernels use
static void canonical_relinearise(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d) {
    ref_fl_canonical_neighbours();
    for (int i = 0; i < feats_down_size; i++) point_selected_surf[i] = Nearest_Points[i].size() < NUM_MATCH_POINTS ? false : true;
    d.converge = false;
    d.valid = true;
    h_share_model(s, d);
    d.converge = true;
}
static void h_share_logged(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d) {
    if (!g_log && !g_canonical) { h_share_model(s, d); return; }  // timing runs: the reference's model and nothing else
    HCall c;
    state_to26(s, c.s26);
    c.converge = d.converge ? 1 : 0;
    h_share_model(s, d);
    if (g_canonical && d.converge) canonical_relinearise(s, d);
    c.valid = d.valid ? 1 : 0;
    c.n_eff = effct_feat_num;
    c.degenerate = is_degenerate ? 1 : 0;
    c.total_residual = total_residual;
    std::fill(c.HtH, c.HtH + 36, 0.0);
    std::fill(c.Hth, c.Hth + 6, 0.0);
    if (d.valid)
        for (int r = 0; r < d.h_x.rows(); r++)
            for (int a = 0; a < 6; a++) {
                c.Hth[a] += d.h_x(r, a) * d.h(r);
                for (int b = 0; b < 6; b++) c.HtH[6 * a + b] += d.h_x(r, a) * d.h_x(r, b);
            }
    g_calls.push_back(c);
}

extern "C" {

// one direct call of the reference's h_share_model on the scan and map the globals hold now, at a state given from outside; converge:
// 0 = reuse the neighbour lists, 1 = search, 2 = search, then canonical order and re-linearisation
// (the resizes are those of fastlio_main, laserMapping.cpp:1214-1222).  Per point: selected flag, plane (a b c pd2), the <=5 neighbours.
int ref_fl_h_share(const double* s26, int converge, uint8_t* selected, float* normvec4, float* nn5x4, int* nn_cnt, double* rows12, double* h, int cap) {
    state_ikfom s;
    s.pos = vect3(Eigen::Vector3d(s26[0], s26[1], s26[2]));
    s.rot.coeffs() = Eigen::Vector4d(s26[3], s26[4], s26[5], s26[6]);
    s.offset_R_L_I.coeffs() = Eigen::Vector4d(s26[7], s26[8], s26[9], s26[10]);
    s.offset_T_L_I = vect3(Eigen::Vector3d(s26[11], s26[12], s26[13]));
    s.vel = vect3(Eigen::Vector3d(s26[14], s26[15], s26[16]));
    s.bg = vect3(Eigen::Vector3d(s26[17], s26[18], s26[19]));
    s.ba = vect3(Eigen::Vector3d(s26[20], s26[21], s26[22]));
    s.grav.vec = Eigen::Vector3d(s26[23], s26[24], s26[25]);
    feats_down_size = feats_down_body->points.size();
    normvec->resize(feats_down_size);
    feats_down_world->resize(feats_down_size);
    Nearest_Points.resize(feats_down_size);
    esekfom::dyn_share_datastruct<double> d;
    d.valid = true;
    d.converge = converge != 0;
    h_share_model(s, d);
    if (converge == 2) canonical_relinearise(s, d);  // search pass in canonical order
    for (int i = 0; i < feats_down_size; i++) {
        selected[i] = point_selected_surf[i] ? 1 : 0;
        const PointType& nv = normvec->points[i];
        normvec4[4 * i] = nv.x; normvec4[4 * i + 1] = nv.y; normvec4[4 * i + 2] = nv.z; normvec4[4 * i + 3] = nv.intensity;
        nn_cnt[i] = (int)Nearest_Points[i].size();
        for (int j = 0; j < 5 && j < nn_cnt[i]; j++) {
            const PointType& q = Nearest_Points[i][j];
            float* o = nn5x4 + 20 * (size_t)i + 4 * j;
            o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.intensity;
        }
    }
    if (!d.valid) return -1;
    const int n = (int)d.h.rows();
    for (int r = 0; r < n && r < cap; r++) {
        for (int c = 0; c < 12; c++) rows12[12 * (size_t)r + c] = d.h_x(r, c);
        h[r] = d.h(r);
    }
    return n;
}

// put the neighbour lists the last search pass refreshed into the oracle's canonical total order (d2, x, y, z).  The reference leaves
// elements 1..4 in whatever order std::nth_element produced (ivox3d.h:159-164); esti_plane's QR then differs in the last bits with
// the row order.  A list the search did NOT refresh (no candidate in the stencil: GetClosestPoint returns false and the stale content
// survives, ivox3d.h:155-157) keeps its order -- whether a list was refreshed is asked from the reference's own GetClosestPoint.
// After this call a converge = 0 pass re-linearises on the canonically ordered lists, which is what the oracle and the HIP kernels compute.
void ref_fl_canonical_neighbours() {
    for (int i = 0; i < feats_down_size && i < (int)Nearest_Points.size(); i++) {
        const PointType& q = feats_down_world->points[i];
        PointVector probe;
        if (!ivox->GetClosestPoint(q, probe, NUM_MATCH_POINTS, 5)) continue;
        auto& v = Nearest_Points[i];
        std::sort(v.begin(), v.end(), [&q](const PointType& a, const PointType& b) {
            const float da = faster_lio::distance2(a, q), db = faster_lio::distance2(b, q);
            if (da != db) return da < db;
            if (a.x != b.x) return a.x < b.x;
            if (a.y != b.y) return a.y < b.y;
            return a.z < b.z;
        });
    }
}

void ref_fl_set_canonical(int on) { g_canonical = on; }
void ref_fl_set_logging(int on) { g_log = on; }
void ref_fl_set_wheelspeed(int on) { wheelspeed_en = on != 0; }  // the file-scope constant of laserMapping.cpp:83 (never set by the reference itself)

// ---- the registration step alone, for bench.py's cpu_baseline ("reference") and full-size parity: a static map, a raw cloud, a prior.
// What runs is the reference's: IVox::AddPoints, SetNearByType, the block of fastlio_main between the downsample and the filter update
// (laserMapping.cpp:1204-1222, 1266-1272) and esekf::update_iterated_dyn_share_modified with h_share_model.
int ref_fl_map_add(const float* xyzi, int n) {
    PointVector v(n);
    for (int i = 0; i < n; i++) { v[i].x = xyzi[4 * i]; v[i].y = xyzi[4 * i + 1]; v[i].z = xyzi[4 * i + 2]; v[i].intensity = xyzi[4 * i + 3]; }
    ivox->AddPoints(v, travel_distance);
    return (int)ivox->NumValidGrids();
}
// every point the map holds, voxel by voxel (the hash map's iteration order), inside a voxel in push_back order (IVox::GetAllPoints, ivox3d.h:263-270);
// returns the count (the first `cap` are written)
int ref_fl_map_dump(float* xyzi, int cap) {
    PointVector v;
    ivox->GetAllPoints(v);
    const int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) { xyzi[4 * i] = v[i].x; xyzi[4 * i + 1] = v[i].y; xyzi[4 * i + 2] = v[i].z; xyzi[4 * i + 3] = v[i].intensity; }
    return n;
}
void ref_fl_set_nearby(int n) {  // 18 / 26 / 74 (fastlio_main switches NEARBY74 -> NEARBY18 one second after the first scan)
    ivox->SetNearByType(n == 18 ? IVoxType::NearbyType::NEARBY18 : n == 26 ? IVoxType::NearbyType::NEARBY26 : IVoxType::NearbyType::NEARBY74);
}
int ref_fl_register(const float* raw_xyzi, int n, const double* s26, const double* P529, double* s26_out, double* P529_out) {
    feats_undistort->points.resize(n);
    for (int i = 0; i < n; i++) {
        PointType& p = feats_undistort->points[i];
        p.x = raw_xyzi[4 * i]; p.y = raw_xyzi[4 * i + 1]; p.z = raw_xyzi[4 * i + 2]; p.intensity = raw_xyzi[4 * i + 3];
    }
    state_ikfom x;
    x.pos = vect3(Eigen::Vector3d(s26[0], s26[1], s26[2]));
    x.rot.coeffs() = Eigen::Vector4d(s26[3], s26[4], s26[5], s26[6]);
    x.offset_R_L_I.coeffs() = Eigen::Vector4d(s26[7], s26[8], s26[9], s26[10]);
    x.offset_T_L_I = vect3(Eigen::Vector3d(s26[11], s26[12], s26[13]));
    x.vel = vect3(Eigen::Vector3d(s26[14], s26[15], s26[16]));
    x.bg = vect3(Eigen::Vector3d(s26[17], s26[18], s26[19]));
    x.ba = vect3(Eigen::Vector3d(s26[20], s26[21], s26[22]));
    x.grav.vec = Eigen::Vector3d(s26[23], s26[24], s26[25]);
    kf.change_x(x);
    esekfom::esekf<state_ikfom, 12, input_ikfom>::cov P;
    for (int r = 0; r < 23; r++) for (int c = 0; c < 23; c++) P(r, c) = P529[23 * r + c];
    kf.change_P(P);
    state_point = kf.get_x();
    flg_EKF_inited = true;
    downSizeFilterSurf.setInputCloud(feats_undistort);
    downSizeFilterSurf.filter(*feats_down_body);
    feats_down_size = feats_down_body->points.size();
    if (feats_down_size < 5) return 2;
    normvec->resize(feats_down_size);
    feats_down_world->resize(feats_down_size);
    Nearest_Points.resize(feats_down_size);
    double solve_H_time = 0;
    kf.update_iterated_dyn_share_modified(LASER_POINT_COV, solve_H_time);
    state_point = kf.get_x();
    state_to26(state_point, s26_out);
    if (P529_out) { auto Q = kf.get_P(); for (int r = 0; r < 23; r++) for (int c = 0; c < 23; c++) P529_out[23 * r + c] = Q(r, c); }
    return 3;
}
void ref_fl_reset_cache() {  // fastlio_init's reset of Nearest_Points / point_selected_surf (laserMapping.cpp:1045, 1089)
    Nearest_Points.clear();
    memset(point_selected_surf, true, sizeof(point_selected_surf));
}
int ref_fl_num_calls() { return (int)g_calls.size(); }
void ref_fl_clear_calls() { g_calls.clear(); }
int ref_fl_call(int i, double* s26, int* flags4, double* total_residual, double* HtH36, double* Hth6) {
    if (i < 0 || i >= (int)g_calls.size()) return -1;
    const HCall& c = g_calls[i];
    std::copy(c.s26, c.s26 + 26, s26);
    flags4[0] = c.converge; flags4[1] = c.valid; flags4[2] = c.n_eff; flags4[3] = c.degenerate;
    *total_residual = c.total_residual;
    std::copy(c.HtH, c.HtH + 36, HtH36);
    std::copy(c.Hth, c.Hth + 6, Hth6);
    return 0;
}

int ref_fl_init(const double* extT3, const double* extR9, int filter_num, int max_point_num, double scan_period, int undistort) {
    std::vector<double> t(extT3, extT3 + 3), r(extR9, extR9 + 9);
    const int rc = fastlio_init(t, r, filter_num, max_point_num, scan_period, undistort != 0);
    double epsi[23];
    std::fill(epsi, epsi + 23, 0.001);
    // laserMapping.cpp:1110-1115 again with the logged measurement model.  A fresh filter first: init_dyn_share appends to the
    // state's S2 / SO3 index lists, calling it twice on one object would visit every manifold block twice.
    kf = esekfom::esekf<state_ikfom, 12, input_ikfom>();
    kf.init_dyn_share(get_f, df_dx, df_dw, h_share_logged, NUM_MAX_ITERATIONS, epsi);
    g_calls.clear();
    return rc;
}
int ref_fl_is_init() { return fastlio_is_init() ? 1 : 0; }
void ref_fl_imu_enqueue(double stamp, const double* acc, const double* gyr) {
    ImuType imu;
    imu.stamp = stamp;
    imu.acc = Eigen::Vector3d(acc[0], acc[1], acc[2]);
    imu.gyr = Eigen::Vector3d(gyr[0], gyr[1], gyr[2]);
    fastlio_imu_enqueue(imu);
}
void ref_fl_ins_enqueue(int rtk_valid, uint64_t timestamp_us, double heading, double pitch, double roll, double Ve, double Vn, double Vu, const char* sensor) {
    RTKType ins;
    ins.timestamp = timestamp_us; ins.heading = heading; ins.pitch = pitch; ins.roll = roll;
    ins.Ve = Ve; ins.Vn = Vn; ins.Vu = Vu; ins.sensor = sensor ? sensor : "";
    fastlio_ins_enqueue(rtk_valid != 0, ins);
}
void ref_fl_pcl_enqueue(const float* xyzi, const uint32_t* stamp_us, int n, uint64_t header_stamp_us) {
    PointCloudAttrPtr f(new PointCloudAttr());
    f->cloud->points.resize(n);
    f->attr.resize(n);
    for (int i = 0; i < n; i++) {
        Point& p = f->cloud->points[i];
        p.x = xyzi[4 * i]; p.y = xyzi[4 * i + 1]; p.z = xyzi[4 * i + 2]; p.intensity = xyzi[4 * i + 3];
        f->attr[i].id = 0;
        f->attr[i].stamp = stamp_us[i];
    }
    f->cloud->width = n; f->cloud->height = 1;
    f->cloud->header.stamp = header_stamp_us;
    f->T = Eigen::Matrix4d::Identity();
    fastlio_pcl_enqueue(f);
}
int ref_fl_main() { return fastlio_main() ? 1 : 0; }
void ref_fl_odometry(double* odom_s16, double* odom_e16) {  // row-major
    Eigen::Matrix4d a, b;
    fastlio_odometry(a, b);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { odom_s16[4 * r + c] = a(r, c); odom_e16[4 * r + c] = b(r, c); }
}
void ref_fl_state(double* s26, double* start26, double* P529) {
    state_to26(state_point, s26);
    state_to26(p_imu->start_state_point, start26);
    if (P529) { auto P = kf.get_P(); for (int r = 0; r < 23; r++) for (int c = 0; c < 23; c++) P529[23 * r + c] = P(r, c); }
}
int ref_fl_fastlio_state(double* out20) { auto v = fastlio_state(); for (size_t i = 0; i < v.size() && i < 20; i++) out20[i] = v[i]; return (int)v.size(); }
int ref_fl_undistorted(float* out8, int cap) { return cloud_out(feats_undistort, out8, cap, 1); }
int ref_fl_down_body(float* out4, int cap) { return cloud_out(feats_down_body, out4, cap, 0); }
int ref_fl_down_world(float* out4, int cap) { return cloud_out(feats_down_world, out4, cap, 0); }
int ref_fl_last_preprocessed(float* out8, int cap) { return lidar_buffer.empty() ? 0 : cloud_out(lidar_buffer.back(), out8, cap, 1); }
int ref_fl_map_voxels() { return ivox ? (int)ivox->NumValidGrids() : 0; }
void ref_fl_info(double* out) {  // effct_feat_num, feats_down_size, is_degenerate, travel_distance, flg_EKF_inited, nearby type
    out[0] = effct_feat_num; out[1] = feats_down_size; out[2] = is_degenerate ? 1 : 0; out[3] = travel_distance; out[4] = flg_EKF_inited ? 1 : 0;
    out[5] = ivox ? (double)(int)ivox->GetNearByType() : -1; out[6] = lidar_buffer.size(); out[7] = imu_buffer.size();
}
}
