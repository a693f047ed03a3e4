// oracle/ref_harness.cpp -- compiles the reference's OWN code for the two numerically delicate pieces of the
// path, from where it lies under /root/reference (nothing is copied into this repo), behind a C ABI:
//   * esti_plane<float>       slam/mapping/fastlio/include/common_lib.h:236-268   (Eigen colPivHouseholderQr)
//   * faster_lio::IVox        slam/mapping/fastlio/include/ivox3d/ivox3d.h, ivox3d_node.hpp, eigen_types.h
//   * calc_dist               common_lib.h:231-234
// Eigen is the copy vendored in the reference tree (slam/thirdparty/fast_gicp/thirdparty/Eigen).  PCL is not
// installed: oracle/ref_shims/pcl/* define the three point structs and PointCloud the headers name, and
// common/mapping_types.h (OpenCV, queues -- unrelated to this path) is replaced by oracle/ref_shims/mapping_types.h.
// The IKFoM filter (needs Boost.PP / Boost.Bind) and PCL's VoxelGrid are NOT covered: see DESIGN.md.
// TEST INFRASTRUCTURE ONLY: built into oracle/_ref/libref_harness.so, used by tests/ and tools/make_golden.py.
#include <deque>
#include <string>
#include <common_lib.h>
#include <ivox3d/ivox3d.h>
#include <Eigen/Eigenvalues>
#include <fast_gicp/so3/so3.hpp>

#include <cstring>

using IVoxT = faster_lio::IVox<3, faster_lio::IVoxNodeType::DEFAULT, PointType>;

static PointType mk(const float* p) {
    PointType q;
    q.x = p[0]; q.y = p[1]; q.z = p[2]; q.intensity = p[3];
    return q;
}

extern "C" {

int ref_esti_plane(const float* five_xyzi, float threshold, float* pabcd) {
    PointVector pts;
    for (int j = 0; j < 5; j++) pts.push_back(mk(five_xyzi + 4 * j));
    Matrix<float, 4, 1> r;
    const bool ok = esti_plane<float>(r, pts, threshold);
    for (int i = 0; i < 4; i++) pabcd[i] = r(i);
    return ok ? 1 : 0;
}

float ref_calc_dist(const float* a, const float* b) { return calc_dist(mk(a), mk(b)); }

void* ref_ivox_create(float res, int stencil, uint64_t capacity, double max_distance) {
    IVoxT::Options o;
    o.resolution_ = res;
    o.capacity_ = (size_t)capacity;
    o.max_distance_ = max_distance;
    o.nearby_type_ = stencil == 1 ? IVoxT::NearbyType::CENTER : stencil == 7 ? IVoxT::NearbyType::NEARBY6 : stencil == 19 ? IVoxT::NearbyType::NEARBY18
                     : stencil == 27 ? IVoxT::NearbyType::NEARBY26 : IVoxT::NearbyType::NEARBY74;
    return new IVoxT(o);
}
void ref_ivox_destroy(void* h) { delete static_cast<IVoxT*>(h); }
void ref_ivox_set_stencil(void* h, int stencil) {
    static_cast<IVoxT*>(h)->SetNearByType(stencil == 1 ? IVoxT::NearbyType::CENTER : stencil == 7 ? IVoxT::NearbyType::NEARBY6
                                          : stencil == 19 ? IVoxT::NearbyType::NEARBY18 : stencil == 27 ? IVoxT::NearbyType::NEARBY26
                                                                                                        : IVoxT::NearbyType::NEARBY74);
}
void ref_ivox_add(void* h, const float* pts, int n, double travel) {
    PointVector v;
    v.reserve(n);
    for (int i = 0; i < n; i++) v.push_back(mk(pts + 4 * (size_t)i));
    static_cast<IVoxT*>(h)->AddPoints(v, travel);
}
uint64_t ref_ivox_num_voxels(void* h) { return static_cast<IVoxT*>(h)->NumValidGrids(); }
// GetClosestPoint(pt, out, 5, 5.0) per query; out_pts n x 5 x 4, out_cnt n.  `out` starts empty for every query.
void ref_ivox_knn(void* h, const float* q, int n, float* out_pts, int* out_cnt) {
    IVoxT* iv = static_cast<IVoxT*>(h);
    for (int i = 0; i < n; i++) {
        PointVector near;
        iv->GetClosestPoint(mk(q + 4 * (size_t)i), near, 5, 5.0);
        out_cnt[i] = (int)near.size();
        for (int k = 0; k < 5; k++) {
            float* o = out_pts + ((size_t)i * 5 + k) * 4;
            if (k < (int)near.size()) { o[0] = near[k].x; o[1] = near[k].y; o[2] = near[k].z; o[3] = near[k].intensity; }
            else { o[0] = o[1] = o[2] = o[3] = 0.f; }
        }
    }
}

// ---- localization matcher pieces that are plain Eigen / header-only reference code ---------------------------
// fast_gicp::se3_exp (slam/thirdparty/fast_gicp/include/fast_gicp/so3/so3.hpp:80-105), the reference's own code
void ref_se3_exp(const double* a6, double* T16) {
    Eigen::Matrix<double, 6, 1> a;
    for (int i = 0; i < 6; i++) a[i] = a6[i];
    const Eigen::Isometry3d T = fast_gicp::se3_exp(a);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T16[i * 4 + j] = T.matrix()(i, j);
}
// the Eigen calls of covariance_regularization.cu:15-52,105-116 (PLANE) and ndt_compute_derivatives.cu:69: the .cu
// files cannot be compiled here, but their arithmetic is these three Eigen expressions on Matrix3f
void ref_eig3_direct(const float* cov9, float* w3, float* V9) {
    Eigen::Matrix3f C;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C(i, j) = cov9[i * 3 + j];
    Eigen::SelfAdjointEigenSolver<Eigen::Matrix3f> eig;
    eig.computeDirect(C);
    for (int i = 0; i < 3; i++) { w3[i] = eig.eigenvalues()[i]; for (int j = 0; j < 3; j++) V9[i * 3 + j] = eig.eigenvectors()(i, j); }
}
void ref_regularize_plane(const float* cov9, float* out9, float* inv9) {
    Eigen::Matrix3f C;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C(i, j) = cov9[i * 3 + j];
    Eigen::SelfAdjointEigenSolver<Eigen::Matrix3f> eig;
    eig.computeDirect(C);
    const Eigen::Matrix3f vecs = eig.eigenvectors();
    const Eigen::Matrix3f vecs_inv = vecs.inverse();
    const Eigen::Matrix3f values_diag = Eigen::Vector3f(1e-3f, 1.0f, 1.0f).asDiagonal();
    const Eigen::Matrix3f R = (vecs * values_diag * vecs_inv).eval();
    const Eigen::Matrix3f Ri = R.inverse();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { out9[i * 3 + j] = R(i, j); inv9[i * 3 + j] = Ri(i, j); }
}

// so3_math.h Exp(ang_vel, dt) -- the reference's own template -- and the per-point compensation of
// ImuProcess::UndistortPcl (IMU_Processing.hpp:386-394).  UndistortPcl itself needs the IKFoM state (boost
// preprocessor), so the one expression is restated here over the same Eigen types (MTK::SO3 derives from
// Eigen::Quaternion, MTK::vect from Eigen::Matrix): what is pinned is Eigen's operation order for it.
void ref_so3_Exp(const double* w3, double dt, double* R9) {
    Eigen::Vector3d w(w3[0], w3[1], w3[2]);
    Eigen::Matrix3d R = Exp(w, dt);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R9[i * 3 + j] = R(i, j);
}
void ref_undistort_point(const double* R_imu9, const double* vel3, const double* pos3, const double* acc3, const double* gyr3, double dt,
                         const float* p_xyz, const double* end_pos3, const double* end_rot_xyzw, const double* ril_xyzw, const double* til3, float* out_xyz) {
    Eigen::Matrix3d R_imu;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R_imu(i, j) = R_imu9[i * 3 + j];
    Eigen::Vector3d vel_imu(vel3[0], vel3[1], vel3[2]), pos_imu(pos3[0], pos3[1], pos3[2]), acc_imu(acc3[0], acc3[1], acc3[2]);
    Eigen::Vector3d angvel_avr(gyr3[0], gyr3[1], gyr3[2]), end_pos(end_pos3[0], end_pos3[1], end_pos3[2]), til(til3[0], til3[1], til3[2]);
    Eigen::Quaterniond rot(end_rot_xyzw[3], end_rot_xyzw[0], end_rot_xyzw[1], end_rot_xyzw[2]);
    Eigen::Quaterniond ril(ril_xyzw[3], ril_xyzw[0], ril_xyzw[1], ril_xyzw[2]);
    Eigen::Matrix3d R_i(R_imu * Exp(angvel_avr, dt));
    Eigen::Vector3d P_i(p_xyz[0], p_xyz[1], p_xyz[2]);
    Eigen::Vector3d T_ei(pos_imu + vel_imu * dt + 0.5 * acc_imu * dt * dt - end_pos);
    Eigen::Vector3d P_compensate = ril.conjugate() * (rot.conjugate() * (R_i * (ril * P_i + til) + T_ei) - til);
    out_xyz[0] = P_compensate(0);
    out_xyz[1] = P_compensate(1);
    out_xyz[2] = P_compensate(2);
}

}  // extern "C"
