// oracle/ref_ukf.cpp -- the reference's OWN unscented Kalman filter and pose system
// (slam/localization/hdl_localization/include/kkl/alg/unscented_kalman_filter.hpp, include/hdl_localization/pose_system.hpp;
// Eigen only) compiled from where they lie under /root/reference, behind a C ABI.  The three lines PoseEstimator::predict
// wraps around ukf->predict (setProcessNoiseCov(process_noise * dt); system.dt = dt; pose_estimator.cpp:153-156,179-185) and
// the constructor's noise / covariance setup (:31-65) are restated here because pose_estimator.cpp itself needs PCL.
// TEST INFRASTRUCTURE ONLY: built into oracle/_ref/libref_ukf.so by `make -C oracle ref`; tools/make_golden.py records its
// outputs in tests/golden/ukf.npz for the GPU box.
#include <hdl_localization/pose_system.hpp>

#include <memory>

using Ukf = kkl::alg::UnscentedKalmanFilterX<float, hdl_localization::PoseSystem>;

struct RefUkf {
    std::unique_ptr<Ukf> ukf;
    Eigen::MatrixXf process_noise;
};

extern "C" {

void* ref_ukf_create(const float* imu_ext16, const float* pos3, const float* quat_wxyz) {
    RefUkf* r = new RefUkf();
    Eigen::MatrixXf process_noise = Eigen::MatrixXf::Identity(23, 23);
    process_noise.middleRows(0, 3) *= 2.0;
    process_noise.middleRows(3, 3) *= 5.0;
    process_noise.middleRows(6, 4) *= 2.0;
    process_noise.middleRows(10, 3) *= 1e-4;
    process_noise.middleRows(13, 3) *= 1e-4;
    process_noise.middleRows(16, 3) *= 5.0;
    process_noise.middleRows(19, 4) *= 1e-4;
    Eigen::MatrixXf measurement_noise = Eigen::MatrixXf::Identity(7, 7);
    measurement_noise.middleRows(0, 3) *= 0.2;
    measurement_noise.middleRows(3, 4) *= 0.1;
    Eigen::VectorXf mean(23);
    mean.setZero();
    mean.middleRows(0, 3) = Eigen::Vector3f(pos3[0], pos3[1], pos3[2]);
    mean.middleRows(6, 4) = Eigen::Vector4f(quat_wxyz[0], quat_wxyz[1], quat_wxyz[2], quat_wxyz[3]);
    mean.middleRows(19, 4) = Eigen::Vector4f(1, 0, 0, 0);
    Eigen::MatrixXf cov = Eigen::MatrixXf::Identity(23, 23) * 0.1;
    cov.middleRows(19, 4) *= 1e-2;
    hdl_localization::PoseSystem system;
    r->process_noise = process_noise;
    r->ukf.reset(new Ukf(system, 23, 6, 7, process_noise, measurement_noise, mean, cov));
    Eigen::Matrix4f ext;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) ext(i, j) = imu_ext16[i * 4 + j];
    r->ukf->system.imu_extrinic = Eigen::Quaternionf(ext.topLeftCorner<3, 3>()).normalized();
    return r;
}
void ref_ukf_destroy(void* h) { delete static_cast<RefUkf*>(h); }

void ref_ukf_predict(void* h, double dt, const float* control6) {
    RefUkf* r = static_cast<RefUkf*>(h);
    r->ukf->setProcessNoiseCov(r->process_noise * dt);
    r->ukf->system.dt = dt;
    if (control6) {
        Eigen::VectorXf control(6);
        for (int i = 0; i < 6; i++) control[i] = control6[i];
        r->ukf->predict(control);
    } else {
        r->ukf->predict();
    }
}
void ref_ukf_correct(void* h, const float* obs7) {
    RefUkf* r = static_cast<RefUkf*>(h);
    Eigen::VectorXf z(7);
    for (int i = 0; i < 7; i++) z[i] = obs7[i];
    r->ukf->correct(z);
}
void ref_ukf_get(void* h, float* mean23, float* cov529) {
    RefUkf* r = static_cast<RefUkf*>(h);
    for (int i = 0; i < 23; i++) mean23[i] = r->ukf->mean[i];
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) cov529[i * 23 + j] = r->ukf->cov(i, j);
}

}  // extern "C"
