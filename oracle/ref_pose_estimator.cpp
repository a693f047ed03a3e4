// oracle/ref_pose_estimator.cpp -- the reference's OWN localisation filter class, hdl_localization::PoseEstimator
// (slam/localization/hdl_localization/src/pose_estimator.cpp over kkl/alg/unscented_kalman_filter.hpp and pose_system.hpp), compiled
// whole from where it lies: constructor, predict (with / without IMU), predict_nostate, get_timed_pose + the INS state queue,
// match with and without a GNSS observation (fusion_pose), the GNSS-only match, correct.  The scan matcher behind `registration` is a
// mock that returns a prescribed pose / convergence flag / fitness score: what is pinned here is the filter logic AROUND the matcher
// (the matcher itself: tests/test_ndt_vs_ref_cuda.py).  Shimmed: PCL containers + Registration base, boost::optional, logging, the plain
// structs of mapping_types.h.  TEST INFRASTRUCTURE ONLY: oracle/_ref/libref_pose_estimator.so by `make -C oracle ref`.
#define __MAPPING_TYPES_H
#include "ref_shims/mapping_types.h"
#include <unistd.h>
#define usleep(x) ((void)0)  // the GNSS-only match sleeps 100 ms when it has nothing to fuse (pose_estimator.cpp:310)
#include <mutex>
#include <memory>
#include <vector>
#define private public  // the filter state (ukf->mean / cov, state_queue) is private; the harness reads it, nothing else
#include <hdl_localization/pose_estimator.hpp>
#undef private
#include <localization/hdl_localization/src/pose_estimator.cpp>  // -I$(REF)/slam
#undef usleep

using hdl_localization::PoseEstimator;
typedef pcl::PointXYZI PointT;

struct MockRegistration : public pcl::Registration<PointT, PointT> {
    Eigen::Matrix4f result = Eigen::Matrix4f::Identity(), last_guess = Eigen::Matrix4f::Identity();
    bool conv = true;
    double fitness = 0.0;
    double getFitnessScore(double) override { return fitness; }

   protected:
    void computeTransformation(PointCloudSource&, const Matrix4& guess) override {
        last_guess = guess;
        final_transformation_ = result;
        converged_ = conv;
    }
};

struct Handle {
    std::unique_ptr<PoseEstimator> pe;
    std::shared_ptr<MockRegistration> reg;
};

static boost::optional<std::shared_ptr<RTKType>> make_gps(const double* T16, double precision, int dimension) {
    boost::optional<std::shared_ptr<RTKType>> g;
    if (!T16) return g;
    std::shared_ptr<RTKType> p(new RTKType());
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) p->T(r, c) = T16[4 * r + c];
    p->precision = precision;
    p->dimension = dimension;
    g = p;
    return g;
}

extern "C" {
void* ref_pe_create(const float* imu_ext16, uint64_t stamp, const float* pos3, const float* quat_wxyz, double cool_time) {
    Eigen::Matrix4f E;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) E(r, c) = imu_ext16[4 * r + c];
    Handle* h = new Handle();
    h->pe.reset(new PoseEstimator(E, stamp, Eigen::Vector3f(pos3[0], pos3[1], pos3[2]), Eigen::Quaternionf(quat_wxyz[0], quat_wxyz[1], quat_wxyz[2], quat_wxyz[3]), cool_time));
    h->reg.reset(new MockRegistration());
    pcl::Registration<PointT, PointT>::Ptr r = h->reg;
    h->pe->set_registration(r);
    return h;
}
void ref_pe_destroy(void* h) { delete static_cast<Handle*>(h); }
void ref_pe_predict(void* h, uint64_t stamp, const float* acc, const float* gyro) {
    PoseEstimator& p = *static_cast<Handle*>(h)->pe;
    if (acc) p.predict(stamp, Eigen::Vector3f(acc[0], acc[1], acc[2]), Eigen::Vector3f(gyro[0], gyro[1], gyro[2]));
    else p.predict(stamp);
}
void ref_pe_predict_nostate(void* h, uint64_t stamp, double* T16) {
    const Eigen::Matrix4d T = static_cast<Handle*>(h)->pe->predict_nostate(stamp);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T16[4 * r + c] = T(r, c);
}
// RTKType fields as the nodelet fills them: acc in g, gyro in deg/s
int ref_pe_get_timed_pose(void* h, uint64_t stamp, const double* acc_g, const double* gyro_dps, double* T16) {
    RTKType ins;
    ins.timestamp = stamp;
    ins.acc_x = acc_g[0]; ins.acc_y = acc_g[1]; ins.acc_z = acc_g[2];
    ins.gyro_x = gyro_dps[0]; ins.gyro_y = gyro_dps[1]; ins.gyro_z = gyro_dps[2];
    Eigen::Matrix4d T = Eigen::Matrix4d::Identity();
    const bool ok = static_cast<Handle*>(h)->pe->get_timed_pose(ins, T);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T16[4 * r + c] = T(r, c);
    return ok ? 1 : 0;
}
// match(observation, observation_cov, stamp, cloud, gps_observation, fitness_score) with the mock matcher returning `aligned16`
int ref_pe_match(void* hh, uint64_t stamp, const float* aligned16, int converged, double fitness_in, const double* gps_T16, double gps_precision,
                 int gps_dimension, float* observation7, float* observation_cov49, float* guess16, double* fitness_out) {
    Handle* h = static_cast<Handle*>(hh);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) h->reg->result(r, c) = aligned16[4 * r + c];
    h->reg->conv = converged != 0;
    h->reg->fitness = fitness_in;
    Eigen::VectorXf obs(7);
    Eigen::MatrixXf cov;
    pcl::PointCloud<PointT>::Ptr cloud(new pcl::PointCloud<PointT>());
    auto gps = make_gps(gps_T16, gps_precision, gps_dimension);
    double fit = -1.0;
    const bool ok = h->pe->match(obs, cov, stamp, cloud, gps, fit);
    for (int i = 0; i < 7; i++) observation7[i] = obs[i];
    for (int r = 0; r < 7; r++) for (int c = 0; c < 7; c++) observation_cov49[7 * r + c] = cov(r, c);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) guess16[4 * r + c] = h->reg->last_guess(r, c);
    *fitness_out = fit;
    return ok ? 1 : 0;
}
int ref_pe_match_gps_only(void* hh, const double* gps_T16, double gps_precision, int gps_dimension, float* observation7, float* observation_cov49) {
    Handle* h = static_cast<Handle*>(hh);
    Eigen::VectorXf obs(7);
    Eigen::MatrixXf cov = Eigen::MatrixXf::Zero(7, 7);
    auto gps = make_gps(gps_T16, gps_precision, gps_dimension);
    const bool ok = h->pe->match(obs, cov, gps);
    for (int i = 0; i < 7; i++) observation7[i] = obs[i];
    for (int r = 0; r < 7; r++) for (int c = 0; c < 7; c++) observation_cov49[7 * r + c] = cov(r, c);
    return ok ? 1 : 0;
}
void ref_pe_correct(void* h, uint64_t stamp, const float* observation7) {
    Eigen::VectorXf obs(7);
    for (int i = 0; i < 7; i++) obs[i] = observation7[i];
    Eigen::MatrixXf cov = Eigen::MatrixXf::Identity(7, 7);
    static_cast<Handle*>(h)->pe->correct(stamp, obs, cov);
}
void ref_pe_matrix(void* h, float* T16) {
    const Eigen::Matrix4f T = static_cast<Handle*>(h)->pe->matrix();
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T16[4 * r + c] = T(r, c);
}
void ref_pe_state(void* h, float* mean23, float* cov529) {
    PoseEstimator& p = *static_cast<Handle*>(h)->pe;
    for (int i = 0; i < 23; i++) mean23[i] = p.ukf->mean[i];
    if (cov529) for (int r = 0; r < 23; r++) for (int c = 0; c < 23; c++) cov529[23 * r + c] = p.ukf->cov(r, c);
}
int ref_pe_queue(void* h, uint64_t* stamps, float* means23, int cap) {
    PoseEstimator& p = *static_cast<Handle*>(h)->pe;
    const int n = (int)p.state_queue.size();
    for (int i = 0; i < n && i < cap; i++) {
        stamps[i] = p.state_queue[i].timestamp;
        for (int k = 0; k < 23; k++) means23[23 * i + k] = p.state_queue[i].mean[k];
    }
    return n;
}
uint64_t ref_pe_get_dt(void* h) { return static_cast<Handle*>(h)->pe->get_dt(); }
uint64_t ref_pe_last_correction_time(void* h) { return static_cast<Handle*>(h)->pe->last_correction_time(); }
}
