// pose_estimator.cpp -- the localisation loop around the NDT matcher: hdl_localization::PoseEstimator
// (/root/reference/slam/localization/hdl_localization/src/pose_estimator.cpp) = an unscented Kalman filter
// (include/kkl/alg/unscented_kalman_filter.hpp:42-262) over the 23-number pose system of include/hdl_localization/
// pose_system.hpp:14-115 -- predict with or without an IMU sample, scan matching from the predicted pose on the device
// (lio_ndt_align), the too-large-transform gate, quaternion sign continuity, 7-number observation, correct.
// Host C++, f32 like the reference (Eigen::MatrixXf there; plain loops here).  Includes the GNSS fusion (fusion_pose), the
// GNSS-only match, get_timed_pose with the INS state queue and predict_nostate; the fitness score of the warm-up phase is
// lio_ndt_fitness_score (ndt.hip).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/lio_hip.h"

namespace {

constexpr int N = 23, K = 7, NK = N + K;

struct Quat { float w, x, y, z; };
inline Quat qnormalized(Quat q) {
    const float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);  // Eigen: coeffs / norm (norm = sqrt(squaredNorm))
    if (n > 0.f) { q.w /= n; q.x /= n; q.y /= n; q.z /= n; }
    return q;
}
inline Quat qmul(const Quat& a, const Quat& b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Quat qconj(const Quat& q) { return {q.w, -q.x, -q.y, -q.z}; }
inline Quat qinverse(const Quat& q) {  // Eigen QuaternionBase::inverse: conjugate / squaredNorm
    const float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (!(n2 > 0.f)) return {0, 0, 0, 0};
    return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
inline void qrot(const Quat& q, const float v[3], float o[3]) {  // Eigen _transformVector
    const float ux = 2.f * (q.y * v[2] - q.z * v[1]), uy = 2.f * (q.z * v[0] - q.x * v[2]), uz = 2.f * (q.x * v[1] - q.y * v[0]);
    o[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
    o[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
    o[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
inline void qtoR(const Quat& q, float R[9]) {
    const float tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z, twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x,
                txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline Quat qfromR(const float m[9]) {  // Eigen's rotation matrix -> quaternion
    Quat q;
    float t = m[0] + m[4] + m[8];
    if (t > 0.f) {
        t = sqrtf(t + 1.0f);
        q.w = 0.5f * t;
        t = 0.5f / t;
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf(m[i * 4] - m[j * 4] - m[k * 4] + 1.0f);
        float v[3];
        v[i] = 0.5f * t;
        t = 0.5f / t;
        q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
        v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

// pose_system.hpp: state = [p3, v3, q(wxyz)4, acc_bias3, gyro_bias3, gyro3, dq_ext(wxyz)4]
struct PoseSystem {
    double dt = 0.01;
    Quat imu_ext{1, 0, 0, 0};
    void f(const float* s, const float* control, float* o) const {
        const float dtf = (float)dt;  // Eigen: float vector * double scalar -> the scalar is cast to the vector's scalar type
        Quat qt = qnormalized({s[6], s[7], s[8], s[9]});
        Quat dq_ext = qnormalized({s[19], s[20], s[21], s[22]});
        for (int i = 0; i < 3; i++) o[i] = s[i] + s[3 + i] * dtf;
        float gyro[3];
        if (control) {
            const float g[3] = {0.0f, 0.0f, 9.81f};
            const float amb[3] = {control[0] - s[10], control[1] - s[11], control[2] - s[12]};
            float a1[3], a2[3], acc[3];
            qrot(qinverse(imu_ext), amb, a1);
            qrot(dq_ext, a1, a2);
            qrot(qt, a2, acc);
            for (int i = 0; i < 3; i++) o[3 + i] = s[3 + i] + (acc[i] - g[i]) * dtf;
            const float gmb[3] = {control[3] - s[13], control[4] - s[14], control[5] - s[15]};
            float g1[3];
            qrot(qinverse(imu_ext), gmb, g1);
            qrot(dq_ext, g1, gyro);
        } else {
            for (int i = 0; i < 3; i++) { o[3 + i] = s[3 + i]; gyro[i] = s[16 + i]; }
        }
        // gyro[i] * dt / 2 is plain C++ there: float * double, evaluated in double, narrowed by the Quaternionf constructor
        Quat dq = qnormalized({1.f, (float)((double)gyro[0] * dt / 2), (float)((double)gyro[1] * dt / 2), (float)((double)gyro[2] * dt / 2)});
        Quat qn = qnormalized(qmul(qt, dq));
        o[6] = qn.w; o[7] = qn.x; o[8] = qn.y; o[9] = qn.z;
        for (int i = 10; i < 16; i++) o[i] = s[i];
        for (int i = 0; i < 3; i++) o[16 + i] = gyro[i];
        o[19] = dq_ext.w; o[20] = dq_ext.x; o[21] = dq_ext.y; o[22] = dq_ext.z;
    }
    static void h(const float* s, float* z) {
        for (int i = 0; i < 3; i++) z[i] = s[i];
        const float n = sqrtf(s[6] * s[6] + s[7] * s[7] + s[8] * s[8] + s[9] * s[9]);
        for (int i = 0; i < 4; i++) z[3 + i] = n > 0.f ? s[6 + i] / n : s[6 + i];
    }
};

// lower Cholesky factor of a symmetric matrix (Eigen::LLT, lower): L L^T = A
bool cholesky(const float* A, int n, float* L) {
    for (int i = 0; i < n * n; i++) L[i] = 0.f;
    for (int j = 0; j < n; j++) {
        float d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0.f)) return false;
        const float l = sqrtf(d);
        L[j * n + j] = l;
        for (int i = j + 1; i < n; i++) {
            float s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / l;
        }
    }
    return true;
}

bool inverse(const float* A, int n, float* out) {  // partial-pivot LU (Eigen's inverse() of a dynamic matrix)
    std::vector<float> M(A, A + n * n);
    std::vector<int> piv(n);
    for (int i = 0; i < n * n; i++) out[i] = 0.f;
    for (int i = 0; i < n; i++) out[i * n + i] = 1.f;
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++)
            if (fabsf(M[r * n + c]) > fabsf(M[p * n + c])) p = r;
        if (M[p * n + c] == 0.f) return false;
        if (p != c)
            for (int k = 0; k < n; k++) { std::swap(M[c * n + k], M[p * n + k]); std::swap(out[c * n + k], out[p * n + k]); }
        const float d = M[c * n + c];
        for (int r = c + 1; r < n; r++) {
            const float f = M[r * n + c] / d;
            if (f == 0.f) continue;
            for (int k = c; k < n; k++) M[r * n + k] -= f * M[c * n + k];
            for (int k = 0; k < n; k++) out[r * n + k] -= f * out[c * n + k];
        }
    }
    for (int c = n - 1; c >= 0; c--) {
        const float d = M[c * n + c];
        for (int k = 0; k < n; k++) out[c * n + k] /= d;
        for (int r = 0; r < c; r++) {
            const float f = M[r * n + c];
            if (f == 0.f) continue;
            for (int k = 0; k < n; k++) out[r * n + k] -= f * out[c * n + k];
        }
    }
    return true;
}

struct Ukf {
    float mean[N];
    float cov[N * N];
    float process_noise[N * N];
    float measurement_noise[K * K];
    PoseSystem system;
    float lambda = 1.f;
    float kalman_gain[NK * K];

    // sigma points of (mean, cov): row 0 = mean, rows 1 + 2i / 2 + 2i = mean +- column i of chol((n + lambda) cov)
    static bool sigma_points(const float* m, const float* c, int n, float lambda, std::vector<float>& sp) {
        std::vector<float> A((size_t)n * n), L((size_t)n * n);
        for (int i = 0; i < n * n; i++) A[i] = ((float)n + lambda) * c[i];
        if (!cholesky(A.data(), n, L.data())) return false;
        sp.assign((size_t)(2 * n + 1) * n, 0.f);
        for (int k = 0; k < n; k++) sp[k] = m[k];
        for (int i = 0; i < n; i++)
            for (int k = 0; k < n; k++) {
                sp[(size_t)(1 + 2 * i) * n + k] = m[k] + L[k * n + i];
                sp[(size_t)(2 + 2 * i) * n + k] = m[k] - L[k * n + i];
            }
        return true;
    }

    bool predict(const float* control) {  // unscented_kalman_filter.hpp:86-146
        std::vector<float> sp;
        if (!sigma_points(mean, cov, N, lambda, sp)) return false;
        const int S = 2 * N + 1;
        std::vector<float> w(S, 1.f / (2.f * ((float)N + lambda)));
        w[0] = lambda / ((float)N + lambda);
        float tmp[N];
        for (int i = 0; i < S; i++) {
            system.f(&sp[(size_t)i * N], control, tmp);
            memcpy(&sp[(size_t)i * N], tmp, sizeof(tmp));
        }
        float mp[N] = {0};
        for (int i = 0; i < S; i++)
            for (int k = 0; k < N; k++) mp[k] += w[i] * sp[(size_t)i * N + k];
        std::vector<float> cp(N * N, 0.f);
        for (int i = 0; i < S; i++) {
            float d[N];
            for (int k = 0; k < N; k++) d[k] = sp[(size_t)i * N + k] - mp[k];
            for (int a = 0; a < N; a++) {
                const float wa = w[i] * d[a];
                for (int b = 0; b < N; b++) cp[a * N + b] += wa * d[b];
            }
        }
        for (int i = 0; i < N * N; i++) cov[i] = cp[i] + process_noise[i];
        memcpy(mean, mp, sizeof(mp));
        return true;
    }

    bool correct(const float* z) {  // unscented_kalman_filter.hpp:152-197
        float em[NK] = {0};
        std::vector<float> ec((size_t)NK * NK, 0.f);
        for (int i = 0; i < N; i++) {
            em[i] = mean[i];
            for (int j = 0; j < N; j++) ec[(size_t)i * NK + j] = cov[i * N + j];
        }
        for (int i = 0; i < K; i++)
            for (int j = 0; j < K; j++) ec[(size_t)(N + i) * NK + N + j] = measurement_noise[i * K + j];
        std::vector<float> sp;
        if (!sigma_points(em, ec.data(), NK, lambda, sp)) return false;
        const int S = 2 * NK + 1;
        std::vector<float> w(S, 1.f / (2.f * ((float)NK + lambda)));
        w[0] = lambda / ((float)NK + lambda);
        std::vector<float> zs((size_t)S * K);
        for (int i = 0; i < S; i++) PoseSystem::h(&sp[(size_t)i * NK], &zs[(size_t)i * K]);
        float zm[K] = {0};
        for (int i = 0; i < S; i++)
            for (int k = 0; k < K; k++) zm[k] += w[i] * zs[(size_t)i * K + k];
        float zc[K * K] = {0};
        for (int i = 0; i < S; i++) {
            float d[K];
            for (int k = 0; k < K; k++) d[k] = zs[(size_t)i * K + k] - zm[k];
            for (int a = 0; a < K; a++)
                for (int b = 0; b < K; b++) zc[a * K + b] += w[i] * d[a] * d[b];
        }
        for (int i = 0; i < K * K; i++) zc[i] += measurement_noise[i];
        std::vector<float> sig((size_t)NK * K, 0.f);
        for (int i = 0; i < S; i++) {
            float db[K];
            for (int k = 0; k < K; k++) db[k] = zs[(size_t)i * K + k] - zm[k];
            for (int a = 0; a < NK; a++) {
                const float da = sp[(size_t)i * NK + a] - em[a];
                for (int b = 0; b < K; b++) sig[(size_t)a * K + b] += w[i] * (da * db[b]);
            }
        }
        float zi[K * K];
        if (!inverse(zc, K, zi)) return false;
        for (int a = 0; a < NK; a++)
            for (int b = 0; b < K; b++) {
                float s = 0.f;
                for (int k = 0; k < K; k++) s += sig[(size_t)a * K + k] * zi[k * K + b];
                kalman_gain[a * K + b] = s;
            }
        float innov[K];
        for (int k = 0; k < K; k++) innov[k] = z[k] - zm[k];
        // ext_cov = ext_cov_pred - K S K^T ; only the state block is kept
        float KS[NK * K];
        for (int a = 0; a < NK; a++)
            for (int b = 0; b < K; b++) {
                float s = 0.f;
                for (int k = 0; k < K; k++) s += kalman_gain[a * K + k] * zc[k * K + b];
                KS[a * K + b] = s;
            }
        for (int a = 0; a < N; a++) {
            float s = 0.f;
            for (int k = 0; k < K; k++) s += kalman_gain[a * K + k] * innov[k];
            mean[a] = em[a] + s;
            for (int b = 0; b < N; b++) {
                float t = 0.f;
                for (int k = 0; k < K; k++) t += KS[a * K + k] * kalman_gain[b * K + k];
                cov[a * N + b] = ec[(size_t)a * NK + b] - t;
            }
        }
        return true;
    }
};

}  // namespace

struct InsState {  // one entry of PoseEstimator::state_queue (an RTKType: timestamp, raw IMU sample, predicted state)
    uint64_t stamp;
    double acc_g[3], gyro_dps[3];  // as RTKType carries them: g and deg/s, doubles
    float mean[N];
};

struct lio_pose_estimator {
    Ukf ukf;
    float process_noise[N * N];
    uint64_t init_stamp = 0, prev_stamp = 0, last_correction_stamp = 0;
    double cool_time_duration = 1.0;
    int bad = 0;  // a Cholesky factorisation failed (covariance not positive definite)
    std::vector<InsState> state_queue;
    std::mutex data_mutex;  // held by predict / predict_nostate / get_timed_pose / correct, as in the reference (the INS callback thread
                            // calls get_timed_pose while the scan thread predicts and corrects)
};

namespace {

// PoseEstimator::predict_imu (pose_estimator.cpp:88-102).  `dt_smooth` is a function-local static there: one value shared by every
// estimator of the process, never reset -- kept that way.
double g_dt_smooth = 0;
void predict_imu(lio_pose_estimator* e, uint64_t pre_stamp, uint64_t next_stamp, const float* pre_state, float* next_state, const float acc[3],
                 const float gyro[3]) {
    double dt = ((double)next_stamp - (double)pre_stamp) / 1000000.0;
    dt = std::max(0.0, std::min(1.0, dt));
    g_dt_smooth = g_dt_smooth * 0.95 + dt * 0.05;
    e->ukf.system.dt = g_dt_smooth;
    const float control[6] = {acc[0], acc[1], acc[2], gyro[0], gyro[1], gyro[2]};
    e->ukf.system.f(pre_state, control, next_state);
}
// Vector3f(ins.acc_x * 9.81, ...) and Vector3f(ins.gyro_x / 180.0 * M_PI, ...): double arithmetic on the RTKType's doubles, narrowed
inline void ins_units(const double acc_g[3], const double gyro_dps[3], float acc[3], float gyr[3]) {
    for (int i = 0; i < 3; i++) {
        acc[i] = (float)(acc_g[i] * 9.81);
        gyr[i] = (float)(gyro_dps[i] / 180.0 * M_PI);
    }
}
inline void pose_of_state(const float* s, double T[16]) {  // Quaternionf(...).normalized() is discarded there: the raw quaternion is used
    const Quat q{s[6], s[7], s[8], s[9]};
    float R[9];
    qtoR(q, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = (double)R[i * 3 + j];
        T[i * 4 + 3] = (double)s[i];
        T[12 + i] = 0.0;
    }
    T[15] = 1.0;
}

// PoseEstimator::fusion_pose (pose_estimator.cpp:420-433): information-form fusion of two Gaussians over dim x dim blocks, f32
bool fusion_pose(const float* cov1, const float* cov2, const float* mean1, const float* mean2_in, int n, bool flip_quat, float* fused_cov,
                 float* fused_mean) {
    float mean2[K];
    for (int i = 0; i < n; i++) mean2[i] = mean2_in[i];
    if (flip_quat) {  // dim == 6 (the default argument): keep the two quaternions on one hemisphere
        float d = 0.f;
        for (int i = n - 4; i < n; i++) d += mean1[i] * mean2[i];
        if ((double)d < 0.0)
            for (int i = n - 4; i < n; i++) mean2[i] *= -1.0f;
    }
    float i1[K * K], i2[K * K], s[K * K];
    if (!inverse(cov1, n, i1) || !inverse(cov2, n, i2)) return false;
    for (int i = 0; i < n * n; i++) s[i] = i1[i] + i2[i];
    if (!inverse(s, n, fused_cov)) return false;
    // fused_cov * inv_cov1 * mean1 + fused_cov * inv_cov2 * mean2, evaluated left to right
    float a[K * K], b[K * K];
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) {
            float x = 0.f, y = 0.f;
            for (int k = 0; k < n; k++) { x += fused_cov[r * n + k] * i1[k * n + c]; y += fused_cov[r * n + k] * i2[k * n + c]; }
            a[r * n + c] = x;
            b[r * n + c] = y;
        }
    for (int r = 0; r < n; r++) {
        float x = 0.f, y = 0.f;
        for (int k = 0; k < n; k++) { x += a[r * n + k] * mean1[k]; y += b[r * n + k] * mean2[k]; }
        fused_mean[r] = x + y;
    }
    return true;
}

// the 7 x 7 pose block (position, quaternion) of the filter's 23-state mean / covariance (pose_estimator.cpp:203-211)
void imu_pose_block(const Ukf& u, float mean7[K], float cov7[K * K]) {
    static const int idx[K] = {0, 1, 2, 6, 7, 8, 9};
    for (int i = 0; i < K; i++) {
        mean7[i] = u.mean[idx[i]];
        for (int j = 0; j < K; j++) cov7[i * K + j] = u.cov[idx[i] * N + idx[j]];
    }
}
// gps_mean / gps_cov of a GNSS observation (pose_estimator.cpp:219-227); noise_scale = precision, or 10 * precision for the GNSS-only match
void gps_gaussian(const lio_gps_observation& g, double noise_scale, float mean7[K], float cov7[K * K]) {
    float R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = (float)g.T[i * 4 + j];
    const Quat q = qfromR(R);  // gps_quat.normalized() discards its result: not normalised
    for (int i = 0; i < 3; i++) mean7[i] = (float)g.T[i * 4 + 3];
    mean7[3] = q.w; mean7[4] = q.x; mean7[5] = q.y; mean7[6] = q.z;
    for (int i = 0; i < K * K; i++) cov7[i] = 0.f;
    for (int i = 0; i < K; i++) cov7[i * K + i] = (i < 3 ? 0.1f : 0.05f) * (float)noise_scale;  // MatrixXf * double: Eigen casts the scalar to float first
}

}  // namespace

extern "C" {

// PoseEstimator::PoseEstimator (pose_estimator.cpp:22-66)
lio_pose_estimator* lio_pose_estimator_create(const float imu_ext[16], uint64_t stamp_us, const float pos[3], const float quat_wxyz[4],
                                              double cool_time_duration) {
    if (!imu_ext || !pos || !quat_wxyz) return nullptr;
    lio_pose_estimator* e = new lio_pose_estimator();
    e->init_stamp = stamp_us;
    e->cool_time_duration = cool_time_duration;
    const float pn[N] = {2, 2, 2, 5, 5, 5, 2, 2, 2, 2, 1e-4f, 1e-4f, 1e-4f, 1e-4f, 1e-4f, 1e-4f, 5, 5, 5, 1e-4f, 1e-4f, 1e-4f, 1e-4f};
    memset(e->process_noise, 0, sizeof(e->process_noise));
    for (int i = 0; i < N; i++) e->process_noise[i * N + i] = pn[i];
    memcpy(e->ukf.process_noise, e->process_noise, sizeof(e->process_noise));
    memset(e->ukf.measurement_noise, 0, sizeof(e->ukf.measurement_noise));
    for (int i = 0; i < K; i++) e->ukf.measurement_noise[i * K + i] = i < 3 ? 0.2f : 0.1f;
    memset(e->ukf.mean, 0, sizeof(e->ukf.mean));
    for (int i = 0; i < 3; i++) e->ukf.mean[i] = pos[i];
    for (int i = 0; i < 4; i++) e->ukf.mean[6 + i] = quat_wxyz[i];
    e->ukf.mean[19] = 1.f;
    memset(e->ukf.cov, 0, sizeof(e->ukf.cov));
    for (int i = 0; i < N; i++) e->ukf.cov[i * N + i] = i >= 19 ? 0.1f * 1e-2f : 0.1f;
    float R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = imu_ext[i * 4 + j];
    e->ukf.system.imu_ext = qnormalized(qfromR(R));
    return e;
}
void lio_pose_estimator_destroy(lio_pose_estimator* e) { delete e; }

// PoseEstimator::predict(stamp) / predict(stamp, acc, gyro) (pose_estimator.cpp:142-186); acc, gyro == NULL: no IMU
int lio_pose_estimator_predict(lio_pose_estimator* e, uint64_t stamp_us, const float acc[3], const float gyro[3]) {
    if (!e || ((acc == nullptr) != (gyro == nullptr))) return LIO_E_INVALID;
    std::lock_guard<std::mutex> lock(e->data_mutex);
    if ((double)(stamp_us - e->init_stamp) / 1000000.0 < e->cool_time_duration || e->prev_stamp == 0 || e->prev_stamp == stamp_us) {
        e->prev_stamp = stamp_us;
        return 0;
    }
    const double dt = ((double)stamp_us - (double)e->prev_stamp) / 1000000.0;
    e->prev_stamp = stamp_us;
    if (dt <= 0 || dt > 1.0) return 0;
    for (int i = 0; i < N * N; i++) e->ukf.process_noise[i] = e->process_noise[i] * (float)dt;
    e->ukf.system.dt = dt;
    float control[6];
    if (acc) { for (int i = 0; i < 3; i++) { control[i] = acc[i]; control[3 + i] = gyro[i]; } }
    if (!e->ukf.predict(acc ? control : nullptr)) { e->bad = 1; return LIO_E_STATE; }
    return 1;
}

// PoseEstimator::get_dt (pose_estimator.cpp:389-391): the filter's current step, in us (the nodelet compensates the scan's motion over it)
uint64_t lio_pose_estimator_get_dt(lio_pose_estimator* e) {
    if (!e) return 0;
    std::lock_guard<std::mutex> lock(e->data_mutex);
    return (uint64_t)(e->ukf.system.dt * 1000000);
}

// PoseEstimator::correct (pose_estimator.cpp:348-382): filter update, then the INS state queue is trimmed to the states after the
// correction and re-predicted from the corrected mean
int lio_pose_estimator_correct(lio_pose_estimator* e, uint64_t stamp_us, const float observation[7]) {
    if (!e || !observation) return LIO_E_INVALID;
    std::lock_guard<std::mutex> lock(e->data_mutex);
    e->last_correction_stamp = stamp_us;
    e->prev_stamp = stamp_us;
    if (!e->ukf.correct(observation)) { e->bad = 1; return LIO_E_STATE; }
    auto& q = e->state_queue;
    size_t drop = 0;
    while (drop < q.size() && q[drop].stamp <= stamp_us) drop++;
    q.erase(q.begin(), q.begin() + drop);
    for (size_t i = 0; i < q.size(); i++) {
        float acc[3], gyr[3];
        ins_units(q[i].acc_g, q[i].gyro_dps, acc, gyr);
        if (i == 0) {
            predict_imu(e, stamp_us, q[i].stamp, e->ukf.mean, q[i].mean, acc, gyr);
        } else {
            float acc0[3], gyr0[3];
            ins_units(q[i - 1].acc_g, q[i - 1].gyro_dps, acc0, gyr0);
            for (int k = 0; k < 3; k++) { acc[k] = (acc[k] + acc0[k]) / 2.0f; gyr[k] = (gyr[k] + gyr0[k]) / 2.0f; }  // (Vector3f + Vector3f) / 2.0
            predict_imu(e, q[i - 1].stamp, q[i].stamp, q[i - 1].mean, q[i].mean, acc, gyr);
        }
    }
    return LIO_OK;
}

// PoseEstimator::predict_nostate (pose_estimator.cpp:70-86): where the filter would be at `stamp`, nothing changed but system.dt
int lio_pose_estimator_predict_nostate(lio_pose_estimator* e, uint64_t stamp_us, double pose[16]) {
    if (!e || !pose) return LIO_E_INVALID;
    std::lock_guard<std::mutex> lock(e->data_mutex);
    const double dt = ((double)stamp_us - (double)e->prev_stamp) / 1000000.0;
    if (dt <= 0) {
        float T[16];
        lio_pose_estimator_matrix(e, T);
        for (int i = 0; i < 16; i++) pose[i] = (double)T[i];
        return 0;
    }
    e->ukf.system.dt = dt;
    float ns[N];
    e->ukf.system.f(e->ukf.mean, nullptr, ns);
    pose_of_state(ns, pose);
    return 1;
}

// PoseEstimator::get_timed_pose (pose_estimator.cpp:104-141): one INS / IMU sample (acceleration in g, rate in deg/s, as RTKType
// carries them) extends the queue of states predicted past the last correction; returns 1 and the pose at that sample, 0 when it is
// not newer than the filter / the queue
int lio_pose_estimator_get_timed_pose(lio_pose_estimator* e, uint64_t stamp_us, const double acc_g[3], const double gyro_dps[3], double pose[16]) {
    if (!e || !acc_g || !gyro_dps || !pose) return LIO_E_INVALID;
    std::lock_guard<std::mutex> lock(e->data_mutex);
    if (stamp_us <= e->prev_stamp) return 0;
    InsState s;
    s.stamp = stamp_us;
    float acc[3], gyr[3];
    for (int i = 0; i < 3; i++) { s.acc_g[i] = acc_g[i]; s.gyro_dps[i] = gyro_dps[i]; }
    ins_units(s.acc_g, s.gyro_dps, acc, gyr);
    if (e->state_queue.empty()) {
        predict_imu(e, e->prev_stamp, stamp_us, e->ukf.mean, s.mean, acc, gyr);
    } else {
        const InsState& last = e->state_queue.back();
        if (stamp_us <= last.stamp) return 0;
        float acc0[3], gyr0[3];
        ins_units(last.acc_g, last.gyro_dps, acc0, gyr0);
        for (int i = 0; i < 3; i++) { acc[i] = (acc[i] + acc0[i]) / 2.0f; gyr[i] = (gyr[i] + gyr0[i]) / 2.0f; }  // (Vector3f + Vector3f) / 2.0
        predict_imu(e, last.stamp, stamp_us, last.mean, s.mean, acc, gyr);
    }
    e->state_queue.push_back(s);
    pose_of_state(s.mean, pose);
    return 1;
}

int lio_pose_estimator_get(lio_pose_estimator* e, float mean23[23], float cov529[529]) {
    if (!e) return LIO_E_INVALID;
    if (mean23) memcpy(mean23, e->ukf.mean, sizeof(e->ukf.mean));
    if (cov529) memcpy(cov529, e->ukf.cov, sizeof(e->ukf.cov));
    return LIO_OK;
}
int lio_pose_estimator_set(lio_pose_estimator* e, const float mean23[23], const float cov529[529]) {
    if (!e) return LIO_E_INVALID;
    if (mean23) memcpy(e->ukf.mean, mean23, sizeof(e->ukf.mean));
    if (cov529) memcpy(e->ukf.cov, cov529, sizeof(e->ukf.cov));
    return LIO_OK;
}

// PoseEstimator::matrix(): quat().normalized().toRotationMatrix() + pos(), row-major 4 x 4
int lio_pose_estimator_matrix(lio_pose_estimator* e, float T[16]) {
    if (!e || !T) return LIO_E_INVALID;
    const Quat q = qnormalized({e->ukf.mean[6], e->ukf.mean[7], e->ukf.mean[8], e->ukf.mean[9]});
    float R[9];
    qtoR(q, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = e->ukf.mean[i];
        T[12 + i] = 0.f;
    }
    T[15] = 1.f;
    return LIO_OK;
}

// ---- PoseEstimator::match (pose_estimator.cpp:191-302) in two host halves around the alignment -------------------------------
// guess: the pose the matcher starts from = the filter's pose, fused with the GNSS observation when there is one (:196-247)
int lio_pose_estimator_guess(lio_pose_estimator* e, const lio_gps_observation* gps, float init_guess[16]) {
    if (!e || !init_guess) return LIO_E_INVALID;
    float fm[K], fc[K * K];
    imu_pose_block(e->ukf, fm, fc);
    if (gps) {
        float gm[K], gc[K * K];
        gps_gaussian(*gps, gps->precision, gm, gc);
        gm[2] = fm[2];  // gps_mean(2) = fused_mean(2): the height is the filter's
        if (gps->dimension == 6) {
            float oc[K * K], om[K];
            if (!fusion_pose(fc, gc, fm, gm, K, true, oc, om)) return LIO_E_STATE;
            memcpy(fm, om, sizeof(om));
            memcpy(fc, oc, sizeof(oc));
        } else if (gps->dimension == 2 || gps->dimension == 3) {
            const float c1[4] = {fc[0], fc[1], fc[K], fc[K + 1]}, c2[4] = {gc[0], gc[1], gc[K], gc[K + 1]};
            float oc[4], om[2];
            if (!fusion_pose(c1, c2, fm, gm, 2, false, oc, om)) return LIO_E_STATE;
            fm[0] = om[0]; fm[1] = om[1];
        }
    }
    const Quat q = qnormalized({fm[3], fm[4], fm[5], fm[6]});
    float R[9];
    qtoR(q, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) init_guess[i * 4 + j] = R[i * 3 + j];
        init_guess[i * 4 + 3] = fm[i];
        init_guess[12 + i] = 0.f;
    }
    init_guess[15] = 1.f;
    return LIO_OK;
}

// observe: what match() makes of the matcher's answer (:250-302): the 5 m / 10 deg gate (skipped with a GNSS observation), the
// quaternion hemisphere, the fusion of the observation with the GNSS one; observation_cov is the fused covariance the reference
// returns (the filter's pose block when there is no GNSS observation).  Returns the reference's bool.
int lio_pose_estimator_observe(lio_pose_estimator* e, const float init_guess[16], const float aligned[16], int converged, const lio_gps_observation* gps,
                               float observation[7], float observation_cov[49]) {
    if (!e || !init_guess || !aligned || !observation) return LIO_E_INVALID;
    int result = converged ? 1 : 0;
    const float* Tf = init_guess;
    const float* M = aligned;
    float fm[K], fc[K * K];
    imu_pose_block(e->ukf, fm, fc);
    float gm[K], gc[K * K];
    if (gps) {  // the covariance handed back starts as in the first half (:217-241)
        gps_gaussian(*gps, gps->precision, gm, gc);
        gm[2] = fm[2];
        if (gps->dimension == 6) {
            float oc[K * K], om[K];
            if (!fusion_pose(fc, gc, fm, gm, K, true, oc, om)) return LIO_E_STATE;
            memcpy(fc, oc, sizeof(oc));
        } else if (gps->dimension == 2 || gps->dimension == 3) {
            const float c1[4] = {fc[0], fc[1], fc[K], fc[K + 1]}, c2[4] = {gc[0], gc[1], gc[K], gc[K + 1]};
            float oc[4], om[2];
            if (!fusion_pose(c1, c2, fm, gm, 2, false, oc, om)) return LIO_E_STATE;
            fc[0] = oc[0]; fc[1] = oc[1]; fc[K] = oc[2]; fc[K + 1] = oc[3];
        }
    } else {
        // delta = init_guess^-1 * observation (general 4 x 4 inverse there; rigid here) and the gate
        float D[9], dtv[3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                float s = 0.f;
                for (int k = 0; k < 3; k++) s += Tf[k * 4 + i] * M[k * 4 + j];
                D[i * 3 + j] = s;
            }
        for (int i = 0; i < 3; i++) {
            float s = 0.f;
            for (int k = 0; k < 3; k++) s += Tf[k * 4 + i] * (M[k * 4 + 3] - Tf[k * 4 + 3]);
            dtv[i] = s;
        }
        const float dx = sqrtf(dtv[0] * dtv[0] + dtv[1] * dtv[1] + dtv[2] * dtv[2]);
        const Quat qd = qfromR(D);
        const float vn = sqrtf(qd.x * qd.x + qd.y * qd.y + qd.z * qd.z);
        const float da = 2.0f * atan2f(vn, fabsf(qd.w)) / (float)M_PI * 180.f;  // Eigen::AngleAxisf(R).angle()
        if (dx > 5.0f || da > 10.0f) result = 0;
    }
    float R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = M[i * 4 + j];
    Quat q = qfromR(R);
    const Quat qf = qnormalized({e->ukf.mean[6], e->ukf.mean[7], e->ukf.mean[8], e->ukf.mean[9]});
    if (qf.x * q.x + qf.y * q.y + qf.z * q.z + qf.w * q.w < 0.0f) { q.w = -q.w; q.x = -q.x; q.y = -q.y; q.z = -q.z; }
    observation[0] = M[3]; observation[1] = M[7]; observation[2] = M[11];
    observation[3] = q.w; observation[4] = q.x; observation[5] = q.y; observation[6] = q.z;
    if (gps) {  // :285-299: the matcher's observation (measurement noise) fused with the GNSS one
        if (gps->dimension == 6) {
            float oc[K * K], om[K];
            if (!fusion_pose(e->ukf.measurement_noise, gc, observation, gm, K, true, oc, om)) return LIO_E_STATE;
            memcpy(observation, om, sizeof(om));
            memcpy(fc, oc, sizeof(oc));
        } else if (gps->dimension == 2 || gps->dimension == 3) {
            const float* mn = e->ukf.measurement_noise;
            const float c1[4] = {mn[0], mn[1], mn[K], mn[K + 1]}, c2[4] = {gc[0], gc[1], gc[K], gc[K + 1]};
            float oc[4], om[2];
            if (!fusion_pose(c1, c2, observation, gm, 2, false, oc, om)) return LIO_E_STATE;
            observation[0] = om[0]; observation[1] = om[1];
            fc[0] = oc[0]; fc[1] = oc[1]; fc[K] = oc[2]; fc[K + 1] = oc[3];
        }
    }
    if (observation_cov) memcpy(observation_cov, fc, sizeof(fc));
    return result;
}

// the GNSS-only match (pose_estimator.cpp:304-346, used while no scan is available): a 6-D GNSS pose fused with the filter's pose block
// at ten times its stated precision; anything else hands back the filter's pose and 0 (the reference also sleeps 100 ms there)
int lio_pose_estimator_match_gps_only(lio_pose_estimator* e, const lio_gps_observation* gps, float observation[7], float observation_cov[49]) {
    if (!e || !observation) return LIO_E_INVALID;
    float fm[K], fc[K * K];
    imu_pose_block(e->ukf, fm, fc);
    if (!gps || gps->dimension != 6) {
        memcpy(observation, fm, sizeof(fm));
        return 0;
    }
    float gm[K], gc[K * K], oc[K * K], om[K];
    gps_gaussian(*gps, 10 * gps->precision, gm, gc);
    if (!fusion_pose(fc, gc, fm, gm, K, true, oc, om)) return LIO_E_STATE;
    memcpy(observation, om, sizeof(om));
    if (observation_cov) memcpy(observation_cov, oc, sizeof(oc));
    return 1;
}

// PoseEstimator::match with the device matcher in the middle: guess -> lio_ndt_align -> observe.  gps == NULL: no GNSS observation.
// Returns 1 (use it), 0 (matcher did not converge or the gate fired; the observation is filled all the same, as the reference corrects
// with it regardless), < 0 on error.
int lio_pose_estimator_match_gps(lio_pose_estimator* e, lio_ndt* ndt, lio_scan* source, const lio_ndt_params* params, const lio_gps_observation* gps,
                                 float observation[7], float observation_cov[49], int* iterations) {
    if (!e || !ndt || !source || !observation) return LIO_E_INVALID;
    float Tf[16];
    int rc = lio_pose_estimator_guess(e, gps, Tf);
    if (rc < 0) return rc;
    double guess[16], out[16];
    for (int i = 0; i < 16; i++) guess[i] = (double)Tf[i];
    int it = 0, conv = 0;
    rc = lio_ndt_align(ndt, source, guess, params, out, &it, &conv);
    if (rc < 0) return rc;
    if (iterations) *iterations = it;
    float M[16];
    for (int i = 0; i < 16; i++) M[i] = (float)out[i];
    return lio_pose_estimator_observe(e, Tf, M, conv, gps, observation, observation_cov);
}

int lio_pose_estimator_match(lio_pose_estimator* e, lio_ndt* ndt, lio_scan* source, const lio_ndt_params* params, float observation[7],
                             int* iterations) {
    return lio_pose_estimator_match_gps(e, ndt, source, params, nullptr, observation, nullptr, iterations);
}

}  // extern "C"
