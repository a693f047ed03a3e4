// pose_estimator.cpp -- the localisation loop around the NDT matcher: hdl_localization::PoseEstimator
// (/root/reference/slam/localization/hdl_localization/src/pose_estimator.cpp) = an unscented Kalman filter
// (include/kkl/alg/unscented_kalman_filter.hpp:42-262) over the 23-number pose system of include/hdl_localization/
// pose_system.hpp:14-115 -- predict with or without an IMU sample, scan matching from the predicted pose on the device
// (lio_ndt_align), the too-large-transform gate, quaternion sign continuity, 7-number observation, correct.
// Host C++, f32 like the reference (Eigen::MatrixXf there; plain loops here).  GNSS fusion (fusion_pose, get_timed_pose's INS
// queue) and the fitness score are not part of the path (out of scope).
#include <math.h>
#include <string.h>

#include <algorithm>
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

struct lio_pose_estimator {
    Ukf ukf;
    float process_noise[N * N];
    uint64_t init_stamp = 0, prev_stamp = 0, last_correction_stamp = 0;
    double cool_time_duration = 1.0;
    int bad = 0;  // a Cholesky factorisation failed (covariance not positive definite)
};

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

// PoseEstimator::correct (pose_estimator.cpp:348-360; the INS state queue is not kept)
int lio_pose_estimator_correct(lio_pose_estimator* e, uint64_t stamp_us, const float observation[7]) {
    if (!e || !observation) return LIO_E_INVALID;
    e->last_correction_stamp = stamp_us;
    e->prev_stamp = stamp_us;
    if (!e->ukf.correct(observation)) { e->bad = 1; return LIO_E_STATE; }
    return LIO_OK;
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

// PoseEstimator::match without GNSS (pose_estimator.cpp:188-300): align the downsampled scan from the filter's pose, gate the
// correction at 5 m / 10 deg, keep the quaternion on the filter's hemisphere.  Returns 1 (use it), 0 (matcher did not converge or
// the gate fired; the observation is filled all the same, as the reference corrects with it regardless), < 0 on error.
int lio_pose_estimator_match(lio_pose_estimator* e, lio_ndt* ndt, lio_scan* source, const lio_ndt_params* params, float observation[7],
                             int* iterations) {
    if (!e || !ndt || !source || !observation) return LIO_E_INVALID;
    float Tf[16];
    lio_pose_estimator_matrix(e, Tf);
    double guess[16], out[16];
    for (int i = 0; i < 16; i++) guess[i] = (double)Tf[i];
    int it = 0, conv = 0;
    const int rc = lio_ndt_align(ndt, source, guess, params, out, &it, &conv);
    if (rc < 0) return rc;
    if (iterations) *iterations = it;
    int result = conv ? 1 : 0;
    float M[16];
    for (int i = 0; i < 16; i++) M[i] = (float)out[i];
    // delta = init_guess^-1 * observation (rigid inverse in f32)
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
    float R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = M[i * 4 + j];
    Quat q = qfromR(R);
    const Quat qf = qnormalized({e->ukf.mean[6], e->ukf.mean[7], e->ukf.mean[8], e->ukf.mean[9]});
    if (qf.x * q.x + qf.y * q.y + qf.z * q.z + qf.w * q.w < 0.0f) { q.w = -q.w; q.x = -q.x; q.y = -q.y; q.z = -q.z; }
    observation[0] = M[3]; observation[1] = M[7]; observation[2] = M[11];
    observation[3] = q.w; observation[4] = q.x; observation[5] = q.y; observation[6] = q.z;
    return result;
}

}  // extern "C"
