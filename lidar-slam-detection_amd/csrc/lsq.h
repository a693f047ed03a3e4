// lsq.h -- the Levenberg-Marquardt driver on SE(3) shared by the matchers (NDT, GICP): fast_gicp's LsqRegistration
// (/root/reference/slam/thirdparty/fast_gicp/include/fast_gicp/gicp/impl/lsq_registration_impl.hpp:71-208) with so3.hpp's se3_exp
// (:80-105), host side, f64.  The cost functions live on the device (ndt.hip, gicp.hip) behind two callables.
#pragma once
#include <math.h>
#include <string.h>

#include <chrono>
#include <cmath>

#include "../../include/lio_hip.h"

#if defined(__HIPCC__)
#define LSQ_FN __host__ __device__ inline
#else
#define LSQ_FN inline
#endif

namespace lio {

// ---- SE(3) and the LM step (lsq_registration_impl.hpp, so3.hpp), f64: host, and device for the batched aligner (ndt.hip) -----------
LSQ_FN void se3_exp_h(const double a[6], double T[16]) {
    const double wx = a[0], wy = a[1], wz = a[2];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    double imag, real;
    if (theta_sq < 1e-10) {
        const double tq = theta_sq * theta_sq;
        imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * tq;
        real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * tq;
    } else {
        const double th = sqrt(theta_sq), h = 0.5 * th;
        imag = sin(h) / th;
        real = cos(h);
    }
    const double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
    const double theta = sqrt(theta_sq);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += O[i * 3 + k] * O[k * 3 + j]; O2[i * 3 + j] = s; }
    if (theta < 1e-10) memcpy(V, R, sizeof(V));
    else {
        const double tsq = theta * theta;
        for (int k = 0; k < 9; k++) V[k] = ((k % 4 == 0) ? 1.0 : 0.0) + (1.0 - cos(theta)) / tsq * O[k] + (theta - sin(theta)) / (tsq * theta) * O2[k];
    }
    for (int k = 0; k < 16; k++) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = V[i * 3] * a[3] + V[i * 3 + 1] * a[4] + V[i * 3 + 2] * a[5];
    }
}
LSQ_FN void mul44_h(const double A[16], const double B[16], double C[16]) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j]; C[i * 4 + j] = s; }
}
LSQ_FN double rot_angle_deg_h(const double T[16]) {  // Eigen::AngleAxisd(delta.linear()).angle() / pi * 180 (via the quaternion)
    const double tr = T[0] + T[5] + T[10];
    double w, x, y, z;
    if (tr > 0) {
        double t = sqrt(tr + 1.0);
        w = 0.5 * t; t = 0.5 / t;
        x = (T[9] - T[6]) * t; y = (T[2] - T[8]) * t; z = (T[4] - T[1]) * t;
    } else {
        int i = 0;
        if (T[5] > T[0]) i = 1;
        if (T[10] > T[i * 5]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(T[i * 5] - T[j * 5] - T[k * 5] + 1.0);
        double q[3];
        q[i] = 0.5 * t; t = 0.5 / t;
        w = (T[k * 4 + j] - T[j * 4 + k]) * t;
        q[j] = (T[j * 4 + i] + T[i * 4 + j]) * t;
        q[k] = (T[k * 4 + i] + T[i * 4 + k]) * t;
        x = q[0]; y = q[1]; z = q[2];
    }
    return 2.0 * atan2(sqrt(x * x + y * y + z * z), fabs(w)) / M_PI * 180.0;
}
// (H + lambda I) d = -b by LDL^T on the lower triangle (the reference: Eigen::LDLT<Matrix<double,6,6>>, same solution)
LSQ_FN bool ldlt_solve6(const double A[36], const double rhs[6], double x[6]) {
    double L[36] = {0}, D[6];
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k] * D[k];
        if (d == 0.0 || !(d - d == 0.0)) return false;  // (zero pivot, NaN or infinity)
        D[j] = d;
        L[j * 6 + j] = 1.0;
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k] * D[k];
            L[i * 6 + j] = s / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s; }
    for (int i = 0; i < 6; i++) y[i] /= D[i];
    for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k]; x[i] = s; }
    return true;
}
LSQ_FN bool converged_h(const lio_ndt_params& p, const double D[16], double loosen) {
    const double r_delta = 1.0 / (p.rotation_epsilon_deg * loosen) * rot_angle_deg_h(D);
    double tmax = 0;
    for (int i = 0; i < 3; i++) tmax = fmax(tmax, 1.0 / (p.transformation_epsilon * loosen) * fabs(D[i * 4 + 3]));
    return fmax(r_delta, tmax) < 1;
}


// linearize(x, H, b, &y): cost, 6 x 6 and 6 at x (correspondences refreshed there); error(x_lin, x, &y): cost at x on what the last
// linearisation at x_lin cached.  Both return LIO_OK or an error code.
template <typename Lin, typename Err>
int lsq_align(const lio_ndt_params& p, const double guess[16], Lin&& linearize, Err&& error, double out[16], int* iterations, int* converged) {
    double x0[16];
    memcpy(x0, guess, sizeof(x0));
    double lambda = -1.0;
    bool conv = false;
    int it_done = 0;
    const auto clock0 = std::chrono::steady_clock::now();
    for (int it = 0; it < p.max_iterations && !conv; it++) {  // LsqRegistration::computeTransformation
        it_done = it;
        double H[36], b[6], delta[16], y0 = 0;
        int rc = linearize(x0, H, b, &y0);
        if (rc != LIO_OK) return rc;
        if (lambda < 0.0) {
            double mx = 0;
            for (int i = 0; i < 6; i++) mx = fmax(mx, fabs(H[i * 7]));
            lambda = p.lm_init_lambda_factor * mx;
        }
        double nu = 2.0;
        bool ok = false;
        for (int i = 0; i < p.lm_max_iterations; i++) {  // step_lm
            double A[36], nb[6], d[6];
            for (int k = 0; k < 36; k++) A[k] = H[k] + ((k % 7 == 0) ? lambda : 0.0);
            for (int k = 0; k < 6; k++) nb[k] = -b[k];
            if (!ldlt_solve6(A, nb, d)) break;
            se3_exp_h(d, delta);
            double xi[16], yi = 0;
            mul44_h(delta, x0, xi);
            rc = error(x0, xi, &yi);
            if (rc != LIO_OK) return rc;
            double den = 0;
            for (int k = 0; k < 6; k++) den += d[k] * (lambda * d[k] - b[k]);
            const double rho = (y0 - yi) / den;
            if (rho < 0) {
                if (converged_h(p, delta, 10.0)) { ok = true; break; }
                lambda = nu * lambda;
                nu = 2 * nu;
                continue;
            }
            memcpy(x0, xi, sizeof(x0));
            lambda = lambda * fmax(1.0 / 3.0, 1 - pow(2 * rho - 1, 3));
            ok = true;
            break;
        }
        if (!ok) break;  // "lm not converged!!"
        conv = converged_h(p, delta, 1.0);
        if (p.max_process_time_ms > 0) {  // lsq_registration_impl.hpp:94-104
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - clock0).count();
            if (ms > p.max_process_time_ms && converged_h(p, delta, 10.0)) { conv = true; break; }
            else if (ms > 1.5 * p.max_process_time_ms) break;
        }
    }
    memcpy(out, x0, sizeof(x0));
    if (iterations) *iterations = it_done;
    if (converged) *converged = conv ? 1 : 0;
    return LIO_OK;
}

// The same driver with the trial evaluation and the linearisation that follows an accepted step fetched TOGETHER: spec(x_lin, xi, H', b', &y', &yi)
// returns the cost yi at xi on the pairs cached at x_lin (what error() returns) and, for the pairs refreshed at xi, cost / H / b (what the
// next iteration's linearize(xi) returns); commit() tells the matcher that xi was accepted (its refreshed pairs become the cached ones).
// Decisions, poses and iteration counts are those of lsq_align: the numbers are the same, they only arrive one hand-over earlier.
template <typename Lin, typename Spec, typename Commit>
int lsq_align_spec(const lio_ndt_params& p, const double guess[16], Lin&& linearize, Spec&& spec, Commit&& commit, double out[16], int* iterations,
                   int* converged) {
    double x0[16];
    memcpy(x0, guess, sizeof(x0));
    double lambda = -1.0;
    bool conv = false, have_next = false;
    int it_done = 0;
    double Hn[36], bn[6], yn = 0;
    const auto clock0 = std::chrono::steady_clock::now();
    for (int it = 0; it < p.max_iterations && !conv; it++) {
        it_done = it;
        double H[36], b[6], delta[16], y0 = 0;
        if (have_next) {
            memcpy(H, Hn, sizeof(H));
            memcpy(b, bn, sizeof(b));
            y0 = yn;
            have_next = false;
        } else {
            const int rc = linearize(x0, H, b, &y0);
            if (rc != LIO_OK) return rc;
        }
        if (lambda < 0.0) {
            double mx = 0;
            for (int i = 0; i < 6; i++) mx = fmax(mx, fabs(H[i * 7]));
            lambda = p.lm_init_lambda_factor * mx;
        }
        double nu = 2.0;
        bool ok = false;
        for (int i = 0; i < p.lm_max_iterations; i++) {  // step_lm
            double A[36], nb[6], d[6];
            for (int k = 0; k < 36; k++) A[k] = H[k] + ((k % 7 == 0) ? lambda : 0.0);
            for (int k = 0; k < 6; k++) nb[k] = -b[k];
            if (!ldlt_solve6(A, nb, d)) break;
            se3_exp_h(d, delta);
            double xi[16], yi = 0;
            mul44_h(delta, x0, xi);
            const int rc = spec(x0, xi, Hn, bn, &yn, &yi);
            if (rc != LIO_OK) return rc;
            double den = 0;
            for (int k = 0; k < 6; k++) den += d[k] * (lambda * d[k] - b[k]);
            const double rho = (y0 - yi) / den;
            if (rho < 0) {
                if (converged_h(p, delta, 10.0)) { ok = true; break; }
                lambda = nu * lambda;
                nu = 2 * nu;
                continue;
            }
            memcpy(x0, xi, sizeof(x0));
            lambda = lambda * fmax(1.0 / 3.0, 1 - pow(2 * rho - 1, 3));
            ok = true;
            have_next = true;
            commit();
            break;
        }
        if (!ok) break;  // "lm not converged!!"
        conv = converged_h(p, delta, 1.0);
        if (p.max_process_time_ms > 0) {  // lsq_registration_impl.hpp:94-104
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - clock0).count();
            if (ms > p.max_process_time_ms && converged_h(p, delta, 10.0)) { conv = true; break; }
            else if (ms > 1.5 * p.max_process_time_ms) break;
        }
    }
    memcpy(out, x0, sizeof(x0));
    if (iterations) *iterations = it_done;
    if (converged) *converged = conv ? 1 : 0;
    return LIO_OK;
}

}  // namespace lio
