// eskf_dev.h -- the iterated-ESKF pass of eskf.cpp (Eskf::step + the measurement bookkeeping of engine.hip's measure_pass /
// run_update) written as a sequence of data-parallel phases, so that ONE workgroup per scan can run the filter on the device and
// the iterate loop of esekf::update_iterated_dyn_share_modified (/root/reference/slam/mapping/fastlio/include/IKFoM_toolkit/
// esekfom/esekfom.hpp:1619-1931) needs no host hand-over between passes.
//
// The same source compiles for the host (plain loops; tests/test_eskf_dev.py holds it against Eskf::step) and for the device
// (EK_FOR strides over the workgroup's threads, EK_SYNC is a workgroup barrier).  Every sum keeps the term order of eskf.cpp, so
// the two agree to the last bit except where libm and the device math library round sin / cos / atan differently.
//
// State layout (26 doubles, lio_hip.h): pos 0..2, rot 3..6 (x, y, z, w), R_il 7..10, t_il 11..13, vel 14..16, bg 17..19, ba 20..22,
// grav 23..25.  DoF layout (23): pos 0, rot 3, R_il 6, t_il 9, vel 12, bg 15, ba 18, grav 21.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIP__)  // compiled as HIP: one definition for both passes, the loops differ
#include <hip/hip_runtime.h>
#define EK_FN __host__ __device__ inline
#else
#define EK_FN inline
#endif
#if defined(__HIP__) && defined(LIO_STEP_TRACE)
static __device__ unsigned long long g_ek_trace[64];
static __device__ unsigned long long g_ek_last[2];  // per stamping wave of the step kernel (wave 0: the pass; wave 1: ek_step_prep beside the measurement head)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// on the device a phase of the filter pass is run by ONE wave of the step kernel's workgroup (any one): loops stride over its 64 lanes, a phase
// boundary is a wave-level fence on LDS (a wave's LDS operations complete in order) -- no workgroup barrier inside the pass
#define EK_FOR(i, n) for (int i = (int)(threadIdx.x & 63u); i < (n); i += 64)
#define EK_SYNC()                                                  \
    do {                                                           \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
        __builtin_amdgcn_wave_barrier();                           \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
    } while (0)
#define EK_LANE(k) ((int)(threadIdx.x & 63u) == (k))
#ifdef LIO_STEP_TRACE  // diagnostic build only: cycle stamps of the filter pass's phases (workgroup 0, lane 0), see tools/experiments/step_trace.py
#define EK_STAMP(k)                                                                          \
    do {                                                                                     \
        if ((threadIdx.x & 63u) == 0 && threadIdx.x < 128 && blockIdx.x == 0) {              \
            const unsigned long long now_ = __builtin_readcyclecounter();                    \
            const int w_ = (int)(threadIdx.x >> 6);                                          \
            if ((k) == 0) g_ek_trace[0] += 1;                                                \
            else if ((k) != 10) g_ek_trace[(k)] += now_ - g_ek_last[w_];  /* (10: wave 1 enters the prep -- only sets its clock) */ \
            g_ek_last[w_] = now_;                                                            \
        }                                                                                    \
    } while (0)
#else
#define EK_STAMP(k) ((void)0)
#endif
#else
#define EK_STAMP(k) ((void)0)
#define EK_FOR(i, n) for (int i = 0; i < (n); i++)
#define EK_SYNC() ((void)0)
#define EK_LANE(k) (true)
#endif

namespace lio {

constexpr int kEkN = 23;
constexpr double kEkTol = 1e-11;           // MTK::tolerance<double>()
constexpr double kEkS2Len = 98090.0 / 10000.0;
constexpr int kEkMaxPass = 6;              // maximum_iter + 1 passes are logged (5 with the reference's constants)

struct EkPassLog {   // = lio_pass_log
    int32_t knn, n_eff, valid, degenerate;
    double sum_abs_res;
    double JtJ[36];
    double Jtr[6];
    double dx[23];
};

enum { EK_RUNNING = 0, EK_DONE = 1, EK_NEEDS_HOST = 2, EK_SKIPPED = 3 };

// device-resident control block of one scan's iterated update
struct EskfDev {
    // filter
    double x[26];
    double P[kEkN * kEkN];
    double x_prop[26];
    double P_prop[kEkN * kEkN];
    double limit[kEkN];
    double R;               // LASER_POINT_COV
    int32_t maximum_iter;   // 4
    int32_t degenerate_detect_en;
    // loop state (esekfom.hpp:1623-1634)
    int32_t i;              // pass index, starts at -1
    int32_t t;              // converged passes so far
    int32_t converge;       // dyn_share.converge: the NEXT measurement redoes the neighbour search
    int32_t status;         // EK_*
    int32_t have_prev;      // a valid measurement has been seen (its h_x / h survive in ekfom_data_geo, laserMapping.cpp:991)
    int32_t prev_rows;
    int32_t is_degenerate;
    int32_t n_pass, n_knn;  // measurement evaluations so far / of those with a neighbour search
    int32_t n_log;
    int32_t n_eff_last;
    int32_t joint;          // the sums handed to the filter are joint ones (several sub-maps / ranks, lio_batch's joint mode): the rows of a pass
                            // live on different GPUs, so a pass with N_eff < 23 uses the information form too (as the host filter does behind
                            // a reduce hook) and a pass that needs the degeneracy sums is handed to the host-driven joint path
    double prev_HTH[36], prev_HTh[6];
    EkPassLog log[kEkMaxPass];
};

// scratch of one pass (LDS on the device)
struct EkWork {
    double P[kEkN * kEkN];
    double L[kEkN * kEkN];
    double dx[kEkN], dx_new[kEkN], dx_out[kEkN];
    double Jr[9], Jl[9], Jg[4];
    double G[kEkN * 6], Pi6[kEkN * 6], Kx[kEkN * 6], Kh[kEkN];
    double M6[36], M6i[36], LU[36], ycol[36];
    int32_t piv[6];
    double HTH[36], HTh[6];
    double eigval[3], eigvec[9];
    double cs[6];           // contri[3], strong[3]
    double xn[26];          // the state being updated
    int32_t flag[8];        // [0] measurement valid, [1] need degeneracy sums, [2] final pass, [3] degenerate, [4] n_rows, [5] stop (needs host)
};

// ---- small math, term by term as eskf.cpp -------------------------------------------------------------------------
EK_FN void ek_cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
EK_FN void ek_quat_mul(const double a[4], const double b[4], double o[4]) {
    const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
EK_FN void ek_quat_to_R(const double q[4], double R[9]) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
EK_FN void ek_hat3(const double v[3], double H[9]) {
    H[0] = 0; H[1] = -v[2]; H[2] = v[1];
    H[3] = v[2]; H[4] = 0; H[5] = -v[0];
    H[6] = -v[1]; H[7] = v[0]; H[8] = 0;
}
EK_FN void ek_mm3(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}
EK_FN void ek_cos_sinc_sqrt(double x2, double& c, double& s) {  // mtkmath.hpp:142-174
    const double eps = 2.220446049250313e-16;
    const double taylor_2 = sqrt(eps), taylor_n = sqrt(taylor_2);
    if (x2 >= taylor_n) {
        const double x = sqrt(x2);
        c = cos(x);
        s = sin(x) / x;
        return;
    }
    const double inv[7] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
    double cosi = 1., sinc = 1., term = -1 / 2. * x2;
    for (int i = 0; i < 3; ++i) {
        cosi += term;
        term *= inv[2 * i];
        sinc += term;
        term *= -inv[2 * i + 1] * x2;
    }
    c = cosi;
    s = sinc;
}
EK_FN void ek_so3_exp(const double v[3], double scale, double q[4]) {
    const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double c, s;
    ek_cos_sinc_sqrt(scale * scale * n2, c, s);
    const double m = s * scale;
    q[0] = m * v[0]; q[1] = m * v[1]; q[2] = m * v[2]; q[3] = c;
}
EK_FN void ek_so3_log(const double q[4], double v[3]) {
    double nv = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv < kEkTol) nv = kEkTol;
    const double s = 2.0 / nv * atan(nv / q[3]);
    v[0] = s * q[0]; v[1] = s * q[1]; v[2] = s * q[2];
}
EK_FN void ek_s2_Bx(const double g[3], double Bx[6]) {
    const double L = kEkS2Len;
    if (g[0] + L > kEkTol) {
        Bx[0] = -g[1]; Bx[1] = -g[2];
        Bx[2] = L - g[1] * g[1] / (L + g[0]); Bx[3] = -g[2] * g[1] / (L + g[0]);
        Bx[4] = -g[2] * g[1] / (L + g[0]); Bx[5] = L - g[2] * g[2] / (L + g[0]);
        for (int i = 0; i < 6; i++) Bx[i] /= L;
    } else {
        for (int i = 0; i < 6; i++) Bx[i] = 0;
        Bx[3] = -1;
        Bx[4] = 1;
    }
}
EK_FN void ek_s2_boxplus(double g[3], const double d[2]) {
    double Bx[6];
    ek_s2_Bx(g, Bx);
    const double Bu[3] = {Bx[0] * d[0] + Bx[1] * d[1], Bx[2] * d[0] + Bx[3] * d[1], Bx[4] * d[0] + Bx[5] * d[1]};
    double e[4], R[9];
    ek_so3_exp(Bu, 0.5, e);
    ek_quat_to_R(e, R);
    const double o[3] = {R[0] * g[0] + R[1] * g[1] + R[2] * g[2], R[3] * g[0] + R[4] * g[1] + R[5] * g[2], R[6] * g[0] + R[7] * g[1] + R[8] * g[2]};
    g[0] = o[0]; g[1] = o[1]; g[2] = o[2];
}
EK_FN void ek_s2_boxminus(const double g[3], const double other[3], double res[2]) {
    double c[3];
    ek_cross3(g, other, c);
    const double v_sin = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const double v_cos = g[0] * other[0] + g[1] * other[1] + g[2] * other[2];
    const double theta = atan2(v_sin, v_cos);
    if (v_sin < kEkTol) {
        if (fabs(theta) > kEkTol) { res[0] = 3.1415926; res[1] = 0; }
        else { res[0] = 0; res[1] = 0; }
        return;
    }
    double Bx[6], t[3];
    ek_s2_Bx(other, Bx);
    ek_cross3(other, g, t);
    const double f = theta / v_sin;
    res[0] = f * (Bx[0] * t[0] + Bx[2] * t[1] + Bx[4] * t[2]);
    res[1] = f * (Bx[1] * t[0] + Bx[3] * t[1] + Bx[5] * t[2]);
}
EK_FN void ek_A_matrix(const double v[3], double A[9]) {
    const double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double n = sqrt(sq);
    for (int i = 0; i < 9; i++) A[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n < kEkTol) return;
    double H[9], HH[9];
    ek_hat3(v, H);
    ek_mm3(H, H, HH);
    const double a = (1 - cos(n)) / sq, b = (1 - sin(n) / n) / sq;
    for (int i = 0; i < 9; i++) A[i] += a * H[i] + b * HH[i];
}
EK_FN void ek_s2_Nx_yy(const double g[3], double Nx[6]) {
    double Bx[6], H[9];
    ek_s2_Bx(g, Bx);
    ek_hat3(g, H);
    const double f = 1 / kEkS2Len / kEkS2Len;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += Bx[k * 2 + i] * H[k * 3 + j];
            Nx[i * 3 + j] = f * s;
        }
}
EK_FN void ek_s2_Mx(const double g[3], const double delta[2], double M[6]) {
    double Bx[6], H[9];
    ek_s2_Bx(g, Bx);
    ek_hat3(g, H);
    const double dn = sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
    double L[9];
    if (dn < kEkTol) {
        for (int i = 0; i < 9; i++) L[i] = H[i];
    } else {
        const double Bu[3] = {Bx[0] * delta[0] + Bx[1] * delta[1], Bx[2] * delta[0] + Bx[3] * delta[1], Bx[4] * delta[0] + Bx[5] * delta[1]};
        double e[4], E[9], A[9], At[9], T[9];
        ek_so3_exp(Bu, 0.0, e);  // S2.hpp:277: scalar(1/2) == 0 -> identity
        ek_quat_to_R(e, E);
        ek_A_matrix(Bu, A);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) At[i * 3 + j] = A[j * 3 + i];
        ek_mm3(E, H, T);
        ek_mm3(T, At, L);
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += L[i * 3 + k] * Bx[k * 2 + j];
            M[i * 2 + j] = -s;
        }
}
// boxminus / boxplus on the 26-number array layout
EK_FN void ek_boxminus(const double x[26], const double o[26], double d[kEkN]) {
    for (int i = 0; i < 3; i++) d[i] = x[i] - o[i];
    double c[4], q[4];
    c[0] = -o[3]; c[1] = -o[4]; c[2] = -o[5]; c[3] = o[6];
    ek_quat_mul(c, x + 3, q);
    ek_so3_log(q, d + 3);
    c[0] = -o[7]; c[1] = -o[8]; c[2] = -o[9]; c[3] = o[10];
    ek_quat_mul(c, x + 7, q);
    ek_so3_log(q, d + 6);
    for (int i = 0; i < 3; i++) { d[9 + i] = x[11 + i] - o[11 + i]; d[12 + i] = x[14 + i] - o[14 + i]; d[15 + i] = x[17 + i] - o[17 + i]; d[18 + i] = x[20 + i] - o[20 + i]; }
    ek_s2_boxminus(x + 23, o + 23, d + 21);
}
EK_FN void ek_boxplus(double x[26], const double d[kEkN]) {
    double e[4], q[4];
    for (int i = 0; i < 3; i++) x[i] += d[i];
    ek_so3_exp(d + 3, 0.5, e);
    ek_quat_mul(x + 3, e, q);
    for (int i = 0; i < 4; i++) x[3 + i] = q[i];
    ek_so3_exp(d + 6, 0.5, e);
    ek_quat_mul(x + 7, e, q);
    for (int i = 0; i < 4; i++) x[7 + i] = q[i];
    for (int i = 0; i < 3; i++) { x[11 + i] += d[9 + i]; x[14 + i] += d[12 + i]; x[17 + i] += d[15 + i]; x[20 + i] += d[18 + i]; }
    ek_s2_boxplus(x + 23, d + 21);
}
EK_FN void ek_eig3_sym(const double Ain[9], double w[3], double V[9]) {  // cyclic Jacobi, as eskf.cpp
    double A[9];
    for (int i = 0; i < 9; i++) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; sweep++) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                const double apq = A[p * 3 + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; k++) {
                    const double akp = A[k * 3 + p], akq = A[k * 3 + q];
                    A[k * 3 + p] = c * akp - s * akq;
                    A[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {
                    const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                    A[p * 3 + k] = c * apk - s * aqk;
                    A[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (w[j] < w[i]) {
                const double t = w[i]; w[i] = w[j]; w[j] = t;
                for (int k = 0; k < 3; k++) { const double u = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + j]; V[k * 3 + j] = u; }
            }
}
EK_FN bool ek_inverse3(const double A[9], double out[9]) {  // mat_inverse(A, 3, out): LU with partial pivoting, column by column
    double LU[9];
    int piv[3] = {0, 1, 2};
    for (int i = 0; i < 9; i++) LU[i] = A[i];
    for (int k = 0; k < 3; k++) {
        int p = k;
        double best = fabs(LU[k * 3 + k]);
        for (int i = k + 1; i < 3; i++)
            if (fabs(LU[i * 3 + k]) > best) { best = fabs(LU[i * 3 + k]); p = i; }
        if (best == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < 3; j++) { const double t = LU[k * 3 + j]; LU[k * 3 + j] = LU[p * 3 + j]; LU[p * 3 + j] = t; }
            const int t = piv[k]; piv[k] = piv[p]; piv[p] = t;
        }
        for (int i = k + 1; i < 3; i++) {
            LU[i * 3 + k] /= LU[k * 3 + k];
            const double f = LU[i * 3 + k];
            for (int j = k + 1; j < 3; j++) LU[i * 3 + j] -= f * LU[k * 3 + j];
        }
    }
    for (int col = 0; col < 3; col++) {
        double y[3];
        for (int i = 0; i < 3; i++) {
            double s = (piv[i] == col) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= LU[i * 3 + j] * y[j];
            y[i] = s;
        }
        for (int i = 2; i >= 0; i--) {
            double s = y[i];
            for (int j = i + 1; j < 3; j++) s -= LU[i * 3 + j] * out[j * 3 + col];
            out[i * 3 + col] = s / LU[i * 3 + i];
        }
    }
    return true;
}

// ---- phases -------------------------------------------------------------------------------------------------------
// boxminus / boxplus with their independent pieces on different lanes: the two SO3 blocks on lanes 0 and 1 (one instruction stream),
// the S2 block on lane 2, the vector parts on lane 3 -- the transcendental functions (atan, sin, cos in f64: hundreds of cycles each on
// one lane) then cost the time of the slowest piece, not their sum.  On the host the pieces simply run one after the other.
EK_FN void ek_boxminus_par(const double x[26], const double o[26], double d[kEkN]) {
    for (int blk = 0; blk < 2; blk++) {
        if (EK_LANE(blk)) {
            const int a = 3 + 4 * blk;
            double c[4], q[4];
            c[0] = -o[a]; c[1] = -o[a + 1]; c[2] = -o[a + 2]; c[3] = o[a + 3];
            ek_quat_mul(c, x + a, q);
            ek_so3_log(q, d + 3 + 3 * blk);
        }
    }
    if (EK_LANE(2)) ek_s2_boxminus(x + 23, o + 23, d + 21);
    if (EK_LANE(3)) {
        for (int i = 0; i < 3; i++) {
            d[i] = x[i] - o[i];
            d[9 + i] = x[11 + i] - o[11 + i]; d[12 + i] = x[14 + i] - o[14 + i]; d[15 + i] = x[17 + i] - o[17 + i]; d[18 + i] = x[20 + i] - o[20 + i];
        }
    }
}
EK_FN void ek_boxplus_par(double x[26], const double d[kEkN]) {
    for (int blk = 0; blk < 2; blk++) {
        if (EK_LANE(blk)) {
            const int a = 3 + 4 * blk;
            double e[4], q[4];
            ek_so3_exp(d + 3 + 3 * blk, 0.5, e);
            ek_quat_mul(x + a, e, q);
            for (int i = 0; i < 4; i++) x[a + i] = q[i];
        }
    }
    if (EK_LANE(2)) ek_s2_boxplus(x + 23, d + 21);
    if (EK_LANE(3)) {
        for (int i = 0; i < 3; i++) { x[i] += d[i]; x[11 + i] += d[9 + i]; x[14 + i] += d[12 + i]; x[17 + i] += d[15 + i]; x[20 + i] += d[18 + i]; }
    }
}

// manifold Jacobians of esekfom.hpp:1661-1699 from (x, x_prop, dx): three independent pieces, one thread each
EK_FN void ek_jacobians(const double x[26], const double x_prop[26], const double dx[kEkN], EkWork& w) {
    for (int blk = 0; blk < 2; blk++) {
        if (EK_LANE(blk)) {  // lanes 0 and 1 run the two SO3 blocks side by side
            double A[9];
            ek_A_matrix(dx + 3 + 3 * blk, A);
            double* J = blk == 0 ? w.Jr : w.Jl;
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) J[a * 3 + b] = A[b * 3 + a];
        }
    }
    if (EK_LANE(2)) {
        double Nx[6], Mx[6];
        ek_s2_Nx_yy(x + 23, Nx);
        ek_s2_Mx(x_prop + 23, dx + 21, Mx);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += Nx[a * 3 + k] * Mx[k * 2 + b];
                w.Jg[a * 2 + b] = s;
            }
    }
}
// M[idx:idx+dim, c] = J * M[idx:idx+dim, c] for every column c < cols (row-major, leading dimension ld): parallel over columns
EK_FN void ek_rows_mul(double* M, int cols, int ld, int idx, int dim, const double* J) {
    EK_FOR(c, cols) {
        double v[3];
        for (int a = 0; a < dim; a++) v[a] = M[(idx + a) * ld + c];
        for (int a = 0; a < dim; a++) {
            double s = 0;
            for (int b = 0; b < dim; b++) s += J[a * dim + b] * v[b];
            M[(idx + a) * ld + c] = s;
        }
    }
}
// M[r, idx:idx+dim] = M[r, idx:idx+dim] * J^T for every row r < rows: parallel over rows
EK_FN void ek_cols_mul_T(double* M, int rows, int ld, int idx, int dim, const double* J) {
    EK_FOR(r, rows) {
        double v[3];
        for (int a = 0; a < dim; a++) v[a] = M[r * ld + idx + a];
        for (int a = 0; a < dim; a++) {
            double s = 0;
            for (int b = 0; b < dim; b++) s += v[b] * J[a * dim + b];
            M[r * ld + idx + a] = s;
        }
    }
}
// 6 x 6 inverse, LU with partial pivoting (mat_inverse of eskf.cpp): the factorisation is serial in k, its row updates and the six
// column solves run in parallel
EK_FN void ek_inverse6(EkWork& w) {
    EK_FOR(i, 36) w.LU[i] = w.M6[i];
    EK_FOR(i, 6) w.piv[i] = i;
    EK_SYNC();
    for (int k = 0; k < 6; k++) {
        if (EK_LANE(0)) {
            int p = k;
            double best = fabs(w.LU[k * 6 + k]);
            for (int i = k + 1; i < 6; i++)
                if (fabs(w.LU[i * 6 + k]) > best) { best = fabs(w.LU[i * 6 + k]); p = i; }
            if (p != k) {
                for (int j = 0; j < 6; j++) { const double t = w.LU[k * 6 + j]; w.LU[k * 6 + j] = w.LU[p * 6 + j]; w.LU[p * 6 + j] = t; }
                const int t = w.piv[k]; w.piv[k] = w.piv[p]; w.piv[p] = t;
            }
            for (int i = k + 1; i < 6; i++) w.LU[i * 6 + k] /= w.LU[k * 6 + k];
        }
        EK_SYNC();
        EK_FOR(e, 36) {
            const int i = e / 6, j = e % 6;
            if (i > k && j > k) w.LU[i * 6 + j] -= w.LU[i * 6 + k] * w.LU[k * 6 + j];
        }
        EK_SYNC();
    }
    EK_FOR(col, 6) {
        double* y = w.ycol + col * 6;
        for (int i = 0; i < 6; i++) {
            double s = (w.piv[i] == col) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= w.LU[i * 6 + j] * y[j];
            y[i] = s;
        }
        for (int i = 5; i >= 0; i--) {
            double s = y[i];
            for (int j = i + 1; j < 6; j++) s -= w.LU[i * 6 + j] * w.M6i[j * 6 + col];
            w.M6i[i * 6 + col] = s / w.LU[i * 6 + i];
        }
    }
    EK_SYNC();
}

// The head of a pass: what engine.hip's measure_pass + run_update's `measure` lambda do with the 29 sums of a linearisation.
//   acc29: 21 J^T J (upper triangle row by row), 6 J^T h, sum |r|, N_eff
// Decides validity (stale rows of the previous pass survive an empty measurement), runs the eigen-decomposition of sum n n^T and
// the eigenvalue bound, and leaves in w.flag[1] whether the six degeneracy sums are needed (w.eigvec then holds the directions).
EK_FN void ek_measure_head(EskfDev& c, EkWork& w, const double acc29[29], int knn_this_pass) {
    if (EK_LANE(0)) {
        const int n_eff = (int)(acc29[28] + 0.5);
        EkPassLog& pl = c.log[c.n_log < kEkMaxPass ? c.n_log : kEkMaxPass - 1];
        pl.knn = knn_this_pass;
        pl.n_eff = n_eff;
        pl.sum_abs_res = acc29[27];
        pl.valid = 0;
        pl.degenerate = 0;
        for (int k = 0; k < 36; k++) pl.JtJ[k] = 0;
        for (int k = 0; k < 6; k++) pl.Jtr[k] = 0;
        for (int k = 0; k < 23; k++) pl.dx[k] = 0;
        c.n_pass++;
        if (knn_this_pass) c.n_knn++;
        c.n_eff_last = n_eff;
        w.flag[0] = 0; w.flag[1] = 0; w.flag[3] = 0; w.flag[4] = n_eff; w.flag[5] = 0;
        if (n_eff >= 1) {
            w.flag[0] = 1;
            int t = 0;
            for (int a = 0; a < 6; a++)
                for (int b = a; b < 6; b++) { w.HTH[a * 6 + b] = acc29[t]; w.HTH[b * 6 + a] = acc29[t]; t++; }
            for (int a = 0; a < 6; a++) w.HTh[a] = acc29[21 + a];
            for (int k = 0; k < 6; k++) w.cs[k] = INFINITY;
            if (c.degenerate_detect_en) {
                double nnT[9];
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++) nnT[a * 3 + b] = w.HTH[a * 6 + b];
                // Every eigenvalue is at least the smallest Gershgorin bound a_ii - sum_j |a_ij|: if that already passes the test below,
                // all three do and the eigen-decomposition (a serial ~2 us on one lane) is not needed -- same decision, well-conditioned
                // scenes only
                double gersh = INFINITY;
                for (int a = 0; a < 3; a++) {
                    double g = nnT[a * 3 + a];
                    for (int b = 0; b < 3; b++)
                        if (b != a) g -= fabs(nnT[a * 3 + b]);
                    if (g < gersh) gersh = g;
                }
                bool need = false;
                if (!(gersh * (1.0 - 1e-5) - 0.030138 * (double)n_eff >= 250.0 + 1e-3)) {
                    ek_eig3_sym(nnT, w.eigval, w.eigvec);
                    for (int i = 0; i < 3; i++)
                        if (!(w.eigval[i] * (1.0 - 1e-5) - 0.030138 * (double)n_eff >= 250.0 + 1e-3)) need = true;
                }
                if (need) { w.flag[1] = 1; for (int k = 0; k < 6; k++) w.cs[k] = 0.0; }
            }
        }
    }
    EK_SYNC();
}

// After the degeneracy sums (if any): the projection of laserMapping.cpp:965-980 on the 6 x 6 normal equations, the stale-measurement
// rule, the N_eff < 23 exit.  Returns (in w.flag[0]) whether a filter step follows.
EK_FN void ek_measure_tail(EskfDev& c, EkWork& w) {
    if (EK_LANE(0)) {
        EkPassLog& pl = c.log[c.n_log < kEkMaxPass ? c.n_log : kEkMaxPass - 1];
        if (w.flag[0]) {
            bool degenerate = false;
            if (c.degenerate_detect_en) {
                bool keep[3];
                for (int i = 0; i < 3; i++) {
                    keep[i] = !((float)w.cs[i] < 250.0f && (float)w.cs[3 + i] < 50.0f);
                    if (!keep[i]) degenerate = true;
                }
                c.is_degenerate = degenerate ? 1 : 0;
                if (degenerate) {
                    double Vt[9], V2[9], Vti[9], Pm[9];
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) {
                            Vt[a * 3 + b] = w.eigvec[b * 3 + a];
                            V2[a * 3 + b] = keep[a] ? Vt[a * 3 + b] : 0.0;
                        }
                    ek_inverse3(Vt, Vti);
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) {
                            double s = 0;
                            for (int k = 0; k < 3; k++) s += Vti[a * 3 + k] * V2[k * 3 + b];
                            Pm[a * 3 + b] = s;
                        }
                    double M[36], T[36], O[36];
                    for (int k = 0; k < 36; k++) M[k] = 0;
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) M[a * 6 + b] = Pm[a * 3 + b];
                    for (int a = 3; a < 6; a++) M[a * 6 + a] = 1.0;
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++) {
                            double s = 0;
                            for (int k = 0; k < 6; k++) s += M[a * 6 + k] * w.HTH[k * 6 + b];
                            T[a * 6 + b] = s;
                        }
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++) {
                            double s = 0;
                            for (int k = 0; k < 6; k++) s += T[a * 6 + k] * M[b * 6 + k];
                            O[a * 6 + b] = s;
                        }
                    for (int k = 0; k < 36; k++) w.HTH[k] = O[k];
                    double v[6];
                    for (int a = 0; a < 6; a++) {
                        double s = 0;
                        for (int k = 0; k < 6; k++) s += M[a * 6 + k] * w.HTh[k];
                        v[a] = s;
                    }
                    for (int a = 0; a < 6; a++) w.HTh[a] = v[a];
                }
            }
            w.flag[3] = degenerate ? 1 : 0;
            if (w.flag[4] < kEkN && !c.joint) {
                // the dense branch of the filter (esekfom.hpp:1715-1744) needs the rows themselves: the host takes over from this pass
                w.flag[5] = 1;
                c.status = EK_NEEDS_HOST;
                c.n_pass--;                    // the host evaluates this pass again
                if (pl.knn) c.n_knn--;
            } else {
                pl.valid = 1;
                pl.degenerate = degenerate ? 1 : 0;
                for (int k = 0; k < 36; k++) { pl.JtJ[k] = w.HTH[k]; c.prev_HTH[k] = w.HTH[k]; }
                for (int k = 0; k < 6; k++) { pl.Jtr[k] = w.HTh[k]; c.prev_HTh[k] = w.HTh[k]; }
                c.have_prev = 1;
                c.prev_rows = w.flag[4];
            }
        } else if (c.have_prev) {  // "No Effective Points": h_x / h of the previous pass survive in the copied struct
            w.flag[0] = 1;
            w.flag[4] = c.prev_rows;
            for (int k = 0; k < 36; k++) { w.HTH[k] = c.prev_HTH[k]; pl.JtJ[k] = w.HTH[k]; }
            for (int k = 0; k < 6; k++) { w.HTh[k] = c.prev_HTh[k]; pl.Jtr[k] = w.HTh[k]; }
            pl.valid = 1;
        }
        if (!w.flag[5]) {
            if (!w.flag[0]) {  // invalid pass: `continue` of esekfom.hpp:1638-1641 -- only the pass counter moves
                c.n_log++;
                c.i++;
                if (c.i >= c.maximum_iter) c.status = EK_DONE;
            }
        }
    }
    EK_SYNC();
}

// Eskf::step (eskf.cpp): one pass of the iterated update with a valid measurement in w.HTH / w.HTh, in two parts.
// ek_step_prep is everything that does NOT depend on the measurement -- boxminus of the iterate against the propagated state, the
// manifold Jacobians, P <- J P_prop J^T, (P / R)[:, 0:6] -- so the step kernel runs it on a second wave WHILE the first one folds the
// partial sums and decides about the measurement; ek_step_solve is the rest.  ek_step = prep, then solve (the host, and the order of
// every floating-point operation, are those of the one-piece form).
EK_FN void ek_step_prep(const EskfDev& c, EkWork& w) {
    constexpr int N = kEkN;
    const double R = c.R;
    EK_STAMP(10);
    ek_boxminus_par(c.x, c.x_prop, w.dx);
    EK_SYNC();
    EK_STAMP(11);
    ek_jacobians(c.x, c.x_prop, w.dx, w);
    EK_FOR(k, N) w.dx_new[k] = w.dx[k];
    EK_FOR(k, N * N) w.P[k] = c.P_prop[k];
    EK_SYNC();
    EK_STAMP(12);
    if (EK_LANE(0)) {
        // the three blocks of dx_new are disjoint: one thread applies them in turn
        double v[3];
        for (int a = 0; a < 3; a++) v[a] = w.dx_new[3 + a];
        for (int a = 0; a < 3; a++) { double s = 0; for (int b = 0; b < 3; b++) s += w.Jr[a * 3 + b] * v[b]; w.dx_new[3 + a] = s; }
        for (int a = 0; a < 3; a++) v[a] = w.dx_new[6 + a];
        for (int a = 0; a < 3; a++) { double s = 0; for (int b = 0; b < 3; b++) s += w.Jl[a * 3 + b] * v[b]; w.dx_new[6 + a] = s; }
        for (int a = 0; a < 2; a++) v[a] = w.dx_new[21 + a];
        for (int a = 0; a < 2; a++) { double s = 0; for (int b = 0; b < 2; b++) s += w.Jg[a * 2 + b] * v[b]; w.dx_new[21 + a] = s; }
    }
    ek_rows_mul(w.P, N, N, 3, 3, w.Jr); EK_SYNC();
    ek_cols_mul_T(w.P, N, N, 3, 3, w.Jr); EK_SYNC();
    ek_rows_mul(w.P, N, N, 6, 3, w.Jl); EK_SYNC();
    ek_cols_mul_T(w.P, N, N, 6, 3, w.Jl); EK_SYNC();
    ek_rows_mul(w.P, N, N, 21, 2, w.Jg); EK_SYNC();
    ek_cols_mul_T(w.P, N, N, 21, 2, w.Jg); EK_SYNC();
    EK_STAMP(13);
    EK_FOR(e, N * 6) { const int a = e / 6, col = e % 6; w.G[e] = w.P[a * N + col] / R; }
    EK_SYNC();
    EK_STAMP(22);
}

EK_FN void ek_step_solve(EskfDev& c, EkWork& w) {
    constexpr int N = kEkN;
    // information form on the leading 6 x 6 block (see eskf.cpp): P_inv[:, 0:6] = (P / R)[:, 0:6] (I6 + HTH (P / R)_66)^-1  (w.G from the prep)
    EK_FOR(e, 36) {
        const int a = e / 6, col = e % 6;
        double v = (a == col) ? 1.0 : 0.0;
        for (int k = 0; k < 6; k++) v += w.HTH[a * 6 + k] * w.G[k * 6 + col];
        w.M6[e] = v;
    }
    EK_SYNC();
    EK_STAMP(14);
    ek_inverse6(w);
    EK_STAMP(15);
    EK_FOR(e, N * 6) {
        const int a = e / 6, col = e % 6;
        double v = 0;
        for (int k = 0; k < 6; k++) v += w.G[a * 6 + k] * w.M6i[k * 6 + col];
        w.Pi6[e] = v;
    }
    EK_SYNC();
    EK_FOR(e, N * 7) {
        const int a = e / 7, col = e % 7;
        if (col == 6) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += w.Pi6[a * 6 + k] * w.HTh[k];
            w.Kh[a] = s;
        } else {
            double v = 0;
            for (int k = 0; k < 6; k++) v += w.Pi6[a * 6 + k] * w.HTH[k * 6 + col];
            w.Kx[a * 6 + col] = v;
        }
    }
    EK_SYNC();
    EK_FOR(a, N) {  // dx_ = K_h + (K_x - I) dx_new   (K_x is zero beyond its sixth column)
        double s = w.Kh[a];
        for (int b = 0; b < N; b++) s += ((b < 6 ? w.Kx[a * 6 + b] : 0.0) - (a == b ? 1.0 : 0.0)) * w.dx_new[b];
        w.dx_out[a] = s;
    }
    EK_SYNC();
    EK_STAMP(16);
    ek_boxplus_par(c.x, w.dx_out);
    EK_STAMP(17);
    {
        EkPassLog& pl = c.log[c.n_log < kEkMaxPass ? c.n_log : kEkMaxPass - 1];
        EK_FOR(k, N) pl.dx[k] = w.dx_out[k];
    }
    EK_SYNC();
    if (EK_LANE(0)) {
        c.n_log++;
        bool converge = true;
        for (int a = 0; a < N; a++)
            if (fabs(w.dx_out[a]) > c.limit[a]) { converge = false; break; }
        if (converge) c.t++;
        if (!c.t && c.i == c.maximum_iter - 2) converge = true;  // force one more neighbour search on the last pass
        c.converge = converge ? 1 : 0;
        w.flag[2] = (c.t > 1 || c.i == c.maximum_iter - 1) ? 1 : 0;
        c.i++;
    }
    EK_SYNC();
    EK_STAMP(18);
    if (!w.flag[2]) {
        // like the reference's P_ member, the filter's covariance is left as this pass transformed it (it only matters if every
        // later pass turns out invalid; a later valid pass starts from P_prop again)
        EK_FOR(k, N * N) c.P[k] = w.P[k];
        EK_SYNC();
        return;
    }
    // final covariance: P = L - K_x[:, 0:15] P[0:15, :] with the manifold Jacobians rebuilt from dx_ (esekfom.hpp:1836-1924)
    ek_jacobians(c.x, c.x_prop, w.dx_out, w);
    EK_FOR(k, N * N) w.L[k] = w.P[k];
    EK_SYNC();
    for (int blk = 0; blk < 3; blk++) {
        const int idx = blk == 0 ? 3 : (blk == 1 ? 6 : 21), dim = blk == 2 ? 2 : 3;
        const double* J = blk == 0 ? w.Jr : (blk == 1 ? w.Jl : w.Jg);
        EK_FOR(col, N) {  // L rows come from P's rows (P itself is only column-transformed)
            for (int a = 0; a < dim; a++) {
                double s = 0;
                for (int b = 0; b < dim; b++) s += J[a * dim + b] * w.P[(idx + b) * N + col];
                w.L[(idx + a) * N + col] = s;
            }
        }
        ek_rows_mul(w.Kx, 6, 6, idx, dim, J);  // rows idx.. of K_x (its columns 6..14 are zero)
        EK_SYNC();
        ek_cols_mul_T(w.L, N, N, idx, dim, J);
        ek_cols_mul_T(w.P, N, N, idx, dim, J);
        EK_SYNC();
    }
    EK_FOR(e, N * N) {
        const int a = e / N, b = e % N;
        double s = 0;
        for (int k = 0; k < 6; k++) s += w.Kx[a * 6 + k] * w.P[k * N + b];
        c.P[e] = w.L[e] - s;
    }
    if (EK_LANE(0)) c.status = EK_DONE;
    EK_SYNC();
    EK_STAMP(19);
}

EK_FN void ek_step(EskfDev& c, EkWork& w) {
    ek_step_prep(c, w);
    ek_step_solve(c, w);
}

// begin of an update: esekfom.hpp:1623-1634
EK_FN void ek_begin(EskfDev& c) {
    EK_FOR(k, 26) c.x_prop[k] = c.x[k];
    EK_FOR(k, kEkN * kEkN) c.P_prop[k] = c.P[k];
    if (EK_LANE(0)) {
        c.i = -1; c.t = 0; c.converge = 1; c.status = EK_RUNNING; c.have_prev = 0; c.prev_rows = 0;
        c.n_pass = 0; c.n_knn = 0; c.n_log = 0; c.n_eff_last = 0;
    }
    EK_SYNC();
}

}  // namespace lio
