// eskf.cpp -- see eskf.h.  Plain host C++ (row-major fixed-size arrays; no Eigen, no Boost).
#include "eskf.h"
#include "eskf_dev.h"
#include "../../include/lio_hip.h"

#include <math.h>
#include <string.h>

#include <vector>

namespace lio {

namespace {

constexpr double kTol = 1e-11;  // MTK::tolerance<double>()
constexpr int N = kDof;

inline void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
inline void quat_mul(const double a[4], const double b[4], double o[4]) {  // Eigen coefficient order (x, y, z, w)
    const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
inline void quat_to_R(const double q[4], double R[9]) {  // Eigen::QuaternionBase::toRotationMatrix
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline void hat3(const double v[3], double H[9]) {
    H[0] = 0; H[1] = -v[2]; H[2] = v[1];
    H[3] = v[2]; H[4] = 0; H[5] = -v[0];
    H[6] = -v[1]; H[7] = v[0]; H[8] = 0;
}
inline void mm3(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}
// mtkmath.hpp:142-174: cos(sqrt(x2)) and sin(sqrt(x2))/sqrt(x2), series near zero
inline void cos_sinc_sqrt(double x2, double& c, double& s) {
    const double eps = 2.220446049250313e-16;
    const double taylor_2 = sqrt(eps), taylor_n = sqrt(taylor_2);
    if (x2 >= taylor_n) {
        const double x = sqrt(x2);
        c = cos(x);
        s = sin(x) / x;
        return;
    }
    static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
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
// mtkmath.hpp:249-256
inline void so3_exp(const double v[3], double scale, double q[4]) {
    const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double c, s;
    cos_sinc_sqrt(scale * scale * n2, c, s);
    const double m = s * scale;
    q[0] = m * v[0]; q[1] = m * v[1]; q[2] = m * v[2]; q[3] = c;
}
// mtkmath.hpp:268-288 with plus/minus periodicity, scale 2 (SOn.hpp:293-297)
inline void so3_log(const double q[4], double v[3]) {
    double nv = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv < kTol) nv = kTol;
    const double s = 2.0 / nv * atan(nv / q[3]);
    v[0] = s * q[0]; v[1] = s * q[1]; v[2] = s * q[2];
}
// S2.hpp:179-245, S2_typ == 1
inline void s2_Bx(const double g[3], double Bx[6]) {  // 3 x 2
    const double L = kS2Length;
    if (g[0] + L > kTol) {
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
inline void s2_boxplus(double g[3], const double d[2]) {  // S2.hpp:136-142
    double Bx[6];
    s2_Bx(g, Bx);
    const double Bu[3] = {Bx[0] * d[0] + Bx[1] * d[1], Bx[2] * d[0] + Bx[3] * d[1], Bx[4] * d[0] + Bx[5] * d[1]};
    double e[4], R[9];
    so3_exp(Bu, 0.5, e);
    quat_to_R(e, R);
    const double o[3] = {R[0] * g[0] + R[1] * g[1] + R[2] * g[2], R[3] * g[0] + R[4] * g[1] + R[5] * g[2], R[6] * g[0] + R[7] * g[1] + R[8] * g[2]};
    g[0] = o[0]; g[1] = o[1]; g[2] = o[2];
}
inline void s2_boxminus(const double g[3], const double other[3], double res[2]) {  // S2.hpp:144-167
    double c[3];
    cross3(g, other, c);
    const double v_sin = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const double v_cos = g[0] * other[0] + g[1] * other[1] + g[2] * other[2];
    const double theta = atan2(v_sin, v_cos);
    if (v_sin < kTol) {
        if (fabs(theta) > kTol) { res[0] = 3.1415926; res[1] = 0; }
        else { res[0] = 0; res[1] = 0; }
        return;
    }
    double Bx[6], t[3];
    s2_Bx(other, Bx);
    cross3(other, g, t);
    const double f = theta / v_sin;
    res[0] = f * (Bx[0] * t[0] + Bx[2] * t[1] + Bx[4] * t[2]);
    res[1] = f * (Bx[1] * t[0] + Bx[3] * t[1] + Bx[5] * t[2]);
}

// blockwise J application helpers on a row-major N x N (or N x cols) matrix
inline void rows_mul(double* M, int cols, int ld, int idx, int dim, const double* J) {  // M[idx:idx+dim, 0:cols] = J * (...)
    for (int c = 0; c < cols; c++) {
        double v[3];
        for (int a = 0; a < dim; a++) v[a] = M[(idx + a) * ld + c];
        for (int a = 0; a < dim; a++) {
            double s = 0;
            for (int b = 0; b < dim; b++) s += J[a * dim + b] * v[b];
            M[(idx + a) * ld + c] = s;
        }
    }
}
inline void cols_mul_T(double* M, int rows, int ld, int idx, int dim, const double* J) {  // M[:, idx:idx+dim] = (...) * J^T
    for (int r = 0; r < rows; r++) {
        double v[3];
        for (int a = 0; a < dim; a++) v[a] = M[r * ld + idx + a];
        for (int a = 0; a < dim; a++) {
            double s = 0;
            for (int b = 0; b < dim; b++) s += v[b] * J[a * dim + b];
            M[r * ld + idx + a] = s;
        }
    }
}

}  // namespace

void quat_rotate(const double q[4], const double v[3], double out[3]) {  // Eigen _transformVector
    double uv[3], c2[3];
    cross3(q, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    cross3(q, uv, c2);
    for (int i = 0; i < 3; i++) out[i] = (v[i] + q[3] * uv[i]) + c2[i];
}

void so3_A_matrix(const double v[3], double A[9]) {  // mtkmath.hpp:235-247
    const double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double n = sqrt(sq);
    for (int i = 0; i < 9; i++) A[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n < kTol) return;
    double H[9], HH[9];
    hat3(v, H);
    mm3(H, H, HH);
    const double a = (1 - cos(n)) / sq, b = (1 - sin(n) / n) / sq;
    for (int i = 0; i < 9; i++) A[i] += a * H[i] + b * HH[i];
}

void s2_Nx_yy(const double g[3], double Nx[6]) {  // S2.hpp:259-264: 1/len^2 * Bx^T * hat(g)
    double Bx[6], H[9];
    s2_Bx(g, Bx);
    hat3(g, H);
    const double f = 1 / kS2Length / kS2Length;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += Bx[k * 2 + i] * H[k * 3 + j];
            Nx[i * 3 + j] = f * s;
        }
}

void s2_Mx(const double g[3], const double delta[2], double M[6]) {  // S2.hpp:266-280
    double Bx[6], H[9];
    s2_Bx(g, Bx);
    hat3(g, H);
    const double dn = sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
    double L[9];  // the 3x3 that multiplies Bx
    if (dn < kTol) {
        memcpy(L, H, sizeof(L));
    } else {
        const double Bu[3] = {Bx[0] * delta[0] + Bx[1] * delta[1], Bx[2] * delta[0] + Bx[3] * delta[1], Bx[4] * delta[0] + Bx[5] * delta[1]};
        // S2.hpp:277 builds exp(Bu, scalar(1/2)) and 1/2 is integer division -> scale 0 -> identity rotation.
        double e[4], E[9], A[9], At[9], T[9];
        so3_exp(Bu, 0.0, e);
        quat_to_R(e, E);
        so3_A_matrix(Bu, A);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) At[i * 3 + j] = A[j * 3 + i];
        mm3(E, H, T);
        mm3(T, At, L);
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += L[i * 3 + k] * Bx[k * 2 + j];
            M[i * 2 + j] = -s;
        }
}

void state_from_array(const double s[26], LioState& x) {
    for (int i = 0; i < 3; i++) { x.pos[i] = s[i]; x.til[i] = s[11 + i]; x.vel[i] = s[14 + i]; x.bg[i] = s[17 + i]; x.ba[i] = s[20 + i]; x.grav[i] = s[23 + i]; }
    for (int i = 0; i < 4; i++) { x.rot[i] = s[3 + i]; x.ril[i] = s[7 + i]; }
}
void state_to_array(const LioState& x, double s[26]) {
    for (int i = 0; i < 3; i++) { s[i] = x.pos[i]; s[11 + i] = x.til[i]; s[14 + i] = x.vel[i]; s[17 + i] = x.bg[i]; s[20 + i] = x.ba[i]; s[23 + i] = x.grav[i]; }
    for (int i = 0; i < 4; i++) { s[3 + i] = x.rot[i]; s[7 + i] = x.ril[i]; }
}

// DoF layout: pos 0, rot 3, R_il 6, t_il 9, vel 12, bg 15, ba 18, grav 21 (use-ikfom.hpp:12-21)
void state_boxplus(LioState& x, const double d[kDof]) {
    double e[4], q[4];
    for (int i = 0; i < 3; i++) x.pos[i] += d[i];
    so3_exp(d + 3, 0.5, e);
    quat_mul(x.rot, e, q);
    memcpy(x.rot, q, sizeof(q));
    so3_exp(d + 6, 0.5, e);
    quat_mul(x.ril, e, q);
    memcpy(x.ril, q, sizeof(q));
    for (int i = 0; i < 3; i++) { x.til[i] += d[9 + i]; x.vel[i] += d[12 + i]; x.bg[i] += d[15 + i]; x.ba[i] += d[18 + i]; }
    s2_boxplus(x.grav, d + 21);
}
void state_boxminus(const LioState& x, const LioState& o, double d[kDof]) {
    for (int i = 0; i < 3; i++) d[i] = x.pos[i] - o.pos[i];
    double c[4], q[4];
    c[0] = -o.rot[0]; c[1] = -o.rot[1]; c[2] = -o.rot[2]; c[3] = o.rot[3];
    quat_mul(c, x.rot, q);
    so3_log(q, d + 3);
    c[0] = -o.ril[0]; c[1] = -o.ril[1]; c[2] = -o.ril[2]; c[3] = o.ril[3];
    quat_mul(c, x.ril, q);
    so3_log(q, d + 6);
    for (int i = 0; i < 3; i++) { d[9 + i] = x.til[i] - o.til[i]; d[12 + i] = x.vel[i] - o.vel[i]; d[15 + i] = x.bg[i] - o.bg[i]; d[18 + i] = x.ba[i] - o.ba[i]; }
    s2_boxminus(x.grav, o.grav, d + 21);
}

bool mat_inverse(const double* A, int n, double* out) {
    std::vector<double> LU(A, A + (size_t)n * n);
    std::vector<int> piv(n);
    for (int i = 0; i < n; i++) piv[i] = i;
    for (int k = 0; k < n; k++) {
        int p = k;
        double best = fabs(LU[k * n + k]);
        for (int i = k + 1; i < n; i++)
            if (fabs(LU[i * n + k]) > best) { best = fabs(LU[i * n + k]); p = i; }
        if (best == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < n; j++) { const double t = LU[k * n + j]; LU[k * n + j] = LU[p * n + j]; LU[p * n + j] = t; }
            const int t = piv[k]; piv[k] = piv[p]; piv[p] = t;
        }
        for (int i = k + 1; i < n; i++) {
            LU[i * n + k] /= LU[k * n + k];
            const double f = LU[i * n + k];
            for (int j = k + 1; j < n; j++) LU[i * n + j] -= f * LU[k * n + j];
        }
    }
    std::vector<double> y(n);
    for (int col = 0; col < n; col++) {
        for (int i = 0; i < n; i++) {
            double s = (piv[i] == col) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= LU[i * n + j] * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = y[i];
            for (int j = i + 1; j < n; j++) s -= LU[i * n + j] * out[j * n + col];
            out[i * n + col] = s / LU[i * n + i];
        }
    }
    return true;
}

// the reference calls Eigen::SelfAdjointEigenSolver<Matrix3d> (laserMapping.cpp:939-941); only the invariant
// subspaces matter downstream (|n.v| sums and the projector sum_kept v v^T), so any convergent solver serves
void eig3_sym(const double Ain[9], double w[3], double V[9]) {
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

Eskf::Eskf() {
    memset(&x, 0, sizeof(x));
    x.rot[3] = 1.0;
    x.ril[3] = 1.0;
    x.grav[0] = kS2Length;  // MTK S2 default (S2.hpp ctor)
    for (int i = 0; i < N * N; i++) P[i] = (i % (N + 1) == 0) ? 1.0 : 0.0;
    for (int i = 0; i < N; i++) limit[i] = 0.001;  // epsi, laserMapping.cpp:1097,1116
}

void Eskf::begin(Work& w) {
    w.x_prop = x;
    memcpy(w.P_prop, P, sizeof(P));
    memset(w.K_x, 0, sizeof(w.K_x));
    memset(w.K_h, 0, sizeof(w.K_h));
    memset(w.dx_new, 0, sizeof(w.dx_new));
}

// the manifold Jacobians of esekfom.hpp:1661-1699: SO3 blocks at 3 and 6, S2 block at 21
static void manifold_jacobians(const LioState& x, const LioState& x_prop, const double dx[kDof], double J_rot[9], double J_ril[9], double J_g[4]) {
    double A[9];
    so3_A_matrix(dx + 3, A);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) J_rot[a * 3 + b] = A[b * 3 + a];
    so3_A_matrix(dx + 6, A);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) J_ril[a * 3 + b] = A[b * 3 + a];
    double Nx[6], Mx[6];
    s2_Nx_yy(x.grav, Nx);
    s2_Mx(x_prop.grav, dx + 21, Mx);
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += Nx[a * 3 + k] * Mx[k * 2 + b];
            J_g[a * 2 + b] = s;
        }
}

int Eskf::step(Work& w, double R, const Measurement& m, int i, bool& converge, int& t, double dx_[kDof]) {
    double dx[N];
    state_boxminus(x, w.x_prop, dx);
    memcpy(w.dx_new, dx, sizeof(dx));
    memcpy(P, w.P_prop, sizeof(P));
    double Jr[9], Jl[9], Jg[4];
    manifold_jacobians(x, w.x_prop, dx, Jr, Jl, Jg);
    rows_mul(w.dx_new, 1, 1, 3, 3, Jr);
    rows_mul(P, N, N, 3, 3, Jr);
    cols_mul_T(P, N, N, 3, 3, Jr);
    rows_mul(w.dx_new, 1, 1, 6, 3, Jl);
    rows_mul(P, N, N, 6, 3, Jl);
    cols_mul_T(P, N, N, 6, 3, Jl);
    rows_mul(w.dx_new, 1, 1, 21, 2, Jg);
    rows_mul(P, N, N, 21, 2, Jg);
    cols_mul_T(P, N, N, 21, 2, Jg);

    memset(w.K_x, 0, sizeof(w.K_x));
    const int ng = m.ws_n > 0 ? m.n_geo : m.n_rows;  // point-to-plane rows; the wheel-speed triples follow them
    if (N > m.n_rows && (m.rows6 || ng == 0)) {
        // K = P H^T (H P H^T / R + I)^-1 / R with H = [rows6 | 0] (+ unit rows at columns 12..14 per wheel-speed triple)   (esekfom.hpp:1715-1744)
        const int d = m.n_rows;
        std::vector<double> PHt((size_t)N * d), S((size_t)d * d), Si((size_t)d * d), K((size_t)N * d), hv(d);
        for (int r = 0; r < ng; r++) hv[r] = m.h[r];
        for (int r = ng; r < d; r++) hv[r] = m.ws_h[(r - ng) / 3][(r - ng) % 3];
        for (int a = 0; a < N; a++)
            for (int r = 0; r < d; r++) {
                double s = 0;
                if (r < ng) for (int c = 0; c < 6; c++) s += P[a * N + c] * m.rows6[r * 6 + c];
                else s = P[a * N + 12 + (r - ng) % 3];
                PHt[(size_t)a * d + r] = s;
            }
        for (int r = 0; r < d; r++)
            for (int q = 0; q < d; q++) {
                double s = 0;
                if (r < ng) for (int c = 0; c < 6; c++) s += m.rows6[r * 6 + c] * PHt[(size_t)c * d + q];
                else s = PHt[(size_t)(12 + (r - ng) % 3) * d + q];
                S[(size_t)r * d + q] = s / R + (r == q ? 1.0 : 0.0);
            }
        mat_inverse(S.data(), d, Si.data());
        for (int a = 0; a < N; a++)
            for (int q = 0; q < d; q++) {
                double s = 0;
                for (int r = 0; r < d; r++) s += PHt[(size_t)a * d + r] * Si[(size_t)r * d + q];
                K[(size_t)a * d + q] = s / R;
            }
        for (int a = 0; a < N; a++) {
            double s = 0;
            for (int r = 0; r < d; r++) s += K[(size_t)a * d + r] * hv[r];
            w.K_h[a] = s;
            for (int c = 0; c < 6; c++) {
                double v = 0;
                for (int r = 0; r < ng; r++) v += K[(size_t)a * d + r] * m.rows6[r * 6 + c];
                w.K_x[a * N + c] = v;
            }
            for (int j = 0; j < 3 && m.ws_n > 0; j++) {
                double v = 0;
                for (int t2 = 0; t2 < m.ws_n; t2++) v += K[(size_t)a * d + ng + 3 * t2 + j];
                w.K_x[a * N + 12 + j] = v;
            }
        }
    } else {
        // esekfom.hpp:1782-1809:  P_temp = (P/R)^-1;  P_temp[0:15,0:15] += HTH;  P_inv = P_temp^-1;
        //                         K_h = P_inv[:, 0:15] h_x^T h;  K_x[:, 0:15] = P_inv[:, 0:15] HTH.
        // HTH is non-zero only on the index set S = {0..5} (extrinsic_est_en == false; plus {12, 13, 14} with wheel-speed rows, whose block
        // is ws_n I3), so with A = (P/R)^-1, E = the unit columns of S, B = HTH_SS and the matrix inversion lemma,
        //   P_inv E = (A + E B E^T)^-1 E = (P/R) E (I + B (P/R)_SS)^-1 ,
        // which is all K_h and K_x need: one |S| x |S| inverse instead of the reference's two 23x23 ones (and no
        // inverse of the possibly singular, degeneracy-projected B).  Same quantities, better conditioned.
        const int ns = m.ws_n > 0 ? 9 : 6;
        const int idx[9] = {0, 1, 2, 3, 4, 5, 12, 13, 14};
        double B[81], Bh[9];
        for (int a = 0; a < ns; a++) {
            for (int c = 0; c < ns; c++) B[a * ns + c] = (a < 6 && c < 6) ? m.HTH[a * 6 + c] : ((a == c) ? (double)m.ws_n : 0.0);
            Bh[a] = a < 6 ? m.HTh[a] : 0.0;
        }
        for (int t2 = 0; t2 < m.ws_n; t2++)
            for (int j = 0; j < 3; j++) Bh[6 + j] += m.ws_h[t2][j];
        double G[N * 9], Mx[81], Mi[81];
        for (int a = 0; a < N; a++)
            for (int c = 0; c < ns; c++) G[a * ns + c] = P[a * N + idx[c]] / R;
        for (int a = 0; a < ns; a++)
            for (int c = 0; c < ns; c++) {
                double v = (a == c) ? 1.0 : 0.0;
                for (int k = 0; k < ns; k++) v += B[a * ns + k] * G[idx[k] * ns + c];
                Mx[a * ns + c] = v;
            }
        mat_inverse(Mx, ns, Mi);
        double Pi[N * 9];  // P_inv[:, S]
        for (int a = 0; a < N; a++)
            for (int c = 0; c < ns; c++) {
                double v = 0;
                for (int k = 0; k < ns; k++) v += G[a * ns + k] * Mi[k * ns + c];
                Pi[a * ns + c] = v;
            }
        for (int a = 0; a < N; a++) {
            double s = 0;
            for (int c = 0; c < ns; c++) s += Pi[a * ns + c] * Bh[c];
            w.K_h[a] = s;
            for (int bcol = 0; bcol < ns; bcol++) {
                double v = 0;
                for (int c = 0; c < ns; c++) v += Pi[a * ns + c] * B[c * ns + bcol];
                w.K_x[a * N + idx[bcol]] = v;
            }
        }
    }
    // dx_ = K_h + (K_x - I) dx_new   (esekfom.hpp:1815)
    for (int a = 0; a < N; a++) {
        double s = w.K_h[a];
        for (int b = 0; b < N; b++) s += (w.K_x[a * N + b] - (a == b ? 1.0 : 0.0)) * w.dx_new[b];
        dx_[a] = s;
    }
    state_boxplus(x, dx_);
    converge = true;
    for (int a = 0; a < N; a++)
        if (fabs(dx_[a]) > limit[a]) { converge = false; break; }
    if (converge) t++;
    if (!t && i == maximum_iter - 2) converge = true;  // force one more neighbour search on the last pass
    if (t > 1 || i == maximum_iter - 1) {
        // P = L - K_x[:, 0:15] P[0:15, :] with the manifold Jacobians rebuilt from dx_ (esekfom.hpp:1836-1924)
        double L[N * N];
        manifold_jacobians(x, w.x_prop, dx_, Jr, Jl, Jg);
        memcpy(L, P, sizeof(L));
        const struct { int idx, dim; const double* J; } blk[3] = {{3, 3, Jr}, {6, 3, Jl}, {21, 2, Jg}};
        for (int k = 0; k < 3; k++) {
            // L rows come from P's rows (P itself is only column-transformed)
            for (int c = 0; c < N; c++)
                for (int a = 0; a < blk[k].dim; a++) {
                    double s = 0;
                    for (int b = 0; b < blk[k].dim; b++) s += blk[k].J[a * blk[k].dim + b] * P[(blk[k].idx + b) * N + c];
                    L[(blk[k].idx + a) * N + c] = s;
                }
            rows_mul(w.K_x, 15, N, blk[k].idx, blk[k].dim, blk[k].J);
            cols_mul_T(L, N, N, blk[k].idx, blk[k].dim, blk[k].J);
            cols_mul_T(P, N, N, blk[k].idx, blk[k].dim, blk[k].J);
        }
        double Pn[N * N];
        for (int a = 0; a < N; a++)
            for (int b = 0; b < N; b++) {
                double s = 0;
                for (int k = 0; k < 15; k++) s += w.K_x[a * N + k] * P[k * N + b];
                Pn[a * N + b] = L[a * N + b] - s;
            }
        memcpy(P, Pn, sizeof(P));
        return 1;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// predict.  F_x1 = I + f_x_final * dt has ~40 entries off the identity and W = dt * f_w_final ~24, so the two
// 23^3 products of the reference are done row-sparse: the non-zero pattern is read off the dense matrices
// (built exactly as esekfom.hpp builds them) and the sums skip exact zeros in the original k order, which
// leaves every result bit unchanged.
// ---------------------------------------------------------------------------------------------------------
void Eskf::predict(double dt, const double Q[12], const double acc[3], const double gyro[3]) {
    constexpr int n = kDof;
    // get_f: 24 entries, only pos / rot / vel rows are non-zero
    double f[24] = {0};
    const double amb[3] = {acc[0] - x.ba[0], acc[1] - x.ba[1], acc[2] - x.ba[2]};
    double a_in[3];
    quat_rotate(x.rot, amb, a_in);
    for (int i = 0; i < 3; i++) { f[i] = x.vel[i]; f[3 + i] = gyro[i] - x.bg[i]; f[12 + i] = a_in[i] + x.grav[i]; }
    // df_dx (24 x 23), df_dw (24 x 12)
    static thread_local double fx[24 * n], fw[24 * 12], fxf[n * n], fwf[n * 12], F1[n * n], FP[n * n], Pn[n * n];
    memset(fx, 0, sizeof(fx));
    memset(fw, 0, sizeof(fw));
    double R[9], H[9], RH[9];
    quat_to_R(x.rot, R);
    hat3(amb, H);
    mm3(R, H, RH);
    for (int i = 0; i < 3; i++) {
        fx[i * n + 12 + i] = 1.0;
        fx[(3 + i) * n + 15 + i] = -1.0;
        for (int j = 0; j < 3; j++) {
            fx[(12 + i) * n + 3 + j] = -RH[i * 3 + j];
            fx[(12 + i) * n + 18 + j] = -R[i * 3 + j];
            fw[(12 + i) * 12 + 3 + j] = -R[i * 3 + j];
        }
        fw[(3 + i) * 12 + i] = -1.0;
        fw[(15 + i) * 12 + 6 + i] = 1.0;
        fw[(18 + i) * 12 + 9 + i] = 1.0;
    }
    const double zero2[2] = {0, 0};
    {
        double Mx[6];
        s2_Mx(x.grav, zero2, Mx);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 2; j++) fx[(12 + i) * n + 21 + j] = Mx[i * 2 + j];
    }
    const LioState x_before = x;
    // x_.oplus(f_, dt): vect += dt * f, SO3 *= exp(f, dt / 2 inside mtk exp), S2 with f == 0 stays
    for (int i = 0; i < 3; i++) {
        x.pos[i] += dt * f[i]; x.til[i] += dt * f[9 + i]; x.vel[i] += dt * f[12 + i]; x.bg[i] += dt * f[15 + i]; x.ba[i] += dt * f[18 + i];
    }
    {
        double e[4], q[4];
        so3_exp(f + 3, dt / 2, e);
        quat_mul(x.rot, e, q);
        memcpy(x.rot, q, sizeof(q));
        so3_exp(f + 6, dt / 2, e);
        quat_mul(x.ril, e, q);
        memcpy(x.ril, q, sizeof(q));
    }
    memset(fxf, 0, sizeof(fxf));
    memset(fwf, 0, sizeof(fwf));
    for (int i = 0; i < n * n; i++) F1[i] = 0;
    for (int i = 0; i < n; i++) F1[i * n + i] = 1.0;
    static const int vect_idx[5] = {0, 9, 12, 15, 18};
    for (int v = 0; v < 5; v++)
        for (int j = 0; j < 3; j++) {
            memcpy(fxf + (vect_idx[v] + j) * n, fx + (vect_idx[v] + j) * n, sizeof(double) * n);
            memcpy(fwf + (vect_idx[v] + j) * 12, fw + (vect_idx[v] + j) * 12, sizeof(double) * 12);
        }
    for (int idx = 3; idx <= 6; idx += 3) {  // SO3 states: F_x1 block = exp(seg, scalar(1/2) == 0) = identity (esekfom.hpp:312)
        const double seg[3] = {-f[idx] * dt, -f[idx + 1] * dt, -f[idx + 2] * dt};
        double A[9];
        so3_A_matrix(seg, A);
        for (int i = 0; i < n; i++) {
            const double v[3] = {fx[idx * n + i], fx[(idx + 1) * n + i], fx[(idx + 2) * n + i]};
            for (int a = 0; a < 3; a++) fxf[(idx + a) * n + i] = A[a * 3] * v[0] + A[a * 3 + 1] * v[1] + A[a * 3 + 2] * v[2];
        }
        for (int i = 0; i < 12; i++) {
            const double v[3] = {fw[idx * 12 + i], fw[(idx + 1) * 12 + i], fw[(idx + 2) * 12 + i]};
            for (int a = 0; a < 3; a++) fwf[(idx + a) * 12 + i] = A[a * 3] * v[0] + A[a * 3 + 1] * v[1] + A[a * 3 + 2] * v[2];
        }
    }
    {  // S2 state: idx 21 (dof), dim 21
        const double seg[3] = {f[21] * dt, f[22] * dt, f[23] * dt};
        double Nx[6], Mx[6], Hb[9], A[9], T[9], res23[6];
        s2_Nx_yy(x.grav, Nx);
        s2_Mx(x_before.grav, zero2, Mx);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                double s2 = 0;
                for (int k = 0; k < 3; k++) s2 += Nx[a * 3 + k] * Mx[k * 2 + b];
                F1[(21 + a) * n + 21 + b] = s2;
            }
        hat3(x_before.grav, Hb);
        so3_A_matrix(seg, A);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s2 = 0;
                for (int k = 0; k < 3; k++) s2 += Hb[i * 3 + k] * A[j * 3 + k];
                T[i * 3 + j] = s2;
            }
        for (int a = 0; a < 2; a++)
            for (int j = 0; j < 3; j++) {
                double s2 = 0;
                for (int k = 0; k < 3; k++) s2 += Nx[a * 3 + k] * T[k * 3 + j];
                res23[a * 3 + j] = -s2;
            }
        for (int i = 0; i < n; i++) {
            const double v[3] = {fx[21 * n + i], fx[22 * n + i], fx[23 * n + i]};
            for (int a = 0; a < 2; a++) fxf[(21 + a) * n + i] = res23[a * 3] * v[0] + res23[a * 3 + 1] * v[1] + res23[a * 3 + 2] * v[2];
        }
        for (int i = 0; i < 12; i++) {
            const double v[3] = {fw[21 * 12 + i], fw[22 * 12 + i], fw[23 * 12 + i]};
            for (int a = 0; a < 2; a++) fwf[(21 + a) * 12 + i] = res23[a * 3] * v[0] + res23[a * 3 + 1] * v[1] + res23[a * 3 + 2] * v[2];
        }
    }
    for (int i = 0; i < n * n; i++) F1[i] += fxf[i] * dt;
    // row-sparse patterns
    int nzc[n][n], nzn[n], wzc[n][12], wzn[n];
    for (int i = 0; i < n; i++) {
        nzn[i] = 0;
        for (int k = 0; k < n; k++)
            if (F1[i * n + k] != 0.0) nzc[i][nzn[i]++] = k;
        wzn[i] = 0;
        for (int k = 0; k < 12; k++) {
            fwf[i * 12 + k] = dt * fwf[i * 12 + k];
            if (fwf[i * 12 + k] != 0.0) wzc[i][wzn[i]++] = k;
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s2 = 0;
            for (int c = 0; c < nzn[i]; c++) { const int k = nzc[i][c]; s2 += F1[i * n + k] * P[k * n + j]; }
            FP[i * n + j] = s2;
        }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s2 = 0;
            for (int c = 0; c < nzn[j]; c++) { const int k = nzc[j][c]; s2 += FP[i * n + k] * F1[j * n + k]; }
            double q = 0;
            if (wzn[i] && wzn[j])
                for (int c = 0; c < wzn[i]; c++) { const int k = wzc[i][c]; q += fwf[i * 12 + k] * Q[k] * fwf[j * 12 + k]; }
            Pn[i * n + j] = s2 + q;
        }
    memcpy(P, Pn, sizeof(Pn));
}

}  // namespace lio

// ---------------------------------------------------------------------------------------------------------------------
// Host-only: the DEVICE-resident filter loop (eskf_dev.h, here compiled for the host: plain loops instead of workgroup-strided
// ones) driven by caller-supplied sums -- what step_kernel (p2plane.hip) does once per pass on the GPU.  tests/test_eskf_dev.py
// holds it against lio_eskf_update_cb (the host filter, itself pinned to the reference's IKFoM code).
extern "C" int lio_eskf_update_sums_cb(const double s26[26], const double P[529], double R, int max_iter, int degenerate_detect_en,
                                       lio_sums_fn fn, lio_degeneracy_fn dfn, void* ctx, double s26_out[26], double P_out[529],
                                       lio_pass_log* logs, int cap_logs, int* status) {
    using namespace lio;
    if (!s26 || !P || !fn || !s26_out || !P_out || max_iter < 0 || max_iter + 1 > kEkMaxPass) return LIO_E_INVALID;
    static_assert(sizeof(EkPassLog) == sizeof(lio_pass_log), "EkPassLog mirrors lio_pass_log");
    EskfDev* c = new EskfDev();
    EkWork* w = new EkWork();
    memset(c, 0, sizeof(*c));
    memset(w, 0, sizeof(*w));
    memcpy(c->x, s26, sizeof(double) * 26);
    memcpy(c->P, P, sizeof(double) * 529);
    for (int k = 0; k < kEkN; k++) c->limit[k] = 0.001;
    c->R = R;
    c->maximum_iter = max_iter;
    c->degenerate_detect_en = degenerate_detect_en;
    ek_begin(*c);
    while (c->status == EK_RUNNING) {
        const int knn = c->converge;
        double acc[29];
        for (int k = 0; k < 29; k++) acc[k] = 0.0;
        if (!fn(ctx, c->x, knn, acc)) acc[28] = 0.0;
        ek_measure_head(*c, *w, acc, knn);
        if (w->flag[1] && dfn) dfn(ctx, w->eigvec, w->cs);
        ek_measure_tail(*c, *w);
        if (c->status != EK_RUNNING) break;
        if (w->flag[0]) ek_step(*c, *w);
    }
    memcpy(s26_out, c->x, sizeof(double) * 26);
    memcpy(P_out, c->P, sizeof(double) * 529);
    if (status) *status = c->status;
    const int n = c->n_log;
    if (logs)
        for (int k = 0; k < n && k < cap_logs; k++) memcpy(&logs[k], &c->log[k], sizeof(lio_pass_log));
    delete c;
    delete w;
    return n;
}
