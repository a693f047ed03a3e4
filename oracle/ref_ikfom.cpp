// oracle/ref_ikfom.cpp -- the reference's OWN error-state Kalman filter (IKFoM: esekfom.hpp predict and
// update_iterated_dyn_share_modified, MTK's SO3 / S2 / vect types, use-ikfom.hpp's process model get_f / df_dx / df_dw /
// process_noise_cov) compiled from where it lies under /root/reference, behind a C ABI.  Nothing is copied.
//
// Boost is not installed.  MTK needs it in two places: boost::math::tools::epsilon (shimmed: numeric_limits) and
// Boost.Preprocessor inside MTK_BUILD_MANIFOLD.  The macro is replaced here by what it expands to for the three manifolds
// of use-ikfom.hpp:12-33 -- members MTK::SubManifold<type, idx, dim> in declaration order, DOF / DIM enums, and the
// member functions assembled from the reference's own per-entry macros (MTK_BOXPLUS, MTK_OPLUS, MTK_BOXMINUS,
// MTK_S2_hat, ...: build_manifold.hpp:99-113).  All arithmetic is the reference's.
// TEST INFRASTRUCTURE ONLY: built into oracle/_ref/libref_ikfom.so by `make -C oracle ref`.
#include "ref_shims/lsd_ikfom_manifolds.h"

#include <cstring>

typedef esekfom::esekf<state_ikfom, 12, input_ikfom> Kf;

static void to_state(const double* s, state_ikfom& x) {  // 26 doubles: pos3 rot4(xyzw) ril4 til3 vel3 bg3 ba3 grav3
    x.pos = vect3(Eigen::Vector3d(s[0], s[1], s[2]));
    x.rot.coeffs() = Eigen::Vector4d(s[3], s[4], s[5], s[6]);
    x.offset_R_L_I.coeffs() = Eigen::Vector4d(s[7], s[8], s[9], s[10]);
    x.offset_T_L_I = vect3(Eigen::Vector3d(s[11], s[12], s[13]));
    x.vel = vect3(Eigen::Vector3d(s[14], s[15], s[16]));
    x.bg = vect3(Eigen::Vector3d(s[17], s[18], s[19]));
    x.ba = vect3(Eigen::Vector3d(s[20], s[21], s[22]));
    x.grav.vec = Eigen::Vector3d(s[23], s[24], s[25]);
}
static void from_state(const state_ikfom& x, double* s) {
    for (int i = 0; i < 3; i++) { s[i] = x.pos[i]; s[11 + i] = x.offset_T_L_I[i]; s[14 + i] = x.vel[i]; s[17 + i] = x.bg[i]; s[20 + i] = x.ba[i]; s[23 + i] = x.grav.vec[i]; }
    for (int i = 0; i < 4; i++) { s[3 + i] = x.rot.coeffs()[i]; s[7 + i] = x.offset_R_L_I.coeffs()[i]; }
}

// measurement model driven from outside: fn(ctx, state26, converge, &n, rows (n x 6, first six columns of h_x), h (n)) -> valid
typedef int (*meas_fn)(void* ctx, const double* s26, int converge, int* n, double* rows6, double* h, int cap);
static meas_fn g_fn = nullptr;
static void* g_ctx = nullptr;
static int g_cap = 0;
static std::vector<double> g_rows, g_h;
static void h_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d) {
    double s26[26];
    from_state(s, s26);
    int n = 0;
    const int valid = g_fn(g_ctx, s26, d.converge ? 1 : 0, &n, g_rows.data(), g_h.data(), g_cap);
    if (!valid) { d.valid = false; return; }
    d.h_x = Eigen::MatrixXd::Zero(n, 15);
    d.h.resize(n);
    for (int r = 0; r < n; r++) {
        for (int c = 0; c < 6; c++) d.h_x(r, c) = g_rows[(size_t)r * 6 + c];
        d.h(r) = g_h[r];
    }
}

extern "C" {

void ref_kf_predict(const double* s26, const double* P, double dt, const double* Qdiag12, const double* acc, const double* gyro,
                    double* s26_out, double* P_out) {
    Kf kf;
    double epsi[23];
    std::fill(epsi, epsi + 23, 0.001);
    kf.init_dyn_share(get_f, df_dx, df_dw, h_model, 4, epsi);
    state_ikfom x;
    to_state(s26, x);
    kf.change_x(x);
    Kf::cov Pm;
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) Pm(i, j) = P[i * 23 + j];
    kf.change_P(Pm);
    Eigen::Matrix<double, 12, 12> Q = process_noise_cov();
    for (int i = 0; i < 12; i++) Q(i, i) = Qdiag12[i];
    input_ikfom in;
    in.acc = vect3(Eigen::Vector3d(acc[0], acc[1], acc[2]));
    in.gyro = vect3(Eigen::Vector3d(gyro[0], gyro[1], gyro[2]));
    kf.predict(dt, Q, in);
    from_state(kf.get_x(), s26_out);
    const Kf::cov& Po = kf.get_P();
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) P_out[i * 23 + j] = Po(i, j);
}

void ref_process_noise_cov(double* Q144) {
    Eigen::Matrix<double, 12, 12> Q = process_noise_cov();
    for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) Q144[i * 12 + j] = Q(i, j);
}

void ref_kf_update(const double* s26, const double* P, double R, int max_iter, meas_fn fn, void* ctx, int cap, double* s26_out, double* P_out) {
    Kf kf;
    double epsi[23];
    std::fill(epsi, epsi + 23, 0.001);
    kf.init_dyn_share(get_f, df_dx, df_dw, h_model, max_iter, epsi);
    state_ikfom x;
    to_state(s26, x);
    kf.change_x(x);
    Kf::cov Pm;
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) Pm(i, j) = P[i * 23 + j];
    kf.change_P(Pm);
    g_fn = fn; g_ctx = ctx; g_cap = cap;
    g_rows.assign((size_t)cap * 6, 0.0);
    g_h.assign(cap, 0.0);
    double solve_time = 0;
    kf.update_iterated_dyn_share_modified(R, solve_time);
    from_state(kf.get_x(), s26_out);
    const Kf::cov& Po = kf.get_P();
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) P_out[i * 23 + j] = Po(i, j);
}

void ref_state_boxplus(const double* s26, const double* d23, double* out26) {
    state_ikfom x;
    to_state(s26, x);
    Eigen::Matrix<double, 23, 1> d;
    for (int i = 0; i < 23; i++) d(i) = d23[i];
    x.boxplus(d);
    from_state(x, out26);
}
void ref_state_boxminus(const double* a26, const double* b26, double* d23) {
    state_ikfom a, b;
    to_state(a26, a);
    to_state(b26, b);
    Eigen::Matrix<double, 23, 1> d;
    a.boxminus(d, b);
    for (int i = 0; i < 23; i++) d23[i] = d(i);
}

}  // extern "C"
