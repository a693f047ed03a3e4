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
#include <IKFoM_toolkit/esekfom/esekfom.hpp>

#undef MTK_BUILD_MANIFOLD
#define MTK_BUILD_MANIFOLD(name, entries) LSD_MANIFOLD_##name

#define LSD_MANIFOLD_BODY(name, ENTRIES)                                                                                   \
    int getDOF() const { return DOF; }                                                                                      \
    void boxplus(const MTK::vectview<const scalar, DOF>& __vec, scalar __scale = 1) { ENTRIES(MTK_BOXPLUS) }                \
    void oplus(const MTK::vectview<const scalar, DIM>& __vec, scalar __scale = 1) { ENTRIES(MTK_OPLUS) }                    \
    void boxminus(MTK::vectview<scalar, DOF> __res, const name& __oth) const { ENTRIES(MTK_BOXMINUS) }                      \
    friend std::ostream& operator<<(std::ostream& __os, const name& __var) { return __os ENTRIES(MTK_OSTREAM); }            \
    void build_S2_state() { ENTRIES(MTK_S2_state) }                                                                         \
    void build_vect_state() { ENTRIES(MTK_vect_state) }                                                                     \
    void build_SO3_state() { ENTRIES(MTK_SO3_state) }                                                                       \
    void S2_hat(Eigen::Matrix<scalar, 3, 3>& res, int idx) { ENTRIES(MTK_S2_hat) }                                          \
    void S2_Nx_yy(Eigen::Matrix<scalar, 2, 3>& res, int idx) { ENTRIES(MTK_S2_Nx_yy) }                                      \
    void S2_Mx(Eigen::Matrix<scalar, 3, 2>& res, Eigen::Matrix<scalar, 2, 1> dx, int idx) { ENTRIES(MTK_S2_Mx) }            \
    friend std::istream& operator>>(std::istream& __is, name& __var) { return __is ENTRIES(MTK_ISTREAM); }

#define LSD_STATE_ENTRIES(M) \
    M(vect3, pos) M(SO3, rot) M(SO3, offset_R_L_I) M(vect3, offset_T_L_I) M(vect3, vel) M(vect3, bg) M(vect3, ba) M(S2, grav)
#define LSD_MANIFOLD_state_ikfom                                                                                           \
    struct state_ikfom {                                                                                                    \
        typedef state_ikfom self;                                                                                           \
        std::vector<std::pair<int, int> > S2_state;                                                                         \
        std::vector<std::pair<int, int> > SO3_state;                                                                        \
        std::vector<std::pair<std::pair<int, int>, int> > vect_state;                                                       \
        MTK::SubManifold<vect3, 0, 0> pos;                                                                                  \
        MTK::SubManifold<SO3, 3, 3> rot;                                                                                    \
        MTK::SubManifold<SO3, 6, 6> offset_R_L_I;                                                                           \
        MTK::SubManifold<vect3, 9, 9> offset_T_L_I;                                                                         \
        MTK::SubManifold<vect3, 12, 12> vel;                                                                                \
        MTK::SubManifold<vect3, 15, 15> bg;                                                                                 \
        MTK::SubManifold<vect3, 18, 18> ba;                                                                                 \
        MTK::SubManifold<S2, 21, 21> grav;                                                                                  \
        enum { DOF = S2::DOF + 21 };                                                                                        \
        enum { DIM = S2::DIM + 21 };                                                                                        \
        typedef S2::scalar scalar;                                                                                          \
        state_ikfom(const vect3& pos = vect3(), const SO3& rot = SO3(), const SO3& offset_R_L_I = SO3(),                    \
                    const vect3& offset_T_L_I = vect3(), const vect3& vel = vect3(), const vect3& bg = vect3(),             \
                    const vect3& ba = vect3(), const S2& grav = S2())                                                       \
            : pos(pos), rot(rot), offset_R_L_I(offset_R_L_I), offset_T_L_I(offset_T_L_I), vel(vel), bg(bg), ba(ba), grav(grav) {} \
        LSD_MANIFOLD_BODY(state_ikfom, LSD_STATE_ENTRIES)                                                                   \
    }

#define LSD_INPUT_ENTRIES(M) M(vect3, acc) M(vect3, gyro)
#define LSD_MANIFOLD_input_ikfom                                                                                           \
    struct input_ikfom {                                                                                                    \
        typedef input_ikfom self;                                                                                           \
        std::vector<std::pair<int, int> > S2_state;                                                                         \
        std::vector<std::pair<int, int> > SO3_state;                                                                        \
        std::vector<std::pair<std::pair<int, int>, int> > vect_state;                                                       \
        MTK::SubManifold<vect3, 0, 0> acc;                                                                                  \
        MTK::SubManifold<vect3, 3, 3> gyro;                                                                                 \
        enum { DOF = vect3::DOF + 3 };                                                                                      \
        enum { DIM = vect3::DIM + 3 };                                                                                      \
        typedef vect3::scalar scalar;                                                                                       \
        input_ikfom(const vect3& acc = vect3(), const vect3& gyro = vect3()) : acc(acc), gyro(gyro) {}                      \
        LSD_MANIFOLD_BODY(input_ikfom, LSD_INPUT_ENTRIES)                                                                   \
    }

#define LSD_NOISE_ENTRIES(M) M(vect3, ng) M(vect3, na) M(vect3, nbg) M(vect3, nba)
#define LSD_MANIFOLD_process_noise_ikfom                                                                                   \
    struct process_noise_ikfom {                                                                                            \
        typedef process_noise_ikfom self;                                                                                   \
        std::vector<std::pair<int, int> > S2_state;                                                                         \
        std::vector<std::pair<int, int> > SO3_state;                                                                        \
        std::vector<std::pair<std::pair<int, int>, int> > vect_state;                                                       \
        MTK::SubManifold<vect3, 0, 0> ng;                                                                                   \
        MTK::SubManifold<vect3, 3, 3> na;                                                                                   \
        MTK::SubManifold<vect3, 6, 6> nbg;                                                                                  \
        MTK::SubManifold<vect3, 9, 9> nba;                                                                                  \
        enum { DOF = vect3::DOF + 9 };                                                                                      \
        enum { DIM = vect3::DIM + 9 };                                                                                      \
        typedef vect3::scalar scalar;                                                                                       \
        process_noise_ikfom(const vect3& ng = vect3(), const vect3& na = vect3(), const vect3& nbg = vect3(),               \
                            const vect3& nba = vect3())                                                                     \
            : ng(ng), na(na), nbg(nbg), nba(nba) {}                                                                         \
        LSD_MANIFOLD_BODY(process_noise_ikfom, LSD_NOISE_ENTRIES)                                                           \
    }

#include <use-ikfom.hpp>

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
