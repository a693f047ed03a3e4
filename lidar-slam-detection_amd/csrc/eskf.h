// eskf.h -- host-side iterated error-state Kalman filter on the FastLIO manifold
// (pos, SO3 rot, SO3 R_il, t_il, vel, bg, ba, S2 grav): 24 stored numbers per block layout below, 23 DoF.
//
// Restates, for the one filter instance the reference uses, what IKFoM generates from
// MTK_BUILD_MANIFOLD(state_ikfom, ...) (/root/reference/slam/mapping/fastlio/include/use-ikfom.hpp:12-21):
//   boxplus / boxminus ........ mtk/types/SOn.hpp:233-245,284-297, mtk/types/S2.hpp:136-167, vect.hpp
//   A_matrix, exp, log ........ mtk/src/mtkmath.hpp:142-174,235-288
//   S2 Bx / Nx_yy / Mx ........ mtk/types/S2.hpp:179-197,259-280
//   iterated update ........... esekfom/esekfom.hpp:1619-1931 (update_iterated_dyn_share_modified)
//   forward propagation ....... esekfom/esekfom.hpp:279-383 (predict), use-ikfom.hpp:47-88 (process model)
// The update consumes the 6x6 / 6 normal equations that the device reduction produces instead of an
// N x 15 Jacobian: with extrinsic_est_en == false (laserMapping.cpp:82) columns 6..14 of h_x are zero, so
// h_x^T h_x and h_x^T h are exactly those blocks.
#pragma once
#include <stdint.h>

namespace lio {

constexpr int kDof = 23;
constexpr double kS2Length = 98090.0 / 10000.0;  // S2<double, 98090, 10000, 1>, use-ikfom.hpp:8

struct LioState {
    double pos[3];
    double rot[4];  // quaternion (x, y, z, w)
    double ril[4];
    double til[3];
    double vel[3], bg[3], ba[3];
    double grav[3];
};

void state_from_array(const double s26[26], LioState& x);
void state_to_array(const LioState& x, double s26[26]);
void state_boxplus(LioState& x, const double d[kDof]);
void state_boxminus(const LioState& x, const LioState& other, double d[kDof]);
void so3_A_matrix(const double v[3], double A[9]);
void s2_Nx_yy(const double g[3], double N[6]);                    // 2 x 3
void s2_Mx(const double g[3], const double delta[2], double M[6]);  // 3 x 2
void quat_rotate(const double q[4], const double v[3], double out[3]);
bool mat_inverse(const double* A, int n, double* out);  // LU with partial pivoting
// symmetric 3x3 eigen-decomposition (cyclic Jacobi): eigenvalues ascending, eigenvectors as columns of V (row-major)
void eig3_sym(const double A[9], double w[3], double V[9]);

// measurement of one pass as the filter sees it
struct Measurement {
    bool valid;
    int n_rows;            // dof_Measurement
    double HTH[36];        // top-left 6 x 6 of h_x^T h_x
    double HTh[6];         // first 6 entries of h_x^T h
    const double* rows6;   // n_geo x 6, only when n_rows < 23 (dense branch, esekfom.hpp:1715-1744)
    const double* h;       // n_geo
    // wheel-speed rows (laserMapping.cpp:794-811, 994-1012): ws_n triples of rows dh/dv = I3 (state columns 12..14) appended after the
    // n_geo point-to-plane rows, ws_h their (weighted) residuals; n_rows = n_geo + 3 ws_n.  ws_n == 0 unless the engine enables them.
    int n_geo = 0;
    int ws_n = 0;
    double ws_h[6][3];
};

struct Eskf {
    LioState x;
    double P[kDof * kDof];
    double limit[kDof];
    int maximum_iter = 4;  // fastlio_init: NUM_MAX_ITERATIONS 4 (laserMapping.cpp:1026,1116)

    Eskf();
    struct Work {
        LioState x_prop;
        double P_prop[kDof * kDof];
        double K_x[kDof * kDof];
        double K_h[kDof];
        double dx_new[kDof];
    };
    // esekf::predict (esekfom.hpp:279-383) with get_f / df_dx / df_dw of use-ikfom.hpp:47-88.  Q is the diagonal of the
    // 12 x 12 process noise (ng, na, nbg, nba).  acc in m/s^2, gyro in rad/s.
    void predict(double dt, const double Q[12], const double acc[3], const double gyro[3]);
    // one call of update_iterated_dyn_share_modified; `measure(x, converge, m)` plays h_dyn_share.
    // Returns the number of measurement evaluations made.
    template <typename F>
    int update_iterated(double R, F&& measure, void (*on_pass)(void*, int, bool, const Measurement&, const double*), void* ctx);
    // the same loop entered at pass `i` with the loop state (converge, t) and the propagated state / covariance in `w` as an earlier
    // part of the update left them (the device-resident loop hands over here when a pass needs the N_eff < 23 dense branch)
    template <typename F>
    int update_iterated_from(Work& w, int i, bool converge, int t, double R, F&& measure,
                             void (*on_pass)(void*, int, bool, const Measurement&, const double*), void* ctx);

    int step(Work& w, double R, const Measurement& m, int i, bool& converge, int& t, double dx_out[kDof]);
    void begin(Work& w);
};

template <typename F>
int Eskf::update_iterated(double R, F&& measure, void (*on_pass)(void*, int, bool, const Measurement&, const double*), void* ctx) {
    Work w;
    begin(w);
    return update_iterated_from(w, -1, true, 0, R, measure, on_pass, ctx);
}

template <typename F>
int Eskf::update_iterated_from(Work& w, int i0, bool converge, int t, double R, F&& measure,
                               void (*on_pass)(void*, int, bool, const Measurement&, const double*), void* ctx) {
    int evals = 0;
    for (int i = i0; i < maximum_iter; i++) {
        Measurement m;
        m.valid = true;
        m.rows6 = nullptr;
        m.h = nullptr;
        const bool knn = converge;
        measure(x, converge, m);
        evals++;
        if (!m.valid) {
            if (on_pass) on_pass(ctx, i, knn, m, nullptr);
            continue;
        }
        double dx[kDof];
        const int done = step(w, R, m, i, converge, t, dx);
        if (on_pass) on_pass(ctx, i, knn, m, dx);
        if (done) break;
    }
    return evals;
}

}  // namespace lio
