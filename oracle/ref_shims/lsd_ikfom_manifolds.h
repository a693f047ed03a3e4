// oracle/ref_shims/lsd_ikfom_manifolds.h -- shared by ref_ikfom.cpp and ref_fastlio.cpp.
// Boost.Preprocessor is not installed; MTK_BUILD_MANIFOLD is replaced by what it expands to for the three manifolds of
// use-ikfom.hpp:12-33 -- members MTK::SubManifold<type, idx, dim> in declaration order, DOF / DIM enums, and the member
// functions assembled from the reference's own per-entry macros (MTK_BOXPLUS, MTK_OPLUS, MTK_BOXMINUS, MTK_S2_hat, ...:
// build_manifold.hpp:99-113).  All arithmetic is the reference's.  Include BEFORE anything that pulls in use-ikfom.hpp.
#pragma once
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
