// engine.hip -- the per-scan driver: what fastlio_main() does after p_imu->Process
// (/root/reference/slam/mapping/fastlio/src/laserMapping.cpp:1189-1304), as host C++ over the kernel-level
// C ABI.  Constants follow fastlio_init (laserMapping.cpp:1025-1124): 4 (+1) filter passes, leaf 0.5 m for
// both filters, INIT_TIME 0.1 s, LASER_POINT_COV 0.001, degeneracy detection on, extrinsic estimation off.
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <array>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "eskf.h"
#include "lio_common.h"

using namespace lio;

// ---- IMU front half of fastlio_main: the buffers of laserMapping.cpp:397-415,311-330,445-520 and the members of
// ImuProcess (IMU_Processing.hpp:31-85) ----
struct ImuSample {
    double stamp;
    double acc[3], gyr[3];  // acc in units of g (fastlio_imu_enqueue divides by 9.81)
};
struct PinnedScan {  // one staging slot: pinned host buffer (xyzi then stamps), its device twin and the copy-done event
    float4* xyzi = nullptr;
    uint32_t* stamp = nullptr;
    float4* d_xyzi = nullptr;
    uint32_t* d_stamp = nullptr;
    hipEvent_t copied = nullptr;
};
struct PendingScan {
    double beg = 0;
    uint32_t n = 0;
    const float4* d_xyzi = nullptr;   // device-resident input (caller keeps it alive until fastlio_main consumed it)
    const uint32_t* d_stamp = nullptr;
    int pinned = -1;                  // index into Frontend::pool for host input
};
struct lio_held_scan { PendingScan sc; };  // sequence batch: the scan between the two halves of fastlio_main
struct Frontend {
    std::mutex mtx;  // mtx_buffer: the enqueue calls come from sensor threads
    std::deque<ImuSample> imu_buffer;
    std::deque<PendingScan> lidar_buffer;
    std::deque<std::pair<double, std::array<double, 3>>> ins_buffer;  // (stamp s, velocity in the IMU frame)
    std::vector<PinnedScan> pool;
    std::vector<int> pool_free;
    int staged = -1;  // slot handed out by lio_fastlio_pcl_stage and not yet committed
    // fastlio_init arguments / Preprocess members
    double scan_period = 0.1, blind = 0.1;
    int filter_num = 1, max_point_num = -1;
    bool undistort = true;
    double Lidar_T[3] = {0, 0, 0}, Lidar_R[4] = {0, 0, 0, 1};
    // ImuProcess members
    bool b_first_frame = true, imu_need_init = true, state_init_done = false;
    int init_iter_num = 1;
    double mean_acc[3] = {0, 0, -1.0}, mean_gyr[3] = {0, 0, 0};
    double cov_acc[3] = {0.1, 0.1, 0.1}, cov_gyr[3] = {0.1, 0.1, 0.1};
    double cov_acc_scale[3] = {0.1, 0.1, 0.1}, cov_gyr_scale[3] = {0.1, 0.1, 0.1};
    double cov_bias_gyr[3] = {0.0001, 0.0001, 0.0001}, cov_bias_acc[3] = {0.0001, 0.0001, 0.0001};
    double vel_last[3] = {0, 0, 0}, angvel_last[3] = {0, 0, 0}, acc_s_last[3] = {0, 0, 0};
    double mean_acc_norm = 0;
    ImuSample last_imu{};
    double last_lidar_end_time = 0;
    LioState start_state, end_state;
    bool have_undistorted = false;
    // device side
    hipStream_t copy_stream = nullptr;  // host scans travel at enqueue time, overlapping the scan being registered
    ImuPoseDev* d_poses = nullptr;
    ImuPoseDev* h_poses = nullptr;  // pinned
    unsigned long long* d_first = nullptr;
    int n_poses = 0;
};

struct JointCtx;
struct lio_devloop;  // batch.hip: the device-resident filter loop of one engine
lio_devloop* devloop_create(lio_scan* sc);
void devloop_destroy(lio_devloop* d);
int devloop_update(lio_devloop* d, lio_map* m, lio_scan* sc, const double* x26, const double* P529, double R, int max_iter, int degenerate_detect_en,
                   const lio::EskfDev** out);

struct lio_engine {
    Frontend* fe = nullptr;
    lio_devloop* dl = nullptr;
    int device_loop = 0;  // lio_engine_set_device_loop / LIO_DEVICE_LOOP=1: the iterate loop of ONE scan on the device too (one submission, one
                          // wait: the host thread is free meanwhile, the scan takes ~15 % longer -- five serial filter kernels and five blind
                          // tie-queue launches against four to five hand-overs); the batched engine always runs it on the device
    lio_map* map;
    lio_scan* scan;
    Eskf kf;
    // fastlio_init constants
    float leaf_surf = 0.5f, leaf_map = 0.5f;
    double init_time = 0.1, laser_cov = 0.001;
    bool degenerate_detect_en = true;
    bool static_map = false;
    bool own_map = true;
    bool map_seeded = false;  // NumValidGrids() != 0 observed (a map never becomes empty again)
    bool n_added_pending = false;  // the last scan's map_incremental runs un-waited on the map's stream: tm.n_added is filled in on demand
    // file-scope state of laserMapping.cpp
    double travel = 0, first_lidar_time = 0;
    double last_pos_lid[3] = {0, 0, 0};
    bool flg_first_scan = true, flg_EKF_inited = false, is_degenerate = false;
    // wheel-speed rows (laserMapping.cpp:794-811): wheelspeed_en, and Measures.ins.back() / lidar_end_time of the scan being registered
    bool wheelspeed_en = false, meas_ins_valid = false;
    double meas_ins_stamp = 0.0, meas_lidar_end = 0.0, meas_ins_vel[3] = {0, 0, 0};
    std::vector<lio_pass_log> log;
    // timing
    bool timing = false;
    hipEvent_t ev[2] = {nullptr, nullptr};  // the two timing events, made once by lio_engine_enable_timing (a pair per pass used to be created and destroyed)
    lio_timings tm;
    std::vector<double> rows6, hvec;
    lio_reduce_fn reduce = nullptr;  // cross-GPU reduction of the normal equations (joint registration)
    void* reduce_ctx = nullptr;
    JointCtx* joint = nullptr;       // native joint registration (lio_engine_set_joint) behind `reduce`
    struct lio_held_scan* held = nullptr;  // sequence batch: the scan between the two halves of fastlio_main (engine_fastlio_front / _back)
};

int engine_resume_update_impl(lio_engine* e, const double* x_now26, const double* x_prop26, const double* P_prop, int i, int converge, int t, bool keep_log);
static void joint_reduce(void* vctx, double* buf, int n);
struct lio_comm;
int comm_reduce_host(lio_comm* c, double* buf, int n);  // comm.hip

// Joint registration natively (lio_engine_set_joint): this engine drives the filter; after every linearisation its reduce step adds, in
// order, what the other LOCAL engines (further sub-maps resident on this GPU) see at the same iterate, then all-gathers and sums across
// the ranks of the communicator (sub-maps on other GPUs).  Same records and order as the generic hook documents.
struct JointCtx {
    lio_engine* self = nullptr;
    std::vector<lio_engine*> others;
    lio_comm* comm = nullptr;
    int knn_seen = 0;
    double nnT[9] = {0};
    int rc = LIO_OK;
    bool begun = false;  // the other sub-maps' linearisations of this pass are in flight (started before the driving engine's own)
    bool share_ds = false;  // this registration hands the driving engine's downsampled cloud to the other sub-maps' scan buffers
};

namespace {

void pose_arrays(const LioState& x, double pose[7], double ext[7]) {
    for (int i = 0; i < 3; i++) { pose[i] = x.pos[i]; ext[i] = x.til[i]; }
    for (int i = 0; i < 4; i++) { pose[3 + i] = x.rot[i]; ext[3 + i] = x.ril[i]; }
}

float ev_us(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f;
}

struct PassCtx {
    lio_engine* e;
    int rc;
};

// h_share_model (laserMapping.cpp:984-1023; wheelspeed_en == false): one device linearisation, then the
// degeneracy projection of :965-980 applied to the 6x6 normal equations instead of to N_eff rows:
// h_x[:, 0:3] <- h_x[:, 0:3] P^T  <=>  JtJ <- M JtJ M^T, Jtr <- M Jtr with M = blkdiag(P, I3).
int measure_pass(lio_engine* e, const LioState& x, bool converge, Measurement& m, lio_pass_log& pl) {
    double pose[7], ext[7];
    pose_arrays(x, pose, ext);
    lio_normal_eq ne;
    hipEvent_t t0, t1;
    if (e->timing) { t0 = e->ev[0]; t1 = e->ev[1]; hipEventRecord(t0, e->scan->stream); }
    if (e->joint && e->reduce == joint_reduce && e->joint->rc == LIO_OK) {
        // joint registration: the other local sub-maps' linearisations of this pass go out first, each on its own scan's stream, so that
        // they run beside the driving engine's own instead of one after the other behind it
        e->joint->begun = true;
        for (lio_engine* o : e->joint->others) {
            const int r0 = p2plane_linearize_begin(o->map, o->scan, pose, ext, converge ? 1 : 0);
            if (r0 != LIO_OK) { e->joint->rc = r0; e->joint->begun = false; break; }
        }
    }
    const int rc = lio_p2plane_linearize(e->map, e->scan, pose, ext, converge ? 1 : 0, &ne);
    if (e->timing) {
        hipEventRecord(t1, e->scan->stream);
        hipEventSynchronize(t1);
        const float us = ev_us(t0, t1);
        if (converge) e->tm.knn_us += us; else e->tm.linearize_us += us;
    }
    e->tm.n_pass++;
    if (converge) e->tm.n_knn_pass++;
    if (rc != LIO_OK && !e->reduce) return rc;
    if (e->reduce) {
        // joint registration: this rank's sums -> global sums (fixed rank order inside the hook), then the
        // degeneracy logic on the GLOBAL eigen-structure.
        // A rank whose own linearisation failed STILL takes part in the collective of this pass -- with a record of NaNs: the sums are then
        // NaN on every rank, and every rank gives the registration up here, in the same pass, instead of the healthy ones waiting for ever
        // in an all-gather the failed one never enters.
        double buf[29];
        int t = 0;
        if (rc == LIO_OK) {
            for (int a = 0; a < 6; a++)
                for (int c = a; c < 6; c++) buf[t++] = ne.JtJ[a * 6 + c];
            for (int a = 0; a < 6; a++) buf[21 + a] = ne.Jtr[a];
            buf[27] = ne.sum_abs_res;
            buf[28] = (double)ne.n_eff;
        } else {
            for (int a = 0; a < 29; a++) buf[a] = NAN;
        }
        e->reduce(e->reduce_ctx, buf, 29);
        if (rc != LIO_OK) return rc;
        if (!(buf[28] == buf[28])) { set_error("joint registration: a rank (or a local sub-map) reported a failure in this pass"); return LIO_E_DEVICE; }
        t = 0;
        for (int a = 0; a < 6; a++)
            for (int c = a; c < 6; c++) { ne.JtJ[a * 6 + c] = buf[t]; ne.JtJ[c * 6 + a] = buf[t]; t++; }
        for (int a = 0; a < 6; a++) ne.Jtr[a] = buf[21 + a];
        ne.sum_abs_res = buf[27];
        ne.n_eff = (uint32_t)(buf[28] + 0.5);
        for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) ne.nnT[a * 3 + c] = ne.JtJ[a * 6 + c];
        eig3_sym(ne.nnT, ne.eigval, ne.eigvec);
        bool need = false;
        for (int i = 0; i < 3; i++)
            if (!(ne.eigval[i] * (1.0 - 1e-5) - 0.030138 * (double)ne.n_eff >= 250.0 + 1e-3)) need = true;
        if (need && ne.n_eff > 0) {  // (a decision made from the GLOBAL sums: the same on every rank)
            double cs[6];
            const int r2 = lio_p2plane_degeneracy(e->scan, ne.eigvec, cs, cs + 3);
            if (r2 != LIO_OK)
                for (int a = 0; a < 6; a++) cs[a] = NAN;
            e->reduce(e->reduce_ctx, cs, 6);
            if (r2 != LIO_OK) return r2;
            if (!(cs[0] == cs[0])) { set_error("joint registration: a rank reported a failure in the degeneracy sums"); return LIO_E_DEVICE; }
            for (int i = 0; i < 3; i++) { ne.contri[i] = cs[i]; ne.strong[i] = cs[3 + i]; }
        } else {
            for (int i = 0; i < 3; i++) { ne.contri[i] = INFINITY; ne.strong[i] = INFINITY; }
        }
    }
    e->tm.n_ds = (int)ne.n_ds;
    e->tm.n_eff_last = (int)ne.n_eff;
    e->tm.knn_candidates = ((uint64_t)ne.n_knn_candidates_hi << 32) | ne.n_knn_candidates_lo;
    memset(&pl, 0, sizeof(pl));
    pl.knn = converge ? 1 : 0;
    pl.n_eff = (int)ne.n_eff;
    pl.sum_abs_res = ne.sum_abs_res;
    if (ne.n_eff < 1) {  // "No Effective Points!" (laserMapping.cpp:888-893)
        m.valid = false;
        pl.valid = 0;
        return LIO_OK;
    }
    m.valid = true;
    m.n_rows = (int)ne.n_eff;
    memcpy(m.HTH, ne.JtJ, sizeof(m.HTH));
    memcpy(m.HTh, ne.Jtr, sizeof(m.HTh));
    double Pm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool degenerate = false;
    if (e->degenerate_detect_en) {
        bool keep[3];
        for (int i = 0; i < 3; i++) {
            keep[i] = !((float)ne.contri[i] < 250.0f && (float)ne.strong[i] < 50.0f);
            if (!keep[i]) degenerate = true;
        }
        e->is_degenerate = degenerate;
        if (degenerate) {
            // mat_p = (V^T)^-1 * V2, V2 = V^T with the degenerate rows zeroed  (laserMapping.cpp:970-977)
            double Vt[9], V2[9], Vti[9];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    Vt[a * 3 + b] = ne.eigvec[b * 3 + a];
                    V2[a * 3 + b] = keep[a] ? Vt[a * 3 + b] : 0.0;
                }
            mat_inverse(Vt, 3, Vti);
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int k = 0; k < 3; k++) s += Vti[a * 3 + k] * V2[k * 3 + b];
                    Pm[a * 3 + b] = s;
                }
            double M[36] = {0}, T[36], O[36];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) M[a * 6 + b] = Pm[a * 3 + b];
            for (int a = 3; a < 6; a++) M[a * 6 + a] = 1.0;
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) {
                    double s = 0;
                    for (int k = 0; k < 6; k++) s += M[a * 6 + k] * m.HTH[k * 6 + b];
                    T[a * 6 + b] = s;
                }
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) {
                    double s = 0;
                    for (int k = 0; k < 6; k++) s += T[a * 6 + k] * M[b * 6 + k];
                    O[a * 6 + b] = s;
                }
            memcpy(m.HTH, O, sizeof(O));
            double v[6];
            for (int a = 0; a < 6; a++) {
                double s = 0;
                for (int k = 0; k < 6; k++) s += M[a * 6 + k] * m.HTh[k];
                v[a] = s;
            }
            memcpy(m.HTh, v, sizeof(v));
        }
    }
    if (m.n_rows < kDof && !e->reduce) {  // dense branch of the filter needs the rows themselves
        e->rows6.resize((size_t)m.n_rows * 6);
        e->hvec.resize(m.n_rows);
        const int r = lio_p2plane_rows(e->scan, pose, ext, e->rows6.data(), e->hvec.data(), (uint32_t)m.n_rows);
        if (r < 0) return r;
        if (degenerate)
            for (int k = 0; k < r; k++) {
                double* row = e->rows6.data() + (size_t)k * 6;
                const double o[3] = {row[0], row[1], row[2]};
                for (int a = 0; a < 3; a++) row[a] = Pm[a * 3] * o[0] + Pm[a * 3 + 1] * o[1] + Pm[a * 3 + 2] * o[2];
            }
        m.rows6 = e->rows6.data();
        m.h = e->hvec.data();
    }
    pl.valid = 1;
    pl.degenerate = degenerate ? 1 : 0;
    memcpy(pl.JtJ, m.HTH, sizeof(pl.JtJ));
    memcpy(pl.Jtr, m.HTh, sizeof(pl.Jtr));
    return LIO_OK;
}

void on_pass(void* vctx, int, bool, const Measurement&, const double* dx) {
    PassCtx* c = static_cast<PassCtx*>(vctx);
    if (dx && !c->e->log.empty()) memcpy(c->e->log.back().dx, dx, sizeof(double) * kDof);
}

// the iterated update with the loop on the device (batch.hip, eskf_dev.h): one submission, one wait
int run_update_device(lio_engine* e) {
    if (!e->dl) {
        e->dl = devloop_create(e->scan);
        if (!e->dl) { set_error("device loop: allocation failed"); return LIO_E_DEVICE; }
    }
    double x0[26];
    state_to_array(e->kf.x, x0);
    double P0[529];
    memcpy(P0, e->kf.P, sizeof(P0));
    const EskfDev* c = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = devloop_update(e->dl, e->map, e->scan, x0, P0, e->laser_cov, e->kf.maximum_iter, e->degenerate_detect_en ? 1 : 0, &c);
    if (rc != LIO_OK) return rc;
    e->log.clear();
    for (int k = 0; k < c->n_log && k < kEkMaxPass; k++) {
        lio_pass_log pl;
        static_assert(sizeof(pl) == sizeof(EkPassLog), "EkPassLog mirrors lio_pass_log");
        memcpy(&pl, &c->log[k], sizeof(pl));
        e->log.push_back(pl);
    }
    e->tm.n_pass = c->n_pass;
    e->tm.n_knn_pass = c->n_knn;
    e->tm.n_eff_last = c->n_eff_last;
    e->tm.n_ds = e->scan->have_ds > 0 ? e->scan->have_ds : e->tm.n_ds;
    e->is_degenerate = c->is_degenerate != 0;
    if (c->status == EK_NEEDS_HOST) {  // a pass with 1 <= N_eff < 23: the host filter takes over from that pass (dense gain, esekfom.hpp:1715-1744)
        const int np = e->tm.n_pass, nk = e->tm.n_knn_pass;
        const int r2 = engine_resume_update_impl(e, c->x, x0, P0, c->i, c->converge, c->t, true);
        e->tm.n_pass += np;
        e->tm.n_knn_pass += nk;
        if (r2 != LIO_OK) return r2;
    } else {
        state_from_array(c->x, e->kf.x);
        memcpy(e->kf.P, c->P, sizeof(e->kf.P));
    }
    e->tm.host_solve_us = (float)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return LIO_OK;
}

// h_share_model's wheel-speed part (laserMapping.cpp:794-811, 994-1012) on top of whatever the point-to-plane part of this pass is -- fresh
// rows, the previous pass's whole measurement (stale copy), or nothing: three rows dh/dv = I3 with h = (rot * v_ins - vel) * weight,
// weight = 1e-4 (1e-3 when degenerate) x the rows before them, as a float.
void append_wheelspeed_rows(bool is_degenerate, const double ins_vel[3], const LioState& x, Measurement& m) {
    if (!m.valid) {  // no point-to-plane rows at all: the three rows are the measurement (n_terms = 3)
        m.valid = true;
        m.n_rows = 0;
        m.n_geo = 0;
        m.ws_n = 0;
        m.rows6 = nullptr;
        m.h = nullptr;
        memset(m.HTH, 0, sizeof(m.HTH));
        memset(m.HTh, 0, sizeof(m.HTh));
    }
    if (m.ws_n == 0) m.n_geo = m.n_rows;
    if (m.ws_n >= 6) return;
    const float weight = !is_degenerate ? (float)(0.0001 * m.n_rows) : (float)(0.001 * m.n_rows);
    double vel[3];
    quat_rotate(x.rot, ins_vel, vel);
    for (int a = 0; a < 3; a++) m.ws_h[m.ws_n][a] = (vel[a] - x.vel[a]) * weight;
    m.ws_n++;
    m.n_rows += 3;
}
void append_wheelspeed(lio_engine* e, const LioState& x, Measurement& m) {
    if (!(e->wheelspeed_en && e->meas_ins_valid && (e->meas_lidar_end - e->meas_ins_stamp) < 0.01)) return;
    append_wheelspeed_rows(e->is_degenerate, e->meas_ins_vel, x, m);
}

// ekfom_data_geo is a COPY of the shared struct (laserMapping.cpp:991): when a later pass finds no effective point, the rows of the previous
// pass -- its point-to-plane rows AND the wheel-speed rows that pass appended -- survive in it and the filter re-uses them (n_terms != 0).
// One keeper serves run_update, the resumed update and the filter-level harness (lio_eskf_update_ws_cb).
struct StaleRows {
    Measurement prev;
    std::vector<double> rows, h;
    bool have = false;
    // a pass without effective points takes over the previous pass's whole measurement; returns whether it did
    bool take_over(Measurement& m) {
        if (m.valid || !have) return false;
        m = prev;
        if (prev.rows6) { m.rows6 = rows.data(); m.h = h.data(); }
        return true;
    }
    // what the next pass's copy of the shared struct holds: the whole measurement of this one
    void remember(const Measurement& m) {
        if (!m.valid) return;
        if (m.rows6 && m.rows6 != rows.data()) {
            const int ng = m.ws_n > 0 ? m.n_geo : m.n_rows;
            rows.assign(m.rows6, m.rows6 + (size_t)ng * 6);
            h.assign(m.h, m.h + ng);
        }
        prev = m;
        have = true;
    }
};

int run_update(lio_engine* e) {
    {   // the previous scan's map_incremental may still be running on the map's stream (process_core does not wait for it): the host looks at
        // its outcome here, before anything of this update is launched or captured
        const int rc_settle = map_settle(e->map);
        if (rc_settle != LIO_OK) return rc_settle;
    }
    if (e->device_loop && !e->reduce && !e->timing && !e->wheelspeed_en && e->kf.maximum_iter + 1 <= kEkMaxPass) return run_update_device(e);
    e->log.clear();
    PassCtx ctx{e, LIO_OK};
    double host_us = 0;
    // ekfom_data_geo is a COPY of the shared struct (laserMapping.cpp:991): when a later pass finds no
    // effective point, the rows of the previous pass survive in it and the filter re-uses them (n_terms != 0).
    StaleRows stale;
    auto measure = [&](const LioState& x, bool converge, Measurement& m) {
        lio_pass_log pl;
        memset(&pl, 0, sizeof(pl));
        if (ctx.rc != LIO_OK) { m.valid = false; return; }  // a device error earlier in this update: no further launches
        const int rc = measure_pass(e, x, converge, m, pl);
        if (rc != LIO_OK) { ctx.rc = rc; m.valid = false; e->log.push_back(pl); return; }
        if (stale.take_over(m)) {
            pl.valid = 1;
            memcpy(pl.JtJ, m.HTH, sizeof(pl.JtJ));
            memcpy(pl.Jtr, m.HTh, sizeof(pl.Jtr));
        }
        append_wheelspeed(e, x, m);
        if (m.valid) pl.valid = 1;
        stale.remember(m);
        e->log.push_back(pl);
    };
    const auto t0 = std::chrono::steady_clock::now();
    e->kf.update_iterated(e->laser_cov, measure, on_pass, &ctx);
    host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    e->tm.host_solve_us = (float)host_us;  // wall time of the whole iterated update (device passes included)
    return ctx.rc;
}

}  // namespace

// The device-resident loop (batch.hip) stops before a pass that needs the rows themselves (1 <= N_eff < 23, esekfom.hpp:1715-1744);
// the host filter continues from exactly there: state at the start of that pass, the propagated state / covariance of the update, the
// loop counters -- through this engine's per-pass path (same scan buffers, neighbour cache and gate flags the device loop left).
int engine_resume_update(lio_engine* e, const double* x_now26, const double* x_prop26, const double* P_prop, int i, int converge, int t) {
    return engine_resume_update_impl(e, x_now26, x_prop26, P_prop, i, converge, t, false);
}
int engine_resume_update_impl(lio_engine* e, const double* x_now26, const double* x_prop26, const double* P_prop, int i, int converge, int t, bool keep_log) {
    if (!e || !x_now26 || !x_prop26 || !P_prop) return LIO_E_INVALID;
    hipSetDevice(e->scan->device);
    if (!keep_log) e->log.clear();
    memset(&e->tm, 0, sizeof(e->tm));
    PassCtx ctx{e, LIO_OK};
    Eskf::Work w;
    state_from_array(x_prop26, w.x_prop);
    memcpy(w.P_prop, P_prop, sizeof(w.P_prop));
    memset(w.K_x, 0, sizeof(w.K_x));
    memset(w.K_h, 0, sizeof(w.K_h));
    memset(w.dx_new, 0, sizeof(w.dx_new));
    state_from_array(x_now26, e->kf.x);
    memcpy(e->kf.P, P_prop, sizeof(w.P_prop));
    StaleRows stale;
    auto measure = [&](const LioState& x, bool conv, Measurement& m) {
        lio_pass_log pl;
        memset(&pl, 0, sizeof(pl));
        if (ctx.rc != LIO_OK) { m.valid = false; return; }
        const int rc = measure_pass(e, x, conv, m, pl);
        if (rc != LIO_OK) { ctx.rc = rc; m.valid = false; e->log.push_back(pl); return; }
        if (stale.take_over(m)) {
            pl.valid = 1;
            memcpy(pl.JtJ, m.HTH, sizeof(pl.JtJ));
            memcpy(pl.Jtr, m.HTh, sizeof(pl.Jtr));
        }
        append_wheelspeed(e, x, m);
        if (m.valid) pl.valid = 1;
        stale.remember(m);
        e->log.push_back(pl);
    };
    e->kf.update_iterated_from(w, i, converge != 0, t, e->laser_cov, measure, on_pass, &ctx);
    return ctx.rc;
}
int engine_joint_register_device(lio_engine* e, const void* d_raw, uint32_t n_raw, double lidar_beg_time, double state26[26], double cov[529]) {
    return lio_engine_joint_register_device(e, d_raw, n_raw, lidar_beg_time, state26, cov);
}
void engine_count_passes(lio_engine* e, int* n_pass, int* n_knn) {
    *n_pass = e->tm.n_pass;
    *n_knn = e->tm.n_knn_pass;
}

extern "C" {

lio_engine* lio_engine_create(int device, float resolution, int stencil, uint64_t max_points, uint64_t max_voxels, uint32_t max_raw,
                              uint32_t max_ds) {
    lio_map* m = lio_map_create(device, resolution, max_points, max_voxels, stencil);
    if (!m) return nullptr;
    lio_scan* s = lio_scan_create(device, max_raw, max_ds);
    if (!s) { lio_map_destroy(m); return nullptr; }
    lio_engine* e = new lio_engine();
    e->map = m;
    e->scan = s;
    { const char* k = getenv("LIO_DEVICE_LOOP"); e->device_loop = (k && k[0] == '1') ? 1 : 0; }
    s->resize_min = 5;
    memset(&e->tm, 0, sizeof(e->tm));
    return e;
}

lio_engine* lio_engine_create_shared(lio_map* shared_map, uint32_t max_raw, uint32_t max_ds) {
    if (!shared_map) return nullptr;
    lio_scan* s = lio_scan_create(shared_map->device, max_raw, max_ds);
    if (!s) return nullptr;
    lio_engine* e = new lio_engine();
    e->map = shared_map;
    e->scan = s;
    { const char* k = getenv("LIO_DEVICE_LOOP"); e->device_loop = (k && k[0] == '1') ? 1 : 0; }
    s->resize_min = 5;
    e->own_map = false;
    e->static_map = true;  // several engines read one map concurrently: nobody inserts
    e->map_seeded = true;
    memset(&e->tm, 0, sizeof(e->tm));
    return e;
}

static void frontend_destroy(lio_engine* e);
void lio_engine_destroy(lio_engine* e) {
    if (!e) return;
    frontend_destroy(e);
    if (e->dl) { hipStreamSynchronize(e->scan->stream); devloop_destroy(e->dl); }
    delete e->joint;  // (the other sub-maps' engines may be gone already: their scans are not touched here -- lio_engine_set_joint(e, NULL, 0, NULL)
                      // before destroying the driver hands them their own degeneracy evaluation back)
    if (e->ev[0]) { hipEventDestroy(e->ev[0]); hipEventDestroy(e->ev[1]); }
    delete e->held;
    lio_scan_destroy(e->scan);
    if (e->own_map) lio_map_destroy(e->map);
    delete e;
}

lio_map* lio_engine_map(lio_engine* e) { return e ? e->map : nullptr; }
lio_scan* lio_engine_scan(lio_engine* e) { return e ? e->scan : nullptr; }

int lio_engine_set_state(lio_engine* e, const double s[26]) { if (!e || !s) return LIO_E_INVALID; state_from_array(s, e->kf.x); return LIO_OK; }
int lio_engine_get_state(lio_engine* e, double s[26]) { if (!e || !s) return LIO_E_INVALID; state_to_array(e->kf.x, s); return LIO_OK; }
int lio_engine_set_cov(lio_engine* e, const double P[529]) { if (!e || !P) return LIO_E_INVALID; memcpy(e->kf.P, P, sizeof(double) * 529); return LIO_OK; }
int lio_engine_get_cov(lio_engine* e, double P[529]) { if (!e || !P) return LIO_E_INVALID; memcpy(P, e->kf.P, sizeof(double) * 529); return LIO_OK; }

int lio_engine_set_flags(lio_engine* e, int ekf_inited, int first_scan, double travel, double first_lidar_time) {
    if (!e) return LIO_E_INVALID;
    e->flg_EKF_inited = ekf_inited != 0;
    e->flg_first_scan = first_scan != 0;
    e->travel = travel;
    e->first_lidar_time = first_lidar_time;
    return LIO_OK;
}
double lio_engine_travel(lio_engine* e) { return e ? e->travel : 0.0; }
int lio_engine_is_degenerate(lio_engine* e) { return e && e->is_degenerate ? 1 : 0; }

int lio_engine_update(lio_engine* e) {
    if (!e) return LIO_E_INVALID;
    const int rc = run_update(e);
    if (rc != LIO_OK) return rc;
    return (int)e->log.size();
}

int lio_engine_pass_log(lio_engine* e, int i, lio_pass_log* out) {
    if (!e || !out || i < 0 || i >= (int)e->log.size()) return LIO_E_INVALID;
    *out = e->log[i];
    return LIO_OK;
}

int lio_engine_enable_timing(lio_engine* e, int on) {
    if (!e) return LIO_E_INVALID;
    if (on && !e->ev[0]) {
        hipSetDevice(e->scan->device);
        LIO_HIP_TRY(hipEventCreate(&e->ev[0]));
        LIO_HIP_TRY(hipEventCreate(&e->ev[1]));
    }
    e->timing = on != 0;
    return LIO_OK;
}
// the joint context goes with its hook: the other sub-maps' scans evaluate their own degeneracy sums again
static void joint_teardown(lio_engine* e) {
    if (!e->joint) return;
    for (lio_engine* o : e->joint->others) lio_scan_set_degeneracy_mode(o->scan, 0);
    delete e->joint;
    e->joint = nullptr;
}

int lio_engine_set_reduce_hook(lio_engine* e, lio_reduce_fn fn, void* ctx) {
    if (!e) return LIO_E_INVALID;
    if (fn != joint_reduce) joint_teardown(e);  // a caller's hook (or none) replaces a native joint registration
    e->reduce = fn;
    e->reduce_ctx = ctx;
    // with a hook the degeneracy sums are evaluated here, after the reduction, never inside linearize
    return lio_scan_set_degeneracy_mode(e->scan, fn ? 2 : 0);
}

static void joint_reduce(void* vctx, double* buf, int n) {
    JointCtx* j = static_cast<JointCtx*>(vctx);
    lio_engine* e = j->self;
    // Every call takes part in the collective, whatever happened locally: a failure (here, or in the caller's own linearisation, which then
    // hands in NaNs) travels as NaN sums, every rank sees them and gives up in the same pass (measure_pass).
    auto poison = [&]() { for (int k = 0; k < n; k++) buf[k] = NAN; };
    if (n == 29) {
        const bool redo = e->tm.n_knn_pass > j->knn_seen;  // (already counts the pass being reduced)
        j->knn_seen = e->tm.n_knn_pass;
        double pose[7], ext[7];
        pose_arrays(e->kf.x, pose, ext);
        for (lio_engine* o : j->others) {  // (all of them, also after a failure: what was begun is ended)
            lio_normal_eq ne;
            const int rc = j->begun ? p2plane_linearize_end(o->map, o->scan, pose, ext, redo ? 1 : 0, &ne)
                                    : (j->rc == LIO_OK ? lio_p2plane_linearize(o->map, o->scan, pose, ext, redo ? 1 : 0, &ne) : j->rc);
            if (rc != LIO_OK) { if (j->rc == LIO_OK) j->rc = rc; continue; }
            int t = 0;
            for (int a = 0; a < 6; a++)
                for (int c = a; c < 6; c++) buf[t++] += ne.JtJ[a * 6 + c];
            for (int a = 0; a < 6; a++) buf[21 + a] += ne.Jtr[a];
            buf[27] += ne.sum_abs_res;
            buf[28] += (double)ne.n_eff;
        }
        j->begun = false;
        if (j->rc != LIO_OK) poison();
        if (j->comm) { const int rc = comm_reduce_host(j->comm, buf, 29); if (rc != LIO_OK) { if (j->rc == LIO_OK) j->rc = rc; poison(); return; } }
        double J[36];
        int t = 0;
        for (int a = 0; a < 6; a++)
            for (int c = a; c < 6; c++) { J[a * 6 + c] = buf[t]; J[c * 6 + a] = buf[t]; t++; }
        for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) j->nnT[a * 3 + c] = J[a * 6 + c];
    } else {  // the six degeneracy sums against the eigenvectors of the GLOBAL sum n n^T (the same decomposition measure_pass made)
        double w[3], V[9];
        eig3_sym(j->nnT, w, V);
        for (lio_engine* o : j->others) {
            double cs[6];
            const int rc = j->rc == LIO_OK ? lio_p2plane_degeneracy(o->scan, V, cs, cs + 3) : j->rc;
            if (rc != LIO_OK) { if (j->rc == LIO_OK) j->rc = rc; continue; }
            for (int k = 0; k < 6; k++) buf[k] += cs[k];
        }
        if (j->rc != LIO_OK) poison();
        if (j->comm) { const int rc = comm_reduce_host(j->comm, buf, 6); if (rc != LIO_OK) { if (j->rc == LIO_OK) j->rc = rc; poison(); return; } }
    }
}

int lio_engine_set_joint(lio_engine* e, lio_engine** others, int n_others, lio_comm* comm) {
    if (!e || n_others < 0 || (n_others && !others)) return LIO_E_INVALID;
    joint_teardown(e);
    if (n_others == 0 && !comm) return lio_engine_set_reduce_hook(e, nullptr, nullptr);
    JointCtx* j = new JointCtx();
    j->self = e;
    j->comm = comm;
    for (int k = 0; k < n_others; k++) {
        if (!others[k] || others[k] == e) { delete j; return LIO_E_INVALID; }
        j->others.push_back(others[k]);
        lio_scan_set_degeneracy_mode(others[k]->scan, 2);  // their degeneracy sums are evaluated after the global reduction
    }
    e->joint = j;
    return lio_engine_set_reduce_hook(e, joint_reduce, j);
}

// one joint registration: the same cloud to every local engine (each downsamples it into its own scan buffers), then the driving engine's
// per-scan body with the joint reduce step.  state26 / cov: prior in, posterior out (identical on every rank of the communicator).
static int joint_register_impl(lio_engine* e, const void* raw, bool on_device, uint32_t n_raw, double lidar_beg_time, double state26[26], double cov[529]) {
    if (!e || !e->joint || !state26 || !cov) return LIO_E_INVALID;
    JointCtx* j = e->joint;
    j->knn_seen = 0;
    j->rc = LIO_OK;
    j->begun = false;
    // Whether a scan is registered at all must not depend on rank-local state: every rank sees the same cloud (hence the same downsampled
    // size and the same "too few points" decision), the first-scan latch and the seeding of an empty map (laserMapping.cpp:1171-1177,
    // 1227-1239) do not apply to a registration against prebuilt sub-maps -- a rank whose sub-map is empty contributes zero rows.
    e->flg_first_scan = false;
    e->map_seeded = true;
    // the cloud is uploaded and downsampled ONCE, by the driving engine; the other sub-maps' scan buffers receive its downsampled points
    // device to device right after (process_core) -- same leaf, same input: what each of them would have computed itself
    j->share_ds = true;
    for (lio_engine* o : j->others)
        if (o->leaf_surf != e->leaf_surf || o->scan->device != e->scan->device) j->share_ds = false;
    if (!j->share_ds)
        for (lio_engine* o : j->others) {
            int rc = on_device ? lio_scan_set_device(o->scan, raw, n_raw) : lio_scan_upload(o->scan, static_cast<const float*>(raw), n_raw);
            if (rc == LIO_OK) rc = lio_scan_voxel_downsample(o->scan, o->leaf_surf, 1, nullptr);
            if (rc != LIO_OK) return rc;
        }
    lio_engine_set_state(e, state26);
    lio_engine_set_cov(e, cov);
    const int rc = on_device ? lio_engine_process_scan_device(e, raw, n_raw, lidar_beg_time) : lio_engine_process_scan(e, static_cast<const float*>(raw), n_raw, lidar_beg_time);
    j->share_ds = false;
    if (j->rc != LIO_OK) return j->rc;
    lio_engine_get_state(e, state26);
    lio_engine_get_cov(e, cov);
    return rc;
}
int lio_engine_joint_register(lio_engine* e, const float* raw_body_xyzi, uint32_t n_raw, double lidar_beg_time, double state26[26], double cov[529]) {
    return joint_register_impl(e, raw_body_xyzi, false, n_raw, lidar_beg_time, state26, cov);
}
int lio_engine_joint_register_device(lio_engine* e, const void* d_raw_body_xyzi, uint32_t n_raw, double lidar_beg_time, double state26[26], double cov[529]) {
    return joint_register_impl(e, d_raw_body_xyzi, true, n_raw, lidar_beg_time, state26, cov);
}

int lio_engine_set_static_map(lio_engine* e, int on) {
    if (!e) return LIO_E_INVALID;
    if (!e->own_map && !on) { set_error("an engine on a shared map is read-only"); return LIO_E_STATE; }
    e->static_map = on != 0;
    return LIO_OK;
}
int lio_engine_timings(lio_engine* e, lio_timings* out) {
    if (!e || !out) return LIO_E_INVALID;
    if (e->n_added_pending) {  // the last scan's map_incremental was not waited for (process_core), and is not waited for here either:
        // its count if it has finished, else -1 ("still running": lio_engine_flush waits)
        const int rc = map_settle_if_done(e->map);
        if (rc != LIO_OK) { e->n_added_pending = false; return rc; }
        if (!e->map->insert_pending) {
            e->n_added_pending = false;
            e->tm.n_added = (int)e->map->settled_n_add;
        } else {
            e->tm.n_added = -1;
        }
    }
    *out = e->tm;
    return LIO_OK;
}
int lio_engine_flush(lio_engine* e) {
    if (!e) return LIO_E_INVALID;
    hipSetDevice(e->map->device);
    const int rc = map_settle(e->map);
    if (rc == LIO_OK && e->n_added_pending) { e->n_added_pending = false; e->tm.n_added = (int)e->map->settled_n_add; }
    return rc;
}

static int process_core(lio_engine* e, double lidar_beg_time);
static int after_update(lio_engine* e, std::chrono::steady_clock::time_point w0);
static int process_common(lio_engine* e, double lidar_beg_time) {
    memset(&e->tm, 0, sizeof(e->tm));
    e->n_added_pending = false;
    if (e->flg_first_scan) {  // laserMapping.cpp:1171-1177
        e->first_lidar_time = lidar_beg_time;
        e->flg_first_scan = false;
        return 0;
    }
    return process_core(e, lidar_beg_time);
}
// fastlio_main from "feats_undistort->empty()" on (laserMapping.cpp:1193-1304)
static int process_core(lio_engine* e, double lidar_beg_time) {
    lio_scan* s = e->scan;
    const auto w0 = std::chrono::steady_clock::now();
    {   // the previous scan's map_incremental, if it has finished by now (it nearly always has): its failure is reported HERE, before the
        // registration of this scan launches anything; if it is still running the update looks again (run_update) before it reads the map
        const int rc_prev = map_settle_if_done(e->map);
        if (rc_prev != LIO_OK) return rc_prev;
    }
    if (s->n_raw == 0) return 2;  // "FastLio undistort points is empty"
    e->flg_EKF_inited = (lidar_beg_time - e->first_lidar_time) < e->init_time ? false : true;
    hipEvent_t t0, t1;
    if (e->timing) { t0 = e->ev[0]; t1 = e->ev[1]; hipEventRecord(t0, s->stream); }
    uint32_t n_ds = 0;
    int rc = lio_scan_voxel_downsample(s, e->leaf_surf, 1, &n_ds);
    if (e->timing) {
        hipEventRecord(t1, s->stream);
        hipEventSynchronize(t1);
        e->tm.downsample_us = ev_us(t0, t1);
    }
    if (rc != LIO_OK) return rc;
    e->tm.n_ds = (int)n_ds;
    if (e->joint && e->joint->share_ds)
        for (lio_engine* o : e->joint->others) {
            const int r2 = scan_share_ds(o->scan, s, n_ds);
            if (r2 != LIO_OK) return r2;
        }
    double pose[7], ext[7];
    uint64_t nv = 0;
    if (!e->map_seeded) {
        rc = lio_map_stats(e->map, nullptr, &nv);
        if (rc != LIO_OK) return rc;
        e->map_seeded = nv != 0;
    }
    if (!e->map_seeded) {  // laserMapping.cpp:1227-1239: seed the map with the first downsampled scan
        if (n_ds > 5) {
            pose_arrays(e->kf.x, pose, ext);
            rc = lio_map_seed(e->map, s, pose, ext, e->travel);
            if (rc < 0) return rc;
        }
        return 1;
    }
    if (e->map->stencil_id != 19 && (lidar_beg_time - e->first_lidar_time) > 10 * e->init_time) lio_map_set_stencil(e->map, 19);  // :1241-1243
    if (n_ds < 5) return 2;
    rc = run_update(e);
    if (rc != LIO_OK) return rc;
    return after_update(e, w0);
}
// fastlio_main after the filter update (laserMapping.cpp:1288-1304): travel distance of the lidar origin, map_incremental
static int after_update(lio_engine* e, std::chrono::steady_clock::time_point w0) {
    lio_scan* s = e->scan;
    double pose[7], ext[7];
    int rc = LIO_OK;
    hipEvent_t t0, t1;
    // travel distance of the lidar origin (laserMapping.cpp:1288-1291)
    double off[3];
    quat_rotate(e->kf.x.rot, e->kf.x.til, off);
    double pos_lid[3], d2 = 0;
    for (int i = 0; i < 3; i++) {
        pos_lid[i] = e->kf.x.pos[i] + off[i];
        const double d = pos_lid[i] - e->last_pos_lid[i];
        d2 += d * d;
        e->last_pos_lid[i] = pos_lid[i];
    }
    e->travel = e->travel + sqrt(d2);
    pose_arrays(e->kf.x, pose, ext);
    if (e->static_map) {
        e->tm.total_device_us = e->tm.downsample_us + e->tm.knn_us + e->tm.linearize_us;
        e->tm.total_wall_us = (float)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        return 3;
    }
    if (!e->timing) {
        // not waited for: the insert chain runs on the map's stream while this call returns and the next scan is uploaded, undistorted
        // and downsampled; the next neighbour search (map_knn_plane) looks at its outcome first.  n_added is filled in by lio_engine_timings
        rc = map_incremental_async(e->map, s, pose, ext, e->leaf_map, e->flg_EKF_inited ? 1 : 0, e->travel);
        if (rc < 0) return rc;
        e->n_added_pending = true;
        e->tm.total_device_us = e->tm.downsample_us + e->tm.knn_us + e->tm.linearize_us;
        e->tm.total_wall_us = (float)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        return 3;
    }
    t0 = e->ev[0]; t1 = e->ev[1]; hipEventRecord(t0, s->stream);
    rc = lio_map_incremental(e->map, s, pose, ext, e->leaf_map, e->flg_EKF_inited ? 1 : 0, e->travel);
    if (e->timing) {
        hipEventRecord(t1, s->stream);
        hipEventSynchronize(t1);
        e->tm.insert_us = ev_us(t0, t1);
    }
    if (rc < 0) return rc;
    e->tm.n_added = rc;
    e->tm.total_device_us = e->tm.downsample_us + e->tm.knn_us + e->tm.linearize_us + e->tm.insert_us;
    e->tm.total_wall_us = (float)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
    return 3;
}

// ---- sequence mode of the batched engine (batch.hip: lio_batch_create_sequences): the host half of fastlio_main around a round.  Internal (not in
// include/lio_hip.h).  A slot's engine owns the session's map and its file-scope state (first_lidar_time, travel, flags); the round does on the
// device what process_core does between the downsample and map_incremental, for B sessions at once. ---------------------------------------------
// Before the round: process_common + process_core up to the update.  Returns < 0 on error, 0 = the session's first scan (only its time is kept:
// laserMapping.cpp:1171-1177), 2 = empty scan, 10 = this scan goes through the engine's own path (the map is not seeded yet: downsample + seed),
// 11 = the scan takes a slot of the round (ekf_inited / the map's stencil are settled here).
int engine_seq_prepare(lio_engine* e, uint32_t n_raw, double lidar_beg_time, int* ekf_inited) {
    if (!e) return LIO_E_INVALID;
    if (e->reduce || e->joint || e->timing || e->wheelspeed_en || e->kf.maximum_iter + 1 > kEkMaxPass) {
        set_error("sequence batch: an engine with a reduce hook, joint registration, per-scan timing or wheel-speed rows runs on its own path only");
        return LIO_E_STATE;
    }
    const int rc_prev = map_settle(e->map);  // an insert this engine's own path left running on the map's stream (the scan before this one went that way)
    if (rc_prev != LIO_OK) return rc_prev;
    if (e->flg_first_scan) {
        memset(&e->tm, 0, sizeof(e->tm));
        e->n_added_pending = false;
        e->first_lidar_time = lidar_beg_time;
        e->flg_first_scan = false;
        return 0;
    }
    if (n_raw == 0) { memset(&e->tm, 0, sizeof(e->tm)); e->n_added_pending = false; return 2; }
    if (!e->map_seeded) {
        uint64_t nv = 0;
        const int rc = lio_map_stats(e->map, nullptr, &nv);
        if (rc != LIO_OK) return rc;
        e->map_seeded = nv != 0;
    }
    if (!e->map_seeded || e->map->n_batches == 0) return 10;
    memset(&e->tm, 0, sizeof(e->tm));
    e->n_added_pending = false;
    e->flg_EKF_inited = (lidar_beg_time - e->first_lidar_time) < e->init_time ? false : true;
    if (e->map->stencil_id != 19 && (lidar_beg_time - e->first_lidar_time) > 10 * e->init_time) lio_map_set_stencil(e->map, 19);  // :1241-1243
    *ekf_inited = e->flg_EKF_inited ? 1 : 0;
    return 11;
}
// the filter constants and the state of laserMapping.cpp's file scope the round needs
void engine_seq_params(lio_engine* e, double* laser_cov, int* maximum_iter, int* degenerate_detect_en, float* leaf_surf, float* leaf_map, int* static_map,
                       double* travel, double last_pos_lid[3]) {
    *laser_cov = e->laser_cov;
    *maximum_iter = e->kf.maximum_iter;
    *degenerate_detect_en = e->degenerate_detect_en ? 1 : 0;
    *leaf_surf = e->leaf_surf;
    *leaf_map = e->leaf_map;
    *static_map = e->static_map ? 1 : 0;
    *travel = e->travel;
    for (int i = 0; i < 3; i++) last_pos_lid[i] = e->last_pos_lid[i];
}
// After the round, for a slot whose update finished on the device: the posterior into the engine's filter, the travel distance (the additions the
// device made for its own copy), the bookkeeping of the insert the round ran (map_insert_dev's host side).  Returns 3 like process_core.
int engine_seq_finish(lio_engine* e, const double x26[26], const double P529[529], int n_ds, int n_pass, int n_knn, int n_eff, int degenerate, int inserted,
                      uint32_t n_add, uint32_t map_err, uint32_t bound) {
    state_from_array(x26, e->kf.x);
    memcpy(e->kf.P, P529, sizeof(e->kf.P));
    e->log.clear();  // (the pass logs stay on the device in this mode)
    e->tm.n_ds = n_ds;
    e->tm.n_pass = n_pass;
    e->tm.n_knn_pass = n_knn;
    e->tm.n_eff_last = n_eff;
    e->is_degenerate = degenerate != 0;
    double off[3];
    quat_rotate(e->kf.x.rot, e->kf.x.til, off);
    double d2 = 0;
    for (int i = 0; i < 3; i++) {
        const double pl = e->kf.x.pos[i] + off[i];
        const double d = pl - e->last_pos_lid[i];
        d2 += d * d;
        e->last_pos_lid[i] = pl;
    }
    e->travel = e->travel + sqrt(d2);
    if (inserted) {
        lio_map* m = e->map;
        m->n_batches++;
        if (m->lru_capacity) m->tomb_bound += bound;
        e->tm.n_added = (int)n_add;
        if (map_err) {
            set_error("map capacity exceeded by map_incremental (err bits 0x%x: 1 table full, 2 point pool full, 4 more than max_voxels voxels, 8 LRU log overrun)", map_err);
            return LIO_E_CAPACITY;
        }
    }
    return 3;
}
// ... and for a slot whose update the device handed over (a pass with 1 <= N_eff < 23): the host filter finishes it through the engine's per-pass
// path, then travel + map_incremental on the engine's own path (not waited for, as process_core leaves it)
int engine_seq_resume(lio_engine* e, const double* x_now26, const double* x_prop26, const double* P_prop, int i, int converge, int t) {
    const auto w0 = std::chrono::steady_clock::now();
    const int rc = engine_resume_update_impl(e, x_now26, x_prop26, P_prop, i, converge, t, false);
    if (rc != LIO_OK) return rc;
    return after_update(e, w0);
}
int engine_seq_has_lru(lio_engine* e) { return e->map->lru_capacity != 0 ? 1 : 0; }

int lio_engine_process_scan(lio_engine* e, const float* raw, uint32_t n_raw, double lidar_beg_time) {
    if (!e) return LIO_E_INVALID;
    const int rc = lio_scan_upload(e->scan, raw, n_raw);
    if (rc != LIO_OK) return rc;
    return process_common(e, lidar_beg_time);
}

int lio_engine_process_scan_device(lio_engine* e, const void* d_raw, uint32_t n_raw, double lidar_beg_time) {
    if (!e) return LIO_E_INVALID;
    const int rc = lio_scan_set_device(e->scan, d_raw, n_raw);
    if (rc != LIO_OK) return rc;
    return process_common(e, lidar_beg_time);
}

int lio_engines_process_batch(lio_engine** engines, int n_engines, lio_scan_job* jobs, int n_jobs) {
    if (!engines || n_engines < 1 || (!jobs && n_jobs)) return LIO_E_INVALID;
    for (int i = 0; i < n_engines; i++)
        if (!engines[i]) return LIO_E_INVALID;
    std::atomic<int> next(0);
    std::atomic<int> first_err(0);
    auto work = [&](int t) {
        lio_engine* e = engines[t];
        for (;;) {
            const int j = next.fetch_add(1);
            if (j >= n_jobs) break;
            lio_scan_job& job = jobs[j];
            int rc = LIO_E_INVALID;
            if (job.flags & ~LIO_JOB_FLAGS_KNOWN) {
                set_error("lio_scan_job.flags = 0x%x: unknown bits (a job array that was not zero-initialised?)", job.flags);
            } else if (job.flags & LIO_JOB_IDLE) {
                rc = 0;  // (sequence mode's marker for a session without a scan: nothing to do)
            } else if (job.state_in && job.cov_in) {
                state_from_array(job.state_in, e->kf.x);
                memcpy(e->kf.P, job.cov_in, sizeof(double) * 529);
                rc = (job.flags & LIO_JOB_KEEP_CACHE) ? LIO_OK : scan_forget_cache(e->scan);  // an independent scan: no neighbours of the engine's previous job
                if (rc == LIO_OK)
                    rc = (job.flags & LIO_JOB_HOST_RAW) ? lio_engine_process_scan(e, static_cast<const float*>(job.d_raw), job.n_raw, job.lidar_beg_time)
                                                        : lio_engine_process_scan_device(e, job.d_raw, job.n_raw, job.lidar_beg_time);
            }
            job.rc = rc;
            job.n_ds = e->tm.n_ds;
            job.n_pass = e->tm.n_pass;
            job.n_knn_pass = e->tm.n_knn_pass;
            if (job.state_out) state_to_array(e->kf.x, job.state_out);
            if (rc < 0) { int z = 0; first_err.compare_exchange_strong(z, rc); }
        }
    };
    if (n_engines == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_engines; t++) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    return first_err.load();
}

void lio_state_boxplus(const double s26[26], const double d23[23], double out26[26]) {
    LioState x;
    state_from_array(s26, x);
    state_boxplus(x, d23);
    state_to_array(x, out26);
}

void lio_state_boxminus(const double a26[26], const double b26[26], double d23[23]) {
    LioState a, b;
    state_from_array(a26, a);
    state_from_array(b26, b);
    state_boxminus(a, b, d23);
}

// =====================================================================================================================
// IMU front half: fastlio_init / fastlio_imu_enqueue / fastlio_pcl_enqueue / sync_packages / ImuProcess / fastlio_main /
// fastlio_odometry / fastlio_state of laserMapping.cpp + IMU_Processing.hpp.  Host side: the buffers, IMU initialisation
// and the 23-DoF forward propagation (a dozen row-sparse 23 x 23 updates per scan); device side: the point filter and
// the per-point motion compensation (undistort.hip), writing straight into the scan buffer the VoxelGrid reads.
// =====================================================================================================================
static void frontend_destroy(lio_engine* e) {
    Frontend* f = e->fe;
    if (!f) return;
    hipSetDevice(e->scan->device);
    hipStreamSynchronize(e->scan->stream);
    if (f->copy_stream) hipStreamSynchronize(f->copy_stream);
    for (PinnedScan& b : f->pool) { hipHostFree(b.xyzi); hipFree(b.d_xyzi); hipEventDestroy(b.copied); }
    if (f->copy_stream) hipStreamDestroy(f->copy_stream);
    if (f->d_poses) hipFree(f->d_poses);
    if (f->d_first) hipFree(f->d_first);
    if (f->h_poses) hipHostFree(f->h_poses);
    delete f;
    e->fe = nullptr;
}

// Eigen's rotation-matrix -> quaternion assignment (what SO3(rotation_matrix) does for Lidar_R_wrt_IMU)
static void quat_from_matrix(const double m[9], double q[4]) {
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

static inline double norm3_eig(const double v[3]) { return sqrt(v[0] * v[0] + (v[1] * v[1] + v[2] * v[2])); }

static void fe_set_Q(const Frontend* f, double Q[12]) {
    for (int i = 0; i < 3; i++) { Q[i] = f->cov_gyr[i]; Q[3 + i] = f->cov_acc[i]; Q[6 + i] = f->cov_bias_gyr[i]; Q[9 + i] = f->cov_bias_acc[i]; }
}

// IMU_Processing.hpp:164-236
static void fe_imu_init(lio_engine* e, const std::vector<ImuSample>& imu, double lidar_end, const double* ins_vel) {
    Frontend* f = e->fe;
    int& N = f->init_iter_num;
    if (f->b_first_frame) {  // Reset()
        f->mean_acc[0] = 0; f->mean_acc[1] = 0; f->mean_acc[2] = -1.0;
        for (int i = 0; i < 3; i++) { f->mean_gyr[i] = 0; f->vel_last[i] = 0; f->angvel_last[i] = 0; }
        f->imu_need_init = true;
        f->state_init_done = false;
        f->last_imu = ImuSample{};
        N = 1;
        f->b_first_frame = false;
        for (int i = 0; i < 3; i++) { f->mean_acc[i] = imu.front().acc[i]; f->mean_gyr[i] = imu.front().gyr[i]; }
    }
    for (const ImuSample& m : imu) {
        for (int i = 0; i < 3; i++) {
            f->mean_acc[i] += (m.acc[i] - f->mean_acc[i]) / N;
            f->mean_gyr[i] += (m.gyr[i] - f->mean_gyr[i]) / N;
            f->cov_acc[i] = f->cov_acc[i] * (N - 1.0) / N + (m.acc[i] - f->mean_acc[i]) * (m.acc[i] - f->mean_acc[i]) * (N - 1.0) / (N * N);
            f->cov_gyr[i] = f->cov_gyr[i] * (N - 1.0) / N + (m.gyr[i] - f->mean_gyr[i]) * (m.gyr[i] - f->mean_gyr[i]) * (N - 1.0) / (N * N);
        }
        N++;
    }
    if (ins_vel)  // IMU_Processing.hpp:201-204: the last INS velocity of this scan is the initial velocity
        for (int i = 0; i < 3; i++) f->vel_last[i] = ins_vel[i];
    const double na = norm3_eig(f->mean_acc), ng = norm3_eig(f->mean_gyr);
    f->mean_acc_norm = na;
    if (fabs(na - 1.0) > 0.1 || ng > (10.0 / 180.0 * M_PI)) {  // "FastLIO IMU init is not stable, reset"
        f->b_first_frame = true;
        return;
    }
    LioState& x = e->kf.x;
    {   // S2(-mean_acc / |mean_acc| * G_m_s2): the S2 ctor re-normalises to length 9.809 (S2.hpp:124-127)
        double g[3] = {-f->mean_acc[0] / na * 9.81, -f->mean_acc[1] / na * 9.81, -f->mean_acc[2] / na * 9.81};
        const double z = g[0] * g[0] + (g[1] * g[1] + g[2] * g[2]);
        if (z > 0) { const double nz = sqrt(z); for (int i = 0; i < 3; i++) g[i] /= nz; }
        for (int i = 0; i < 3; i++) x.grav[i] = g[i] * kS2Length;
    }
    for (int i = 0; i < 3; i++) { x.vel[i] = f->vel_last[i]; x.bg[i] = 0; x.ba[i] = 0; x.til[i] = f->Lidar_T[i]; }
    for (int i = 0; i < 4; i++) x.ril[i] = f->Lidar_R[i];
    double* P = e->kf.P;
    for (int i = 0; i < kDof * kDof; i++) P[i] = 0;
    for (int i = 0; i < kDof; i++) P[i * kDof + i] = 1.0;
    for (int i = 6; i < 12; i++) P[i * kDof + i] = 0.00001;
    for (int i = 15; i < 18; i++) P[i * kDof + i] = 0.0001;
    for (int i = 18; i < 21; i++) P[i * kDof + i] = 0.001;
    P[21 * kDof + 21] = P[22 * kDof + 22] = 0.00001;
    f->last_imu = imu.back();
    f->last_lidar_end_time = lidar_end;
    f->start_state = x;
}

static void fe_after_predict(lio_engine* e, const double angvel_avr[3], const double acc_avr[3]) {
    Frontend* f = e->fe;
    const LioState& x = e->kf.x;
    const double amb[3] = {acc_avr[0] - x.ba[0], acc_avr[1] - x.ba[1], acc_avr[2] - x.ba[2]};
    double a[3];
    quat_rotate(x.rot, amb, a);
    for (int i = 0; i < 3; i++) { f->angvel_last[i] = angvel_avr[i] - x.bg[i]; f->acc_s_last[i] = a[i] + x.grav[i]; }
}

static void quat_to_R_eig(const double q[4], double R[9]) {  // Eigen::QuaternionBase::toRotationMatrix
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

static int fe_push_pose(lio_engine* e, double off) {
    Frontend* f = e->fe;
    if (f->n_poses >= kMaxImuPoses) { set_error("more than %d IMU samples in one scan", kMaxImuPoses - 2); return LIO_E_CAPACITY; }
    ImuPoseDev& p = f->h_poses[f->n_poses++];
    const LioState& x = e->kf.x;
    p.off = off;
    for (int i = 0; i < 3; i++) { p.acc[i] = f->acc_s_last[i]; p.gyr[i] = f->angvel_last[i]; p.vel[i] = x.vel[i]; p.pos[i] = x.pos[i]; }
    quat_to_R_eig(x.rot, p.R);
    return LIO_OK;
}

// ImuProcess::UndistortPcl (IMU_Processing.hpp:238-406): forward propagation on the host, backward propagation on the device
static int fe_undistort(lio_engine* e, const PendingScan& sc, const std::vector<ImuSample>& meas_imu, double lidar_end) {
    Frontend* f = e->fe;
    lio_scan* s = e->scan;
    const auto h0 = std::chrono::steady_clock::now();
    const float4* d_in = sc.d_xyzi;
    const uint32_t* d_stamp = sc.d_stamp;
    if (sc.pinned >= 0) {  // a host scan has been travelling since lio_fastlio_pcl_enqueue
        PinnedScan b;
        { std::lock_guard<std::mutex> lk(f->mtx); b = f->pool[sc.pinned]; }
        LIO_HIP_TRY(hipStreamWaitEvent(s->stream, b.copied, 0));
        d_in = b.d_xyzi;
        d_stamp = b.d_stamp;
    }
    const double na = norm3_eig(f->mean_acc);
    double Q[12];
    fe_set_Q(f, Q);
    if (sc.beg > f->last_lidar_end_time) {  // predict the state at the scan start time
        const double acc_avr[3] = {f->last_imu.acc[0] * 9.81 / na, f->last_imu.acc[1] * 9.81 / na, f->last_imu.acc[2] * 9.81 / na};
        e->kf.predict(sc.beg - f->last_lidar_end_time, Q, acc_avr, f->last_imu.gyr);
        fe_after_predict(e, f->last_imu.gyr, acc_avr);
        f->last_lidar_end_time = sc.beg;
    }
    f->start_state = e->kf.x;
    const double pcl_beg = sc.beg, pcl_end = lidar_end;
    f->n_poses = 0;
    int rc = fe_push_pose(e, 0.0);
    const size_t nv = meas_imu.size() + 1;  // v_imu = last_imu_ + meas.imu
    auto v_imu = [&](size_t k) -> const ImuSample& { return k == 0 ? f->last_imu : meas_imu[k - 1]; };
    const double imu_end_time = v_imu(nv - 1).stamp;
    for (size_t k = 0; k + 1 < nv && rc == LIO_OK; k++) {
        const ImuSample &head = v_imu(k), &tail = v_imu(k + 1);
        if (tail.stamp < f->last_lidar_end_time) continue;
        double angvel_avr[3], acc_avr[3];
        for (int i = 0; i < 3; i++) { angvel_avr[i] = 0.5 * (head.gyr[i] + tail.gyr[i]); acc_avr[i] = 0.5 * (head.acc[i] + tail.acc[i]) * 9.81 / na; }
        double dt = head.stamp < f->last_lidar_end_time ? tail.stamp - f->last_lidar_end_time : tail.stamp - head.stamp;
        dt = std::min(1.0, dt);
        e->kf.predict(dt, Q, acc_avr, angvel_avr);
        fe_after_predict(e, angvel_avr, acc_avr);
        rc = fe_push_pose(e, tail.stamp - pcl_beg);
    }
    if (rc != LIO_OK) return rc;
    {   // the pos and attitude prediction at the frame end
        const ImuSample& b = v_imu(nv - 1);
        const double acc_avr[3] = {b.acc[0] * 9.81 / na, b.acc[1] * 9.81 / na, b.acc[2] * 9.81 / na};
        const double note = pcl_end > imu_end_time ? 1.0 : -1.0;
        const double dt = std::min(1.0, note * (pcl_end - imu_end_time));
        e->kf.predict(dt, Q, acc_avr, b.gyr);
        fe_after_predict(e, b.gyr, acc_avr);
        rc = fe_push_pose(e, pcl_end - pcl_beg);
        if (rc != LIO_OK) return rc;
    }
    f->last_imu = meas_imu.back();
    f->last_lidar_end_time = pcl_end;
    e->tm.imu_host_us = (float)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    // backward propagation
    UndistortArgs A;
    const LioState& x = e->kf.x;
    for (int i = 0; i < 3; i++) { A.pos_e[i] = x.pos[i]; A.til[i] = x.til[i]; }
    for (int i = 0; i < 4; i++) { A.rot_e[i] = x.rot[i]; A.ril[i] = x.ril[i]; }
    A.blind2 = f->blind * f->blind;
    A.n_poses = f->n_poses;
    A.filter_num = f->max_point_num > 0 ? std::max(1, (int)sc.n / f->max_point_num) : f->filter_num;
    A.undistort = f->undistort ? 1 : 0;
    hipEvent_t t0, t1;
    if (e->timing) { t0 = e->ev[0]; t1 = e->ev[1]; hipEventRecord(t0, s->stream); }
    LIO_HIP_TRY(hipMemcpyAsync(f->d_poses, f->h_poses, sizeof(ImuPoseDev) * (size_t)f->n_poses, hipMemcpyHostToDevice, s->stream));
    rc = undistort_launch(s->stream, d_in, d_stamp, sc.n, s->raw_own, f->d_poses, A, f->d_first);
    if (e->timing) {
        hipEventRecord(t1, s->stream);
        hipEventSynchronize(t1);
        e->tm.undistort_us = ev_us(t0, t1);
    }
    if (rc != LIO_OK) return rc;
    s->raw = s->raw_own;
    s->n_raw = sc.n;
    f->have_undistorted = sc.n != 0;
    return LIO_OK;
}

static void fe_release(Frontend* f, const PendingScan& sc) {
    if (sc.pinned < 0) return;
    std::lock_guard<std::mutex> lk(f->mtx);
    f->pool_free.push_back(sc.pinned);
}

int lio_fastlio_init(lio_engine* e, const double extT[3], const double extR[9], int filter_num, int max_point_num, double scan_period, int undistort) {
    if (!e || !extT || !extR || !(scan_period > 0)) return LIO_E_INVALID;
    lio_scan* s = e->scan;
    hipSetDevice(s->device);
    frontend_destroy(e);
    Frontend* f = new Frontend();
    e->fe = f;
    for (int i = 0; i < 3; i++) f->Lidar_T[i] = extT[i];
    quat_from_matrix(extR, f->Lidar_R);
    f->filter_num = filter_num > 0 ? filter_num : 1;
    f->max_point_num = max_point_num;
    f->scan_period = scan_period;
    f->undistort = undistort != 0;
    LIO_HIP_TRY(hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking));
    LIO_HIP_TRY(hipMalloc(&f->d_poses, sizeof(ImuPoseDev) * kMaxImuPoses));
    LIO_HIP_TRY(hipMalloc(&f->d_first, sizeof(unsigned long long) * ((size_t)s->max_raw / 256 + 1)));  // workgroup minima of undistort_kernel
    LIO_HIP_TRY(hipHostMalloc(&f->h_poses, sizeof(ImuPoseDev) * kMaxImuPoses, hipHostMallocDefault));
    // the file-scope state fastlio_init resets (laserMapping.cpp:1033-1046,1107-1110)
    e->kf = Eskf();
    e->travel = 0;
    e->first_lidar_time = 0;
    e->flg_first_scan = true;
    e->flg_EKF_inited = false;
    e->is_degenerate = false;
    for (int i = 0; i < 3; i++) e->last_pos_lid[i] = 0;
    f->start_state = e->kf.x;
    f->end_state = e->kf.x;
    lio_scan_reset(s);
    // ivox_options of fastlio_init (laserMapping.cpp:1060-1064): capacity_ 100000 voxels, max_distance_ 100 m.  Only an engine
    // that owns a still empty map with room above the soft capacity gets the LRU list; otherwise the map keeps everything.
    if (e->own_map && e->map->n_batches == 0 && e->map->lru_capacity == 0 && e->map->max_voxels > 100000)
        return lio_map_set_lru(e->map, 100000, 100.0);
    return LIO_OK;
}

int lio_fastlio_is_init(lio_engine* e) { return e && e->fe && e->fe->state_init_done ? 1 : 0; }  // ImuProcess::IsInit: state_init_done_

int lio_fastlio_imu_enqueue(lio_engine* e, double stamp, const double gyr[3], const double acc_ms2[3]) {
    if (!e || !e->fe || !gyr || !acc_ms2) return LIO_E_INVALID;
    ImuSample m;
    m.stamp = stamp;
    for (int i = 0; i < 3; i++) { m.gyr[i] = gyr[i]; m.acc[i] = acc_ms2[i] / 9.81; }
    std::lock_guard<std::mutex> lk(e->fe->mtx);
    e->fe->imu_buffer.push_back(m);
    return LIO_OK;
}

// fastlio_ins_enqueue (laserMapping.cpp:417-441) after its ENU -> ego -> IMU rotation: the caller passes the velocity in the
// IMU frame (the reference zeroes its third component, :436); consumed by IMU initialisation only (wheelspeed_en == false)
int lio_fastlio_set_wheelspeed(lio_engine* e, int enable) {
    if (!e) return LIO_E_INVALID;
    e->wheelspeed_en = enable != 0;
    return LIO_OK;
}

int lio_engine_set_device_loop(lio_engine* e, int on) {
    if (!e) return LIO_E_INVALID;
    e->device_loop = on ? 1 : 0;
    return LIO_OK;
}

int lio_fastlio_ins_enqueue(lio_engine* e, double stamp, const double vel_imu[3]) {
    if (!e || !e->fe || !vel_imu) return LIO_E_INVALID;
    std::lock_guard<std::mutex> lk(e->fe->mtx);
    e->fe->ins_buffer.push_back({stamp, {vel_imu[0], vel_imu[1], vel_imu[2]}});
    return LIO_OK;
}

// a staging slot (pinned host buffer + device twin + event): recycled ones first, the pool grows on demand
static int fe_take_slot(lio_engine* e, int* slot_out) {
    Frontend* f = e->fe;
    lio_scan* s = e->scan;
    int slot = -1;
    {
        std::lock_guard<std::mutex> lk(f->mtx);
        if (!f->pool_free.empty()) { slot = f->pool_free.back(); f->pool_free.pop_back(); }
    }
    hipSetDevice(s->device);
    if (slot < 0) {
        PinnedScan b;
        void *mem = nullptr, *dmem = nullptr;
        const size_t bytes = (size_t)s->max_raw * (sizeof(float4) + sizeof(uint32_t));
        LIO_HIP_TRY(hipHostMalloc(&mem, bytes, hipHostMallocDefault));
        LIO_HIP_TRY(hipMalloc(&dmem, bytes));
        LIO_HIP_TRY(hipEventCreateWithFlags(&b.copied, hipEventDisableTiming));
        b.xyzi = static_cast<float4*>(mem);
        b.stamp = reinterpret_cast<uint32_t*>(b.xyzi + s->max_raw);
        b.d_xyzi = static_cast<float4*>(dmem);
        b.d_stamp = reinterpret_cast<uint32_t*>(b.d_xyzi + s->max_raw);
        std::lock_guard<std::mutex> lk(f->mtx);
        f->pool.push_back(b);
        slot = (int)f->pool.size() - 1;
    }
    *slot_out = slot;
    return LIO_OK;
}
// the slot's content is complete: start its trip to the device on the copy stream and queue the scan
static int fe_send_slot(lio_engine* e, int slot, uint32_t n, double header_stamp) {
    Frontend* f = e->fe;
    PinnedScan b;
    { std::lock_guard<std::mutex> lk(f->mtx); b = f->pool[slot]; }
    hipSetDevice(e->scan->device);
    if (n) {
        LIO_HIP_TRY(hipMemcpyAsync(b.d_xyzi, b.xyzi, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, f->copy_stream));
        LIO_HIP_TRY(hipMemcpyAsync(b.d_stamp, b.stamp, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, f->copy_stream));
    }
    LIO_HIP_TRY(hipEventRecord(b.copied, f->copy_stream));
    PendingScan sc;
    sc.beg = header_stamp;
    sc.n = n;
    sc.pinned = slot;
    std::lock_guard<std::mutex> lk(f->mtx);
    f->lidar_buffer.push_back(sc);
    return LIO_OK;
}
int lio_fastlio_pcl_stage(lio_engine* e, uint32_t n, float** xyzi, uint32_t** stamp_us) {
    if (!e || !e->fe || !xyzi || !stamp_us) return LIO_E_INVALID;
    if (n > e->scan->max_raw) { set_error("scan of %u points exceeds max_raw %u", n, e->scan->max_raw); return LIO_E_CAPACITY; }
    if (e->fe->staged >= 0) { set_error("lio_fastlio_pcl_stage: the previous staged scan was not committed"); return LIO_E_STATE; }
    int slot = -1;
    const int rc = fe_take_slot(e, &slot);
    if (rc != LIO_OK) return rc;
    e->fe->staged = slot;
    PinnedScan b;
    { std::lock_guard<std::mutex> lk(e->fe->mtx); b = e->fe->pool[slot]; }
    *xyzi = reinterpret_cast<float*>(b.xyzi);
    *stamp_us = b.stamp;
    return LIO_OK;
}
int lio_fastlio_pcl_commit(lio_engine* e, uint32_t n, double header_stamp) {
    if (!e || !e->fe) return LIO_E_INVALID;
    if (e->fe->staged < 0) { set_error("lio_fastlio_pcl_commit: nothing staged"); return LIO_E_STATE; }
    if (n > e->scan->max_raw) return LIO_E_CAPACITY;
    const int slot = e->fe->staged;
    e->fe->staged = -1;
    return fe_send_slot(e, slot, n, header_stamp);
}

static int fe_enqueue(lio_engine* e, const float* xyzi, const uint32_t* stamp_us, uint32_t n, double header_stamp, bool device) {
    if (!e || !e->fe || (n && (!xyzi || !stamp_us))) return LIO_E_INVALID;
    Frontend* f = e->fe;
    lio_scan* s = e->scan;
    if (n > s->max_raw) { set_error("scan of %u points exceeds max_raw %u", n, s->max_raw); return LIO_E_CAPACITY; }
    PendingScan sc;
    sc.beg = header_stamp;
    sc.n = n;
    if (device) {
        sc.d_xyzi = reinterpret_cast<const float4*>(xyzi);
        sc.d_stamp = stamp_us;
    } else {
        int slot = -1;
        const int rc = fe_take_slot(e, &slot);
        if (rc != LIO_OK) return rc;
        PinnedScan b;
        { std::lock_guard<std::mutex> lk(f->mtx); b = f->pool[slot]; }
        memcpy(b.xyzi, xyzi, (size_t)n * sizeof(float4));
        memcpy(b.stamp, stamp_us, (size_t)n * sizeof(uint32_t));
        return fe_send_slot(e, slot, n, header_stamp);
    }
    std::lock_guard<std::mutex> lk(f->mtx);
    f->lidar_buffer.push_back(sc);
    return LIO_OK;
}
int lio_fastlio_pcl_enqueue(lio_engine* e, const float* xyzi, const uint32_t* stamp_us, uint32_t n, double header_stamp) {
    return fe_enqueue(e, xyzi, stamp_us, n, header_stamp, false);
}
int lio_fastlio_pcl_enqueue_device(lio_engine* e, const void* d_xyzi, const void* d_stamp_us, uint32_t n, double header_stamp) {
    return fe_enqueue(e, static_cast<const float*>(d_xyzi), static_cast<const uint32_t*>(d_stamp_us), n, header_stamp, true);
}

// fastlio_main in two halves around process_core, so that the batched sequence mode (batch.hip: lio_batch_fastlio_main) can run the middle of
// many sessions as one round.  The front half: sync_packages, the first-scan / IMU-initialisation returns, forward propagation + undistortion
// (enqueued on the scan's stream).  kFrontReady: the scan in `sc` is ready for process_core(e, sc.beg).
constexpr int kFrontReady = 1000;
static int fastlio_front(lio_engine* e, PendingScan& sc) {
    Frontend* f = e->fe;
    lio_scan* s = e->scan;
    std::vector<ImuSample> meas_imu;
    double lidar_end;
    double ins_vel[3] = {0, 0, 0};
    bool have_ins = false;
    {   // sync_packages (laserMapping.cpp:445-520)
        std::lock_guard<std::mutex> lk(f->mtx);
        if (f->lidar_buffer.empty() || f->imu_buffer.empty()) return LIO_MAIN_IDLE;
        sc = f->lidar_buffer.front();
        f->lidar_buffer.pop_front();
        lidar_end = sc.beg + f->scan_period;
        while (!f->imu_buffer.empty() && !(f->imu_buffer.front().stamp > lidar_end)) {
            meas_imu.push_back(f->imu_buffer.front());
            f->imu_buffer.pop_front();
        }
        while (!f->ins_buffer.empty() && !(f->ins_buffer.front().first > lidar_end)) {  // meas.ins (laserMapping.cpp:495-503)
            for (int i = 0; i < 3; i++) ins_vel[i] = f->ins_buffer.front().second[i];
            e->meas_ins_stamp = f->ins_buffer.front().first;
            have_ins = true;
            f->ins_buffer.pop_front();
        }
        e->meas_ins_valid = have_ins;  // meas.ins is rebuilt for every scan
        for (int i = 0; i < 3; i++) e->meas_ins_vel[i] = ins_vel[i];
        e->meas_lidar_end = lidar_end;
    }
    hipSetDevice(s->device);
    memset(&e->tm, 0, sizeof(e->tm));
    if (e->flg_first_scan) {  // laserMapping.cpp:1171-1177
        e->first_lidar_time = sc.beg;
        e->flg_first_scan = false;
        fe_release(f, sc);
        return LIO_MAIN_FIRST_SCAN;
    }
    if (meas_imu.empty()) {
        // ImuProcess::Process returns at once: feats_undistort still holds the PREVIOUS scan's cloud, which fastlio_main
        // then registers again (IMU_Processing.hpp:413, laserMapping.cpp:1189-1197)
        fe_release(f, sc);
        sc.pinned = -1;  // (released: the back half has nothing to give back)
        if (!f->have_undistorted) return LIO_MAIN_IMU_INIT;
        return kFrontReady;
    }
    if (f->imu_need_init) {  // IMU_Processing.hpp:416-441
        fe_imu_init(e, meas_imu, lidar_end, have_ins ? ins_vel : nullptr);
        f->imu_need_init = true;
        f->last_imu = meas_imu.back();
        if (f->init_iter_num > 100) {  // MAX_INI_COUNT
            f->imu_need_init = false;
            for (int i = 0; i < 3; i++) { f->cov_acc[i] = f->cov_acc_scale[i]; f->cov_gyr[i] = f->cov_gyr_scale[i]; }
        }
        fe_release(f, sc);
        return LIO_MAIN_IMU_INIT;
    }
    f->state_init_done = true;  // IMU_Processing.hpp:443: set by the first scan that is undistorted, one scan after imu_need_init_ clears
    const int rc = fe_undistort(e, sc, meas_imu, lidar_end);
    if (rc != LIO_OK) { fe_release(f, sc); return rc; }
    return kFrontReady;
}
// ... and what follows process_core
static void fastlio_back(lio_engine* e, const PendingScan& sc) {
    Frontend* f = e->fe;
    if (sc.pinned >= 0) hipStreamSynchronize(e->scan->stream);  // the staging buffer goes back to the pool only after its copy ran
    fe_release(f, sc);
    f->end_state = e->kf.x;
}

int lio_fastlio_main(lio_engine* e) {
    if (!e || !e->fe) return LIO_E_INVALID;
    PendingScan sc;
    const int fr = fastlio_front(e, sc);
    if (fr != kFrontReady) return fr;
    const int rc = process_core(e, sc.beg);
    fastlio_back(e, sc);
    return rc;
}

// the two halves for the sequence batch (internal; batch.hip).  engine_fastlio_front: LIO_MAIN_* / an error for a scan that ends in the front
// half, else 1000 with the scan ready in the engine's own buffers -- the undistortion has been WAITED for (the round reads the cloud from
// another stream) -- and its registration inputs in the out arguments; the held scan is remembered in the engine until engine_fastlio_back.
int engine_fastlio_front(lio_engine* e, const void** d_raw, uint32_t* n_raw, double* beg, double state26[26], double cov529[529]) {
    if (!e || !e->fe) return LIO_E_INVALID;
    if (!e->held) e->held = new lio_held_scan();
    const int fr = fastlio_front(e, e->held->sc);
    if (fr != kFrontReady) return fr;
    lio_scan* s = e->scan;
    if (hipStreamSynchronize(s->stream) != hipSuccess) { set_error("fastlio front half: %s", hipGetErrorString(hipGetLastError())); fe_release(e->fe, e->held->sc); return LIO_E_DEVICE; }
    *d_raw = s->raw;
    *n_raw = s->n_raw;
    *beg = e->held->sc.beg;
    state_to_array(e->kf.x, state26);
    memcpy(cov529, e->kf.P, sizeof(double) * 529);
    return kFrontReady;
}
void engine_fastlio_back(lio_engine* e) {
    if (e && e->fe && e->held) fastlio_back(e, e->held->sc);
}

static void odom_matrix(const LioState& x, double T[16]) {  // pos + Quaterniond(rot).normalized().toRotationMatrix()
    double q[4] = {x.rot[0], x.rot[1], x.rot[2], x.rot[3]};
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (n2 > 0) { const double n = sqrt(n2); for (int i = 0; i < 4; i++) q[i] /= n; }
    double R[9];
    quat_to_R_eig(q, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = x.pos[i];
        T[12 + i] = 0;
    }
    T[15] = 1;
}
int lio_fastlio_odometry(lio_engine* e, double odom_s[16], double odom_e[16]) {
    if (!e || !e->fe || !odom_s || !odom_e) return LIO_E_INVALID;
    odom_matrix(e->fe->start_state, odom_s);
    odom_matrix(e->kf.x, odom_e);
    return LIO_OK;
}
int lio_fastlio_state(lio_engine* e, double out[20]) {  // laserMapping.cpp:714-738: start_state_point + mean_acc_norm
    if (!e || !e->fe || !out) return LIO_E_INVALID;
    const LioState& x = e->fe->start_state;
    for (int i = 0; i < 3; i++) { out[i] = x.pos[i]; out[7 + i] = x.vel[i]; out[10 + i] = x.ba[i]; out[13 + i] = x.bg[i]; out[16 + i] = x.grav[i]; }
    for (int i = 0; i < 4; i++) out[3 + i] = x.rot[i];
    out[19] = e->fe->mean_acc_norm;
    return LIO_OK;
}
int lio_fastlio_start_state(lio_engine* e, double s26[26]) {
    if (!e || !e->fe || !s26) return LIO_E_INVALID;
    state_to_array(e->fe->start_state, s26);
    return LIO_OK;
}
int lio_fastlio_download_undistorted(lio_engine* e, float* out_xyzi, uint32_t cap, uint32_t* n) {
    if (!e || !e->fe || !n) return LIO_E_INVALID;
    lio_scan* s = e->scan;
    *n = e->fe->have_undistorted ? s->n_raw : 0;
    if (*n > cap) return LIO_E_CAPACITY;
    if (*n && !out_xyzi) return LIO_E_INVALID;
    hipSetDevice(s->device);
    if (*n) {
        LIO_HIP_TRY(hipMemcpyAsync(out_xyzi, s->raw, (size_t)*n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
        LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return LIO_OK;
}
int lio_state_predict(const double s26[26], const double P[529], double dt, const double Q[12], const double acc[3], const double gyro[3],
                      double s26_out[26], double P_out[529]) {
    if (!s26 || !P || !Q || !acc || !gyro || !s26_out || !P_out) return LIO_E_INVALID;
    Eskf kf;
    state_from_array(s26, kf.x);
    memcpy(kf.P, P, sizeof(double) * 529);
    kf.predict(dt, Q, acc, gyro);
    state_to_array(kf.x, s26_out);
    memcpy(P_out, kf.P, sizeof(double) * 529);
    return LIO_OK;
}

// Host-only: one esekf::update_iterated_dyn_share_modified (esekfom.hpp:1619-1931) on the product's filter with the
// measurement model supplied by the caller -- fn(ctx, state26, converge, &n, rows n x 6 (first six columns of h_x), h n,
// cap) -> valid.  The device path hands the filter the 6 x 6 normal equations; here they are summed from the rows.  Used
// by the CPU tests that pin the filter algebra against the reference's own IKFoM code.
int lio_eskf_update_cb(const double s26[26], const double P[529], double R, int max_iter, lio_meas_fn fn, void* ctx, int cap, double s26_out[26],
                       double P_out[529]) {
    return lio_eskf_update_ws_cb(s26, P, R, max_iter, fn, ctx, cap, nullptr, 0, s26_out, P_out);
}

// ... with the wheel-speed rows of laserMapping.cpp:794-811 appended in every pass when ins_vel is given (Measures.ins.back() in the IMU frame)
int lio_eskf_update_ws_cb(const double s26[26], const double P[529], double R, int max_iter, lio_meas_fn fn, void* ctx, int cap, const double* ins_vel,
                          int degenerate, double s26_out[26], double P_out[529]) {
    if (!s26 || !P || !fn || !s26_out || !P_out || cap <= 0 || max_iter < 0) return LIO_E_INVALID;
    Eskf kf;
    state_from_array(s26, kf.x);
    memcpy(kf.P, P, sizeof(double) * 529);
    kf.maximum_iter = max_iter;
    std::vector<double> rows((size_t)cap * 6), hv(cap);
    StaleRows stale;  // the keeper run_update uses: a model that reports "no effective points" (return value 2) gets the previous pass's rows back
    auto measure = [&](const LioState& x, bool converge, Measurement& m) {
        double st[26];
        state_to_array(x, st);
        int n = 0;
        const int rc = fn(ctx, st, converge ? 1 : 0, &n, rows.data(), hv.data(), cap);
        m.valid = false;
        m.n_rows = 0;
        m.n_geo = 0;
        m.ws_n = 0;
        m.rows6 = nullptr;
        m.h = nullptr;
        if (rc == 2) {
            // h_share_model_geometric returned early ("No Effective Points", laserMapping.cpp:888-893): the copy of the shared struct keeps what
            // the previous pass left in it
            stale.take_over(m);
        } else if (rc && n > 0 && n <= cap) {
            m.valid = true;
            m.n_rows = n;
            for (int a = 0; a < 36; a++) m.HTH[a] = 0;
            for (int a = 0; a < 6; a++) m.HTh[a] = 0;
            for (int r = 0; r < n; r++)
                for (int a = 0; a < 6; a++) {
                    for (int b = 0; b < 6; b++) m.HTH[a * 6 + b] += rows[(size_t)r * 6 + a] * rows[(size_t)r * 6 + b];
                    m.HTh[a] += rows[(size_t)r * 6 + a] * hv[r];
                }
            m.rows6 = rows.data();
            m.h = hv.data();
        } else {
            return;  // the model reports the whole pass invalid
        }
        if (ins_vel) append_wheelspeed_rows(degenerate != 0, ins_vel, x, m);
        stale.remember(m);
    };
    kf.update_iterated(R, measure, nullptr, nullptr);
    state_to_array(kf.x, s26_out);
    memcpy(P_out, kf.P, sizeof(double) * 529);
    return LIO_OK;
}

}  // extern "C"