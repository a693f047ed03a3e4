// engine.hip -- the per-scan driver: what fastlio_main() does after p_imu->Process
// (/root/reference/slam/mapping/fastlio/src/laserMapping.cpp:1189-1304), as host C++ over the kernel-level
// C ABI.  Constants follow fastlio_init (laserMapping.cpp:1025-1124): 4 (+1) filter passes, leaf 0.5 m for
// both filters, INIT_TIME 0.1 s, LASER_POINT_COV 0.001, degeneracy detection on, extrinsic estimation off.
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "eskf.h"
#include "lio_common.h"

using namespace lio;

struct lio_engine {
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
    // file-scope state of laserMapping.cpp
    double travel = 0, first_lidar_time = 0;
    double last_pos_lid[3] = {0, 0, 0};
    bool flg_first_scan = true, flg_EKF_inited = false, is_degenerate = false;
    std::vector<lio_pass_log> log;
    // timing
    bool timing = false;
    hipEvent_t ev[8];
    lio_timings tm;
    std::vector<double> rows6, hvec;
    lio_reduce_fn reduce = nullptr;  // cross-GPU reduction of the normal equations (joint registration)
    void* reduce_ctx = nullptr;
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
    if (e->timing) { hipEventCreate(&t0); hipEventCreate(&t1); hipEventRecord(t0, e->scan->stream); }
    const int rc = lio_p2plane_linearize(e->map, e->scan, pose, ext, converge ? 1 : 0, &ne);
    if (e->timing) {
        hipEventRecord(t1, e->scan->stream);
        hipEventSynchronize(t1);
        const float us = ev_us(t0, t1);
        if (converge) e->tm.knn_us += us; else e->tm.linearize_us += us;
        hipEventDestroy(t0); hipEventDestroy(t1);
    }
    e->tm.n_pass++;
    if (converge) e->tm.n_knn_pass++;
    if (rc != LIO_OK) return rc;
    if (e->reduce) {
        // joint registration: this rank's sums -> global sums (fixed rank order inside the hook), then the
        // degeneracy logic on the GLOBAL eigen-structure
        double buf[29];
        int t = 0;
        for (int a = 0; a < 6; a++)
            for (int c = a; c < 6; c++) buf[t++] = ne.JtJ[a * 6 + c];
        for (int a = 0; a < 6; a++) buf[21 + a] = ne.Jtr[a];
        buf[27] = ne.sum_abs_res;
        buf[28] = (double)ne.n_eff;
        e->reduce(e->reduce_ctx, buf, 29);
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
        if (need && ne.n_eff > 0) {
            double cs[6];
            const int r2 = lio_p2plane_degeneracy(e->scan, ne.eigvec, cs, cs + 3);
            if (r2 != LIO_OK) return r2;
            e->reduce(e->reduce_ctx, cs, 6);
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

int run_update(lio_engine* e) {
    e->log.clear();
    PassCtx ctx{e, LIO_OK};
    double host_us = 0;
    // ekfom_data_geo is a COPY of the shared struct (laserMapping.cpp:991): when a later pass finds no
    // effective point, the rows of the previous pass survive in it and the filter re-uses them (n_terms != 0).
    Measurement prev;
    std::vector<double> prev_rows, prev_h;
    bool have_prev = false;
    auto measure = [&](const LioState& x, bool converge, Measurement& m) {
        lio_pass_log pl;
        const int rc = measure_pass(e, x, converge, m, pl);
        if (rc != LIO_OK) { ctx.rc = rc; m.valid = false; e->log.push_back(pl); return; }
        if (!m.valid && have_prev) {
            m = prev;
            if (prev.rows6) { m.rows6 = prev_rows.data(); m.h = prev_h.data(); }
            pl.valid = 1;
            memcpy(pl.JtJ, m.HTH, sizeof(pl.JtJ));
            memcpy(pl.Jtr, m.HTh, sizeof(pl.Jtr));
        } else if (m.valid) {
            prev = m;
            if (m.rows6) { prev_rows.assign(m.rows6, m.rows6 + (size_t)m.n_rows * 6); prev_h.assign(m.h, m.h + m.n_rows); }
            have_prev = true;
        }
        e->log.push_back(pl);
    };
    const auto t0 = std::chrono::steady_clock::now();
    e->kf.update_iterated(e->laser_cov, measure, on_pass, &ctx);
    host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    e->tm.host_solve_us = (float)host_us;  // wall time of the whole iterated update (device passes included)
    return ctx.rc;
}

}  // namespace

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
    e->own_map = false;
    e->static_map = true;  // several engines read one map concurrently: nobody inserts
    e->map_seeded = true;
    memset(&e->tm, 0, sizeof(e->tm));
    return e;
}

void lio_engine_destroy(lio_engine* e) {
    if (!e) return;
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

int lio_engine_enable_timing(lio_engine* e, int on) { if (!e) return LIO_E_INVALID; e->timing = on != 0; return LIO_OK; }
int lio_engine_set_reduce_hook(lio_engine* e, lio_reduce_fn fn, void* ctx) {
    if (!e) return LIO_E_INVALID;
    e->reduce = fn;
    e->reduce_ctx = ctx;
    // with a hook the degeneracy sums are evaluated here, after the reduction, never inside linearize
    return lio_scan_set_degeneracy_mode(e->scan, fn ? 2 : 0);
}

int lio_engine_set_static_map(lio_engine* e, int on) {
    if (!e) return LIO_E_INVALID;
    if (!e->own_map && !on) { set_error("an engine on a shared map is read-only"); return LIO_E_STATE; }
    e->static_map = on != 0;
    return LIO_OK;
}
int lio_engine_timings(lio_engine* e, lio_timings* out) { if (!e || !out) return LIO_E_INVALID; *out = e->tm; return LIO_OK; }

static int process_common(lio_engine* e, double lidar_beg_time) {
    lio_scan* s = e->scan;
    const auto w0 = std::chrono::steady_clock::now();
    memset(&e->tm, 0, sizeof(e->tm));
    if (e->flg_first_scan) {  // laserMapping.cpp:1171-1177
        e->first_lidar_time = lidar_beg_time;
        e->flg_first_scan = false;
        return 0;
    }
    if (s->n_raw == 0) return 2;  // "FastLio undistort points is empty"
    e->flg_EKF_inited = (lidar_beg_time - e->first_lidar_time) < e->init_time ? false : true;
    hipEvent_t t0, t1;
    if (e->timing) { hipEventCreate(&t0); hipEventCreate(&t1); hipEventRecord(t0, s->stream); }
    uint32_t n_ds = 0;
    int rc = lio_scan_voxel_downsample(s, e->leaf_surf, 1, &n_ds);
    if (e->timing) {
        hipEventRecord(t1, s->stream);
        hipEventSynchronize(t1);
        e->tm.downsample_us = ev_us(t0, t1);
        hipEventDestroy(t0); hipEventDestroy(t1);
    }
    if (rc != LIO_OK) return rc;
    e->tm.n_ds = (int)n_ds;
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
    if (e->timing) { hipEventCreate(&t0); hipEventCreate(&t1); hipEventRecord(t0, s->stream); }
    rc = lio_map_incremental(e->map, s, pose, ext, e->leaf_map, e->flg_EKF_inited ? 1 : 0, e->travel);
    if (e->timing) {
        hipEventRecord(t1, s->stream);
        hipEventSynchronize(t1);
        e->tm.insert_us = ev_us(t0, t1);
        hipEventDestroy(t0); hipEventDestroy(t1);
    }
    if (rc < 0) return rc;
    e->tm.n_added = rc;
    e->tm.total_device_us = e->tm.downsample_us + e->tm.knn_us + e->tm.linearize_us + e->tm.insert_us;
    e->tm.total_wall_us = (float)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
    return 3;
}

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
            if (job.state_in && job.cov_in) {
                state_from_array(job.state_in, e->kf.x);
                memcpy(e->kf.P, job.cov_in, sizeof(double) * 529);
                rc = lio_engine_process_scan_device(e, job.d_raw, job.n_raw, job.lidar_beg_time);
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

}  // extern "C"
