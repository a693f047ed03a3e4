// batch.hip -- throughput mode: B scans per launch, iterate loop on the device (lio_batch_* of include/lio_hip.h).
//
// What fastlio_main does per scan after IMU processing (/root/reference/slam/mapping/fastlio/src/laserMapping.cpp:1189-1304, static
// map: no map_incremental) for B independent scans at once.  Round-1's throughput mode ran one engine + host thread + stream per scan
// in flight: ~25 launches and 4.75 host hand-overs per scan, each launch on a 5-15 us latency floor with a few hundred workgroups.
// Here a round of B scans is ONE blind submission: every kernel of the chain is launched once with blockIdx.y = slot, the filter
// lives on the device (eskf_dev.h), and the only host work per scan is filling 9 KB of staging and reading a 300-byte result record.
#include <sched.h>
#include <stdlib.h>

#include <chrono>
#include <deque>
#include <vector>

#include "eskf.h"
#include "lio_common.h"

using namespace lio;

int engine_resume_update(lio_engine* e, const double* x_now26, const double* x_prop26, const double* P_prop, int i, int converge, int t);  // engine.hip
void engine_count_passes(lio_engine* e, int* n_pass, int* n_knn);
int engine_joint_register_device(lio_engine* e, const void* d_raw, uint32_t n_raw, double lidar_beg_time, double state26[26], double cov[529]);
struct lio_comm;
extern "C" int lio_comm_world(const lio_comm*);
extern "C" {  // engine.hip: the host half of fastlio_main around a sequence round
int engine_seq_prepare(lio_engine* e, uint32_t n_raw, double lidar_beg_time, int* ekf_inited);
void engine_seq_params(lio_engine* e, double* laser_cov, int* maximum_iter, int* degenerate_detect_en, float* leaf_surf, float* leaf_map, int* static_map,
                       double* travel, double last_pos_lid[3]);
int engine_seq_finish(lio_engine* e, const double x26[26], const double P529[529], int n_ds, int n_pass, int n_knn, int n_eff, int degenerate, int inserted,
                      uint32_t n_add, uint32_t map_err, uint32_t bound);
int engine_seq_resume(lio_engine* e, const double* x_now26, const double* x_prop26, const double* P_prop, int i, int converge, int t);
int engine_seq_has_lru(lio_engine* e);
int engine_fastlio_front(lio_engine* e, const void** d_raw, uint32_t* n_raw, double* beg, double state26[26], double cov529[529]);
void engine_fastlio_back(lio_engine* e);
}

namespace {

struct Group {
    hipStream_t stream = nullptr;
    std::vector<lio_engine*> eng;
    std::vector<lio_engine*> sub_eng;    // joint mode: the scan buffer sets of the further local sub-maps, [(m - 1) * B + slot]
    double* d_local32 = nullptr;         // joint mode: this rank's record per slot (B x 32), and every rank's (world x B x 32)
    double* d_gathered = nullptr;
    char* d_ds_send = nullptr;           // joint mode over several ranks: this rank's share of the round's downsampled clouds (slot chunks), and every rank's
    char* d_ds_all = nullptr;
    int ds_world = 0;                    // ... the world size the two were sized for
    MapRef* d_rowmaps = nullptr;         // joint mode: the sub-map behind every descriptor row [m * B + slot] -- ONE neighbour-search launch per pass
    std::vector<MapRef> rowmap_rows;  // ... one row per sub-map as uploaded (refreshed when any word of a sub-map's row changed)
    char* d_block = nullptr;             // [SlotDesc x B (x sub-maps)][EskfDev x B]: one upload per round
    char* h_block = nullptr;             // pinned staging, same layout
    size_t block_bytes = 0;
    SlotDesc* d_desc = nullptr;
    SlotDesc* h_desc = nullptr;
    EskfDev* d_ctrl = nullptr;
    EskfDev* h_ctrl = nullptr;
    hipGraphExec_t exec[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // the whole round as a graph, per number of radix passes
    const void* graph_table = nullptr;   // the graphs hold the map's table address and stencil by value: dropped when either changes
    int graph_stencil = 0;
    lio_batch_result* h_res = nullptr;   // pinned, mapped
    lio_batch_result* h_res_dev = nullptr;
    std::vector<int> job_of_slot;
    float4* d_rawbuf = nullptr;          // LIO_JOB_HOST_RAW: the round's clouds copied from host memory, [slot][max_raw]; allocated with the first such job
    // sequence mode: the slots' own maps as the kernels see them (part of the uploaded block), the insert half's records (device + pinned copy)
    MapRef* d_maps = nullptr;
    MapRef* h_maps = nullptr;
    SeqDev* d_seq = nullptr;
    SeqDev* h_seq = nullptr;
    std::vector<int> seq_path;           // per slot, this round: what engine_seq_prepare said (0, 2, 10, 11) or < 0
    int any_lru = 0;
    int n_active = 0;
    int launched_passes = 4;
    uint32_t seq = 0;
    uint32_t max_n_raw = 0;
    BatchTimer* bt = nullptr;
};

}  // namespace

struct lio_batch {
    lio_map* map = nullptr;
    int device = 0;
    int n_slots = 0;
    uint32_t max_raw = 0, max_ds = 0;
    int pred_passes = 4;  // radix passes the last rounds needed
    bool broken = false;  // sequence mode: a step failed after some of its rounds had run on the device (lio_batch_sequences_step refuses further steps)
    int use_graph = 1;    // LIO_BATCH_GRAPH=0: plain launches instead of one hipGraphLaunch per round
    int count_touched = 0;  // lio_batch_enable_kernel_timing(b, 2): the kNN kernel's diagnostic variant that also counts the points it loads
    // joint mode (lio_batch_create_joint): every job is registered against all of `maps` (sub-maps on this GPU; maps[0] == map) and, through
    // `comm`, against the sub-maps of the other ranks
    std::vector<lio_map*> maps;
    lio_comm* comm = nullptr;
    int world = 1;
    int rank = 0;
    int split_ds = 1;     // the downsample of a joint round divided among the ranks, clouds all-gathered (LIO_JOINT_SPLIT_DS=0: every rank downsamples every scan)
    uint32_t ds_cap = 0;  // ... points per slot chunk of that all-gather: 1.25 x the largest cloud seen so far (0: not chosen yet), the same on every rank
    uint32_t ds_seen_max = 0;
    bool ds_cap_full = false;  // the re-run of a job whose cloud was cut: chunks of the buffers' full size
    uint64_t n_ds_cut = 0;     // jobs re-run for that reason
    int (*gather_hook)(void*, const double*, double*, uint32_t, void*) = nullptr;  // lio_batch_set_gather_hook
    void* gather_ctx = nullptr;
    bool joint = false;
    bool sequences = false;  // lio_batch_create_sequences: every slot is a SLAM session with its own map (the slot's engine owns it)
    std::vector<lio_scan_job> fl_jobs;   // lio_batch_fastlio_main: the round's jobs and their priors / posteriors
    std::vector<double> fl_state, fl_cov;
    std::vector<Group> groups;
    double t_submit = 0, t_wait = 0, t_collect = 0;  // host seconds (LIO_BATCH_PROFILE=1 prints them when the object is destroyed)
    uint64_t n_rounds = 0;
};

static void fill_mapref(MapRef& r, const lio_map* m);

namespace {

// joint mode: the MapRef of every descriptor row (sub-map m, slot s) on the device, so that ONE launch of the neighbour search serves all local
// sub-maps of a pass (knn_seq_kernel, the kernel of the sequence batch: table / pool / counters from the row's MapRef).  Null when the sub-maps do
// not share one stencil (the per-sub-map launches stay then).
const MapRef* joint_rowmaps(lio_batch* b, Group& g) {
    const int M = (int)b->maps.size(), B = b->n_slots;
    if (M < 2) return nullptr;
    for (int m = 1; m < M; m++)
        if (b->maps[m]->stencil_id != b->maps[0]->stencil_id || b->maps[m]->stencil.n != b->maps[0]->stencil.n) return nullptr;
    // the cached device rows are fresh only if EVERY word of every sub-map's row is what it would be now (table, mask, pool, counters, stencil id,
    // tie mode ...: lio_map_set_stencil and a table rebuild both change a row without changing the other's key)
    std::vector<MapRef> now((size_t)M);
    for (int m = 0; m < M; m++) {
        memset(&now[m], 0, sizeof(MapRef));
        fill_mapref(now[m], b->maps[m]);
    }
    bool fresh = g.d_rowmaps != nullptr && (int)g.rowmap_rows.size() == M;
    for (int m = 0; m < M && fresh; m++) fresh = memcmp(&g.rowmap_rows[m], &now[m], sizeof(MapRef)) == 0;
    if (fresh) return g.d_rowmaps;
    std::vector<MapRef> rows((size_t)M * B);
    for (int m = 0; m < M; m++)
        for (int s = 0; s < B; s++) rows[(size_t)m * B + s] = now[m];
    if (!g.d_rowmaps && hipMalloc(reinterpret_cast<void**>(&g.d_rowmaps), sizeof(MapRef) * rows.size()) != hipSuccess) { g.d_rowmaps = nullptr; return nullptr; }
    // (stream-ordered behind the round before it, which may still read the old table)
    if (hipMemcpyAsync(g.d_rowmaps, rows.data(), sizeof(MapRef) * rows.size(), hipMemcpyHostToDevice, g.stream) != hipSuccess) return nullptr;
    if (hipStreamSynchronize(g.stream) != hipSuccess) return nullptr;  // (rows is a local: rare -- creation, a rebuilt table, another stencil)
    g.rowmap_rows = now;
    return g.d_rowmaps;
}

void fill_desc(SlotDesc& d, lio_scan* sc, EskfDev* d_ctrl, lio_batch_result* d_res) {
    d.max_ds = sc->max_ds;
    d.partial_blocks = sc->partial_blocks;
    d.sd = sc->dev;
    d.keys_a = sc->keys_a; d.keys_b = sc->keys_b; d.vals_a = sc->vals_a; d.vals_b = sc->vals_b;
    d.hist = sc->hist; d.blockcnt = sc->blockcnt; d.hpos = sc->hpos; d.longlist = sc->longlist; d.tie_list = sc->tie_list;
    d.sorted = sc->sorted; d.ds_body = sc->ds_body; d.ds_world = sc->ds_world; d.nn_pts = sc->nn_pts; d.normvec = sc->normvec;
    d.nn_cnt = sc->nn_cnt; d.selected = sc->selected; d.partial = sc->partial;
    d.host_nds = sc->host_nds_dev;
    d.ctrl = d_ctrl;
    d.result = d_res;
}

void fill_ctrl(EskfDev& c, const double* x26, const double* P529, double R, int max_iter, int degenerate_detect_en, int joint = 0) {
    c.joint = joint;
    memcpy(c.x, x26, sizeof(double) * 26);
    memcpy(c.P, P529, sizeof(double) * 529);
    for (int k = 0; k < kEkN; k++) c.limit[k] = 0.001;
    c.R = R;
    c.maximum_iter = max_iter;
    c.degenerate_detect_en = degenerate_detect_en;
    c.is_degenerate = 0;
    ek_begin(c);
}

void group_free(Group& g) {
    for (lio_engine* e : g.eng) lio_engine_destroy(e);  // (joint mode: the drivers first -- they hold pointers to the sub-maps' engines)
    for (lio_engine* e : g.sub_eng) lio_engine_destroy(e);
    if (g.d_local32) hipFree(g.d_local32);
    if (g.d_gathered) hipFree(g.d_gathered);
    if (g.d_ds_send) hipFree(g.d_ds_send);
    if (g.d_ds_all) hipFree(g.d_ds_all);
    if (g.d_rowmaps) hipFree(g.d_rowmaps);
    for (int k = 0; k < 5; k++)
        if (g.exec[k]) hipGraphExecDestroy(g.exec[k]);
    if (g.d_block) hipFree(g.d_block);
    if (g.d_rawbuf) hipFree(g.d_rawbuf);
    if (g.h_block) hipHostFree(g.h_block);
    if (g.d_seq) hipFree(g.d_seq);
    if (g.h_seq) hipHostFree(g.h_seq);
    if (g.h_res) hipHostFree(g.h_res);
    if (g.stream) hipStreamDestroy(g.stream);
    if (g.bt) {
        if (g.bt->created)
            for (int c = 0; c < BatchTimer::kClasses; c++)
                for (int i = 0; i < BatchTimer::kPool; i++) { hipEventDestroy(g.bt->ev[c][i][0]); hipEventDestroy(g.bt->ev[c][i][1]); }
        delete g.bt;
    }
}

// LIO_JOB_HOST_RAW: the job's cloud lives in HOST memory (pinned: lio_pinned_alloc, or any hipHostMalloc / hipHostRegister'ed range; pageable memory
// works through the runtime's staging, slower).  It is copied into the slot's row of the group's raw ring on the round's stream, ahead of the
// round's kernels: with several rounds in flight the copy of one round runs beside the kernels of the others (the copy engine and the CUs are
// separate; py_utils.cpp:149-181 + slam_wrapper.cpp:64-84 are this copy in the reference).  Returns the device address the chain reads.
const float4* stage_host_raw(lio_batch* b, Group& g, int slot, const lio_scan_job& job, int* rc) {
    *rc = LIO_OK;
    if (!g.d_rawbuf) {
        if (hipMalloc(reinterpret_cast<void**>(&g.d_rawbuf), sizeof(float4) * (size_t)b->max_raw * (size_t)b->n_slots) != hipSuccess) {
            set_error("lio_batch: no memory for the raw ring of LIO_JOB_HOST_RAW jobs (%d x %u points)", b->n_slots, b->max_raw);
            *rc = LIO_E_DEVICE;
            return nullptr;
        }
    }
    float4* dst = g.d_rawbuf + (size_t)slot * b->max_raw;
    if (hipMemcpyAsync(dst, job.d_raw, sizeof(float4) * (size_t)job.n_raw, hipMemcpyHostToDevice, g.stream) != hipSuccess) {
        set_error("lio_batch: copy of a LIO_JOB_HOST_RAW cloud failed: %s", hipGetErrorString(hipGetLastError()));
        *rc = LIO_E_DEVICE;
        return nullptr;
    }
    return dst;
}

// one round: jobs[first .. first + n) into the slots of g, everything enqueued on g.stream
int submit(lio_batch* b, Group& g, lio_scan_job* jobs, int first, int n, int passes) {
    const int B = b->n_slots;
    g.seq++;
    g.n_active = 0;
    g.max_n_raw = 0;
    g.launched_passes = passes;
    for (int s = 0; s < B; s++) {
        SlotDesc& d = g.h_desc[s];
        d.active = 0;
        g.job_of_slot[s] = -1;
        if (s >= n) continue;
        lio_scan_job& job = jobs[first + s];
        g.job_of_slot[s] = first + s;
        job.rc = LIO_E_INVALID;
        job.n_ds = job.n_pass = job.n_knn_pass = 0;
        // (unknown bits first: a garbage flags word must be rejected, not read as IDLE)
        if (job.flags & ~LIO_JOB_FLAGS_KNOWN) { set_error("lio_scan_job.flags = 0x%x: unknown bits (a job array that was not zero-initialised?)", job.flags); continue; }
        if (job.flags & LIO_JOB_IDLE) { job.rc = 0; g.job_of_slot[s] = -1; continue; }  // (a session without a scan, sequence mode's marker: nothing to do here either)
        if (!job.state_in || !job.cov_in || (!job.d_raw && job.n_raw)) continue;
        if (job.n_raw > b->max_raw) { set_error("scan of %u points exceeds max_raw %u", job.n_raw, b->max_raw); job.rc = LIO_E_CAPACITY; continue; }
        if (job.n_raw == 0) { job.rc = 2; continue; }  // "FastLio undistort points is empty"
        d.raw = static_cast<const float4*>(job.d_raw);
        if (job.flags & LIO_JOB_HOST_RAW) {
            int src = LIO_OK;
            d.raw = stage_host_raw(b, g, s, job, &src);
            if (src != LIO_OK) { job.rc = src; continue; }
        }
        d.n_raw = job.n_raw;
        d.nblocks = (job.n_raw + 2047u) / 2048u;
        d.active = 1;
        d.seq = g.seq;
        d.min_ds = 5;  // laserMapping.cpp:1246: fewer than five downsampled points are not registered
        d.reset_cache = (job.flags & LIO_JOB_KEEP_CACHE) ? 0u : 1u;
        fill_ctrl(g.h_ctrl[s], job.state_in, job.cov_in, 0.001 /* LASER_POINT_COV */, 4, 1, b->joint ? 1 : 0);
        g.h_res[s].seq = g.seq - 1;
        g.n_active++;
        if (job.n_raw > g.max_n_raw) g.max_n_raw = job.n_raw;
    }
    const int M = (int)b->maps.size();
    for (int m = 1; m < M; m++)  // the per-round words of the further sub-maps' rows follow the slot's own
        for (int s = 0; s < B; s++) {
            SlotDesc& d = g.h_desc[(size_t)m * B + s];
            const SlotDesc& p = g.h_desc[s];
            d.raw = p.raw; d.n_raw = p.n_raw; d.nblocks = p.nblocks; d.active = p.active; d.seq = p.seq; d.min_ds = p.min_ds; d.reset_cache = p.reset_cache;
        }
    if (g.n_active == 0) return LIO_OK;
    // The round is ONE submission: upload of the staging block, voxel-grid chain, (maximum_iter + 1) x {kNN, linearise, filter pass}.
    // Every argument is fixed per group (pointers into the group's blocks; what changes from round to round travels in the block), so
    // the sequence is captured once into a graph and replayed with a single hipGraphLaunch -- ~35 API calls of 5-15 us each otherwise,
    // which made the one submitting host thread the bottleneck.  Grids are sized for the batch's max_raw (surplus workgroups exit at once).
    const bool timed = g.bt && g.bt->on;
    const MapRef* rowmaps = b->joint ? joint_rowmaps(b, g) : nullptr;
    auto enqueue = [&](BatchTimer* bt) -> int {
        LIO_HIP_TRY(hipMemcpyAsync(g.d_block, g.h_block, g.block_bytes, hipMemcpyHostToDevice, g.stream));
        const uint32_t ds_bound = b->max_raw < b->max_ds ? b->max_raw : b->max_ds;
        int rc = LIO_OK;
        if (b->joint && b->world > 1 && b->split_ds) {
            // this rank downsamples its share of the round's scans; one all-gather brings everybody's clouds (p2plane.hip: pack / unpack)
            const int per = (B + b->world - 1) / b->world;
            const int s0 = std::min(B, b->rank * per), s1 = std::min(B, s0 + per);
            if (g.ds_world != b->world) {
                if (g.d_ds_send) hipFree(g.d_ds_send);
                if (g.d_ds_all) hipFree(g.d_ds_all);
                g.d_ds_send = g.d_ds_all = nullptr;
                g.ds_world = 0;
                const size_t chunk = (size_t)per * ds_exchange_slot_bytes(ds_bound);
                LIO_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g.d_ds_send), chunk));
                LIO_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g.d_ds_all), chunk * (size_t)b->world));
                g.ds_world = b->world;
            }
            if (bt) bt->begin(0);
            if (s1 > s0) rc = vg_downsample_batch(g.stream, g.d_desc + s0, s1 - s0, b->max_raw, b->max_ds, 0.5f, passes);
            if (bt) bt->end(0);
            if (rc != LIO_OK) return rc;
            if (b->ds_cap == 0) {  // before any result: a quarter of the worst case, at least 4096 points (LIO_JOINT_DS_CAP: the tests' way to a cut cloud)
                const char* k = getenv("LIO_JOINT_DS_CAP");
                const uint32_t want = k ? (uint32_t)strtoul(k, nullptr, 10) : ((ds_bound / 4u + 1023u) & ~1023u);
                b->ds_cap = std::min(ds_bound, std::max(k ? 1u : 4096u, want));
            }
            const uint32_t cap = b->ds_cap_full ? ds_bound : b->ds_cap;
            rc = p2plane_batch_exchange_ds(b->comm, b->world, b->gather_hook, b->gather_ctx, g.stream, g.d_desc, B, per, s0, s1, cap, g.d_ds_send, g.d_ds_all);
            if (rc == LIO_OK && s0 > 0) rc = scan_begin_rows(g.stream, g.d_desc, s0);
            if (rc == LIO_OK && s1 < B) rc = scan_begin_rows(g.stream, g.d_desc + s1, B - s1);
            if (rc != LIO_OK) return rc;
        } else {
            if (bt) bt->begin(0);
            rc = vg_downsample_batch(g.stream, g.d_desc, B, b->max_raw, b->max_ds, 0.5f, passes);
            if (bt) bt->end(0);
            if (rc != LIO_OK) return rc;
        }
        if (b->joint) {
            // the slot's downsampled cloud to the other local sub-maps' scan buffers, their neighbour caches resized / forgotten like the slot's own
            rc = p2plane_batch_share(g.stream, g.d_desc, B, M, ds_bound);
            if (rc == LIO_OK) rc = scan_begin_rows(g.stream, g.d_desc + B, B * (M - 1));
            if (rc != LIO_OK) return rc;
            return p2plane_batch_update_joint(b->maps.data(), M, b->comm, b->world, b->gather_hook, b->gather_ctx, g.stream, g.d_desc, B, ds_bound, 5, g.d_local32, g.d_gathered, bt,
                                              rowmaps);
        }
        return p2plane_batch_update(b->map, g.stream, g.d_desc, B, ds_bound, 5, bt, bt ? b->count_touched : 0);
    };
    if (!b->use_graph || timed || b->joint) return enqueue(timed ? g.bt : nullptr);
    if (g.graph_table != b->map->table || g.graph_stencil != b->map->stencil.n) {
        for (int k = 0; k < 5; k++)
            if (g.exec[k]) { hipGraphExecDestroy(g.exec[k]); g.exec[k] = nullptr; }
        g.graph_table = b->map->table;
        g.graph_stencil = b->map->stencil.n;
    }
    if (!g.exec[passes]) {
        hipGraph_t graph = nullptr;
        LIO_HIP_TRY(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue(nullptr);
        const hipError_t e2 = hipStreamEndCapture(g.stream, &graph);
        if (rc != LIO_OK) { if (graph) hipGraphDestroy(graph); return rc; }
        LIO_HIP_TRY(e2);
        LIO_HIP_TRY(hipGraphInstantiate(&g.exec[passes], graph, nullptr, nullptr, 0));
        hipGraphDestroy(graph);
    }
    LIO_HIP_TRY(hipGraphLaunch(g.exec[passes], g.stream));
    return LIO_OK;
}

int wait_group(Group& g, int B) {
    for (int s = 0; s < B; s++) {
        if (!g.h_desc[s].active) continue;
        volatile uint32_t* seq = &g.h_res[s].seq;
        for (uint64_t spin = 0; *seq != g.seq; spin++) {
            __builtin_ia32_pause();
            if (spin > 2000 && (spin & 31) == 0) sched_yield();
            if (spin > 40000000ull) {
                LIO_HIP_TRY(hipStreamSynchronize(g.stream));
                if (*seq != g.seq) { set_error("batch slot %d did not report (seq %u, expected %u)", s, *seq, g.seq); return LIO_E_DEVICE; }
                break;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return LIO_OK;
}

}  // namespace

// HIP deals streams to GPU_MAX_HW_QUEUES hardware queues (4 by default) round robin; with four, the rounds in flight of a batch share queues
// with the idle per-slot streams and mostly run back to back (measured 0.085 -> 0.070 ms per scan with 8).  The runtime reads the variable
// when it initialises (the first HIP call of the process): the APPLICATION exports GPU_MAX_HW_QUEUES=8 before that (bench.py, the tests'
// conftest and the pybind module's loader do) -- the library does not touch the process environment (until round 3 a load-time constructor
// called setenv: a process-wide side effect from dlopen, not thread-safe); lio_batch_create leaves a note in lio_last_warning when the value
// in force is lower.

extern "C" {

static lio_batch* batch_create_impl(lio_map** maps, int n_maps, lio_comm* comm, int n_slots, int n_groups, uint32_t max_raw, uint32_t max_ds) {
    lio_map* map = n_maps > 0 && maps ? maps[0] : nullptr;
    if (!map || n_maps > 64 || n_slots < 1 || n_slots > 256 || n_groups < 1 || n_groups > 8 || max_raw == 0 || max_ds == 0) { set_error("lio_batch_create: bad argument"); return nullptr; }
    for (int m = 1; m < n_maps; m++)
        if (!maps[m] || maps[m]->device != map->device) { set_error("lio_batch_create_joint: the sub-maps of a batch live on one device"); return nullptr; }
    if (hipSetDevice(map->device) != hipSuccess) { set_error("lio_batch_create: no HIP device %d", map->device); return nullptr; }
    lio_batch* b = new lio_batch();
    b->map = map;
    b->device = map->device;
    b->n_slots = n_slots;
    b->max_raw = max_raw;
    b->max_ds = max_ds;
    b->maps.assign(maps, maps + n_maps);
    b->comm = comm;
    b->world = comm ? lio_comm_world(comm) : 1;
    b->rank = comm ? lio_comm_rank(comm) : 0;
    { const char* k = getenv("LIO_JOINT_SPLIT_DS"); b->split_ds = (k && k[0] == '0') ? 0 : 1; }
    b->joint = n_maps > 1 || comm != nullptr;
    b->groups.resize(n_groups);
    { const char* k = getenv("LIO_BATCH_GRAPH"); b->use_graph = (k && k[0] == '0') ? 0 : 1; }
    bool ok = true;
    const int M = n_maps;
    // the groups' streams first: HIP deals streams to its few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) round robin in creation
    // order -- created after the ~B scan streams of the slots, two groups can land on one queue and their rounds run back to back
    for (Group& g : b->groups) ok = ok && hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking) == hipSuccess;
    for (Group& g : b->groups) {
        g.job_of_slot.assign(n_slots, -1);
        const size_t desc_bytes = sizeof(SlotDesc) * (size_t)n_slots * (size_t)M;
        g.block_bytes = desc_bytes + sizeof(EskfDev) * (size_t)n_slots;
        ok = ok && hipMalloc(reinterpret_cast<void**>(&g.d_block), g.block_bytes) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&g.h_block), g.block_bytes, hipHostMallocDefault) == hipSuccess;
        if (ok) {
            g.d_desc = reinterpret_cast<SlotDesc*>(g.d_block);
            g.h_desc = reinterpret_cast<SlotDesc*>(g.h_block);
            g.d_ctrl = reinterpret_cast<EskfDev*>(g.d_block + desc_bytes);
            g.h_ctrl = reinterpret_cast<EskfDev*>(g.h_block + desc_bytes);
        }
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&g.h_res), sizeof(lio_batch_result) * n_slots, hipHostMallocMapped) == hipSuccess;
        ok = ok && hipHostGetDevicePointer(reinterpret_cast<void**>(&g.h_res_dev), g.h_res, 0) == hipSuccess;
        if (b->joint) {
            ok = ok && hipMalloc(reinterpret_cast<void**>(&g.d_local32), sizeof(double) * 32 * n_slots) == hipSuccess;
            ok = ok && hipMalloc(reinterpret_cast<void**>(&g.d_gathered), sizeof(double) * 32 * n_slots * (size_t)b->world) == hipSuccess;
        }
        if (!ok) break;
        memset(g.h_block, 0, g.block_bytes);
        memset(g.h_res, 0, sizeof(lio_batch_result) * n_slots);
        ok = hipMemset(g.d_block, 0, g.block_bytes) == hipSuccess;
        for (int s = 0; s < n_slots && ok; s++) {
            lio_engine* e = lio_engine_create_shared(map, max_raw, max_ds);
            if (!e) { ok = false; break; }
            lio_engine_set_flags(e, 1, 0, 0.0, -10.0);
            g.eng.push_back(e);
            fill_desc(g.h_desc[s], lio_engine_scan(e), &g.d_ctrl[s], &g.h_res_dev[s]);
        }
        for (int m = 1; m < M && ok; m++)
            for (int s = 0; s < n_slots && ok; s++) {
                lio_engine* e = lio_engine_create_shared(maps[m], max_raw, max_ds);
                if (!e) { ok = false; break; }
                lio_engine_set_flags(e, 1, 0, 0.0, -10.0);
                g.sub_eng.push_back(e);
                fill_desc(g.h_desc[(size_t)m * n_slots + s], lio_engine_scan(e), &g.d_ctrl[s], nullptr);
            }
        // the host-driven joint path behind every slot (lio_engine_set_joint): takes over the rare scan whose pass needs the degeneracy sums
        for (int s = 0; s < n_slots && ok && b->joint; s++) {
            std::vector<lio_engine*> others;
            for (int m = 1; m < M; m++) others.push_back(g.sub_eng[(size_t)(m - 1) * n_slots + s]);
            ok = lio_engine_set_joint(g.eng[s], others.empty() ? nullptr : others.data(), (int)others.size(), comm) == LIO_OK;
        }
    }
    if (!ok) {
        set_error("lio_batch_create: device setup failed: %s", hipGetErrorString(hipGetLastError()));
        lio_batch_destroy(b);
        return nullptr;
    }
    {   // not an error: a note for whoever wonders why rounds do not overlap
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        if (n_groups > 1 && (!q || atoi(q) < 8))
            set_warning("lio_batch_create: GPU_MAX_HW_QUEUES=%s -- with fewer than 8 hardware queues the %d rounds in flight mostly serialise (~20 %% slower); "
                        "export GPU_MAX_HW_QUEUES=8 before the process's first HIP call", q ? q : "(unset: 4)", n_groups);
    }
    return b;
}

lio_batch* lio_batch_create(lio_map* map, int n_slots, int n_groups, uint32_t max_raw, uint32_t max_ds) {
    return batch_create_impl(&map, map ? 1 : 0, nullptr, n_slots, n_groups, max_raw, max_ds);
}

lio_batch* lio_batch_create_joint(lio_map** sub_maps, int n_sub_maps, lio_comm* comm, int n_slots, int n_groups, uint32_t max_raw, uint32_t max_ds) {
    return batch_create_impl(sub_maps, n_sub_maps, comm, n_slots, n_groups, max_raw, max_ds);
}

int lio_batch_set_gather_hook(lio_batch* b, lio_gather_fn fn, void* ctx, int rank, int world) {
    if (!b || !fn || world < 1 || rank < 0 || rank >= world) return LIO_E_INVALID;
    if (!b->joint || b->comm) { set_error("lio_batch_set_gather_hook: for a batch made by lio_batch_create_joint without a communicator"); return LIO_E_STATE; }
    hipSetDevice(b->map->device);
    for (auto& g : b->groups) {  // room for every rank's records
        LIO_HIP_TRY(hipStreamSynchronize(g.stream));
        if (g.d_gathered) hipFree(g.d_gathered);
        g.d_gathered = nullptr;
        LIO_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g.d_gathered), sizeof(double) * 32 * b->n_slots * (size_t)world));
    }
    b->gather_hook = fn;
    b->gather_ctx = ctx;
    b->world = world;
    b->rank = rank;
    return LIO_OK;
}

int lio_batch_exchange_stats(lio_batch* b, uint32_t* chunk_points, uint64_t* jobs_rerun, uint64_t* bytes_per_rank_and_round) {
    if (!b) return LIO_E_INVALID;
    const bool on = b->joint && b->world > 1 && b->split_ds;
    if (chunk_points) *chunk_points = on ? b->ds_cap : 0u;
    if (jobs_rerun) *jobs_rerun = b->n_ds_cut;
    if (bytes_per_rank_and_round) {
        const int per = (b->n_slots + b->world - 1) / b->world;
        *bytes_per_rank_and_round = on && b->ds_cap ? (uint64_t)per * ds_exchange_slot_bytes(b->ds_cap) : 0u;
    }
    return LIO_OK;
}

void lio_batch_destroy(lio_batch* b) {
    if (!b) return;
    hipSetDevice(b->device);
    if (getenv("LIO_BATCH_PROFILE") && b->n_rounds)
        fprintf(stderr, "lio_batch: %llu rounds of %d slots; host per round: submit %.1f us, wait %.1f us, collect %.1f us\n", (unsigned long long)b->n_rounds,
                b->n_slots, 1e6 * b->t_submit / b->n_rounds, 1e6 * b->t_wait / b->n_rounds, 1e6 * b->t_collect / b->n_rounds);
    for (Group& g : b->groups) {
        if (g.stream) hipStreamSynchronize(g.stream);
        group_free(g);
    }
    delete b;
}

int lio_batch_enable_kernel_timing(lio_batch* b, int on) {
    if (!b) return LIO_E_INVALID;
    hipSetDevice(b->device);
    for (Group& g : b->groups) {
        if (!g.bt) { g.bt = new BatchTimer(); g.bt->stream = g.stream; }
        if (on && !g.bt->created) {
            for (int c = 0; c < BatchTimer::kClasses; c++)
                for (int i = 0; i < BatchTimer::kPool; i++) {
                    LIO_HIP_TRY(hipEventCreate(&g.bt->ev[c][i][0]));
                    LIO_HIP_TRY(hipEventCreate(&g.bt->ev[c][i][1]));
                }
            g.bt->created = true;
        }
        g.bt->on = on != 0;
    }
    b->count_touched = (on & 2) ? 1 : 0;
    if (b->count_touched && b->map) { const int rc = map_enable_touch_bits(b->map); if (rc != LIO_OK) return rc; }
    return LIO_OK;
}

int lio_batch_kernel_times(lio_batch* b, lio_batch_times* out, int reset) {
    if (!b || !out) return LIO_E_INVALID;
    memset(out, 0, sizeof(*out));
    for (Group& g : b->groups) {
        if (!g.bt) continue;
        out->downsample_us += g.bt->us[0]; out->knn_us += g.bt->us[1]; out->linearize_us += g.bt->us[2]; out->step_us += g.bt->us[3];
        out->downsample_launches += g.bt->launches[0]; out->knn_launches += g.bt->launches[1]; out->linearize_launches += g.bt->launches[2];
        out->step_launches += g.bt->launches[3];
        out->insert_us += g.bt->us[4];
        out->insert_launches += g.bt->launches[4];
        if (reset)
            for (int c = 0; c < BatchTimer::kClasses; c++) { g.bt->us[c] = 0; g.bt->launches[c] = 0; }
    }
    return LIO_OK;
}

lio_engine* lio_batch_engine(lio_batch* b, int group, int slot) {
    if (!b || group < 0 || group >= (int)b->groups.size() || slot < 0 || slot >= b->n_slots) return nullptr;
    return b->groups[group].eng[slot];
}

int lio_batch_process(lio_batch* b, lio_scan_job* jobs, int n_jobs) {
    if (!b || (!jobs && n_jobs) || n_jobs < 0) return LIO_E_INVALID;
    if (b->sequences) { set_error("lio_batch_process: a sequence batch is driven by lio_batch_sequences_step"); return LIO_E_STATE; }
    hipSetDevice(b->device);
    const int B = b->n_slots;
    int next = 0, first_err = 0;
    std::deque<int> inflight;
    std::vector<int> retry;  // jobs whose sort was launched with too few radix passes (the bounding box grew across a power of 256)
    auto note = [&](int rc) { if (rc < 0 && !first_err) first_err = rc; };
    auto collect = [&](Group& g) {
        int need_max = 1;
        for (int s = 0; s < B; s++) {
            const int j = g.job_of_slot[s];
            if (j < 0) continue;
            if (!g.h_desc[s].active) { note(jobs[j].rc); continue; }  // rejected at submission (bad pointers, more points than max_raw): counted, not lost
            lio_scan_job& job = jobs[j];
            const lio_batch_result& r = g.h_res[s];
            if (r.radix_passes > need_max) need_max = r.radix_passes;
            if (r.radix_passes > g.launched_passes) { retry.push_back(j); continue; }
            lio_engine* e = g.eng[s];
            lio_scan* sc = lio_engine_scan(e);
            sc->have_ds = r.n_ds;
            sc->n_raw = g.h_desc[s].n_raw;
            job.n_ds = r.n_ds;
            job.n_pass = r.n_pass;
            job.n_knn_pass = r.n_knn_pass;
            if (r.err & 4) {  // the cloud did not fit its chunk of the round's all-gather (every rank sees the flag): again, with full-size chunks
                hipMemsetAsync(&sc->dev->err, 0, 4, g.stream);
                b->n_ds_cut++;
                retry.push_back(j);
                continue;
            }
            if (r.err & 1) {
                set_error("downsampled scan exceeds max_ds %u", b->max_ds);
                hipMemsetAsync(&sc->dev->err, 0, 4, g.stream);
                job.rc = LIO_E_CAPACITY;
                note(job.rc);
                continue;
            }
            if ((uint32_t)r.n_ds > b->ds_seen_max) b->ds_seen_max = (uint32_t)r.n_ds;
            if (r.status == EK_SKIPPED) { job.rc = 2; continue; }  // fewer than five downsampled points
            if (r.status == EK_NEEDS_HOST && b->joint && b->gather_hook) {
                set_error("lio_batch (gather hook mode): a scan needs the host-driven joint path, which only exists over a lio_comm");
                job.rc = LIO_E_STATE;
                note(job.rc);
                continue;
            }
            if (r.status == EK_NEEDS_HOST && b->joint) {
                // a pass of this scan needs the degeneracy sums of laserMapping.cpp:946-964, which live on several sub-maps / ranks: the scan is
                // registered again by the host-driven joint path of the slot's engines (every rank takes this branch for the same jobs in the
                // same order: the decision was made from the gathered sums)
                if (!(job.flags & LIO_JOB_KEEP_CACHE)) {
                    scan_forget_cache(sc);
                    for (int m = 1; m < (int)b->maps.size(); m++) scan_forget_cache(lio_engine_scan(g.sub_eng[(size_t)(m - 1) * B + s]));
                }
                double st26[26], cov[529];
                memcpy(st26, job.state_in, sizeof(st26));
                memcpy(cov, job.cov_in, sizeof(cov));
                // (a LIO_JOB_HOST_RAW job's d_raw is a HOST address: its staged copy in the group's raw ring is still intact for this slot)
                const void* d_cloud = (job.flags & LIO_JOB_HOST_RAW) ? static_cast<const void*>(g.d_rawbuf + (size_t)s * b->max_raw) : job.d_raw;
                const int rc = engine_joint_register_device(e, d_cloud, job.n_raw, job.lidar_beg_time, st26, cov);
                if (rc < 0) { job.rc = rc; note(rc); continue; }
                engine_count_passes(e, &job.n_pass, &job.n_knn_pass);
                if (job.state_out) memcpy(job.state_out, st26, sizeof(st26));
                job.rc = rc;
                continue;
            }
            if (r.status == EK_NEEDS_HOST) {
                // a pass with 1 <= N_eff < 23: the dense gain of esekfom.hpp:1715-1744 needs the rows -- the host filter takes over from
                // that pass on, through the per-pass path of the slot's engine (same scan buffers, same neighbour cache and gates)
                const int rc = engine_resume_update(e, r.state, job.state_in, job.cov_in, r.loop_i, r.loop_converge, r.loop_t);
                if (rc != LIO_OK) { job.rc = rc; note(rc); continue; }
                int np = 0, nk = 0;
                engine_count_passes(e, &np, &nk);
                job.n_pass += np;
                job.n_knn_pass += nk;
                if (job.state_out) lio_engine_get_state(e, job.state_out);
                job.rc = 3;
                continue;
            }
            if (job.state_out) memcpy(job.state_out, r.state, sizeof(double) * 26);
            job.rc = 3;
        }
        if (need_max > 4) need_max = 4;
        b->pred_passes = need_max;
        if (b->joint && b->world > 1 && b->split_ds && b->ds_seen_max && !getenv("LIO_JOINT_DS_CAP")) {
            const uint32_t ds_bound = b->max_raw < b->max_ds ? b->max_raw : b->max_ds;
            const uint32_t want = (b->ds_seen_max + b->ds_seen_max / 4u + 1023u) & ~1023u;
            b->ds_cap = std::min(ds_bound, std::max(4096u, want));
        }
    };
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point c) { return std::chrono::duration<double>(c - a).count(); };
    // a failed submission / wait: the rounds of the other groups are still in flight and own their pinned blocks and result records --
    // wait for them before handing the error back (their jobs keep LIO_E_INVALID), so that the next call starts from idle groups
    auto bail = [&](int rc) {
        // every group's stream, not only the rounds in flight: the group whose submission failed may have queued copies out of the caller's
        // LIO_JOB_HOST_RAW buffers, which only have to stay valid until this call returns
        for (auto& gg : b->groups) hipStreamSynchronize(gg.stream);
        inflight.clear();
        return rc;
    };
    for (size_t gi = 0; gi < b->groups.size() && next < n_jobs; gi++) {
        const int n = n_jobs - next < B ? n_jobs - next : B;
        const auto t0 = clk::now();
        const int rc = submit(b, b->groups[gi], jobs, next, n, b->pred_passes);
        b->t_submit += secs(t0, clk::now());
        b->n_rounds++;
        if (rc != LIO_OK) return bail(rc);
        next += n;
        inflight.push_back((int)gi);
    }
    while (!inflight.empty()) {
        const int gi = inflight.front();
        inflight.pop_front();
        Group& g = b->groups[gi];
        const auto t0 = clk::now();
        if (g.n_active) {
            const int rc = wait_group(g, B);
            if (rc != LIO_OK) return bail(rc);
        }
        if (g.bt && g.bt->on) { hipStreamSynchronize(g.stream); g.bt->resolve(); }
        const auto t1 = clk::now();
        collect(g);
        const auto t2 = clk::now();
        b->t_wait += secs(t0, t1);
        b->t_collect += secs(t1, t2);
        if (next < n_jobs) {
            const int n = n_jobs - next < B ? n_jobs - next : B;
            const int rc = submit(b, g, jobs, next, n, b->pred_passes);
            b->t_submit += secs(t2, clk::now());
            b->n_rounds++;
            if (rc != LIO_OK) return bail(rc);
            next += n;
            inflight.push_back(gi);
        }
    }
    // the rare re-runs, one by one with all four radix passes
    const std::vector<int> todo = retry;
    retry.clear();
    b->ds_cap_full = true;
    for (const int j : todo) {
        Group& g = b->groups[0];
        int rc = submit(b, g, jobs, j, 1, 4);
        if (rc == LIO_OK && g.n_active) rc = wait_group(g, B);
        if (rc != LIO_OK) { b->ds_cap_full = false; return bail(rc); }
        collect(g);
    }
    b->ds_cap_full = false;
    for (const int j : retry) { jobs[j].rc = LIO_E_DEVICE; note(LIO_E_DEVICE); }
    return first_err;
}

// ---------------------------------------------------------------------------------------------------------------------
// Sequence mode: B x G independent SLAM sessions, one per slot, each with ITS OWN map (the slot's engine owns it).  A step registers the next
// scan of every session and inserts it into the session's map -- fastlio_main from the downsample to map_incremental (laserMapping.cpp:
// 1193-1304) -- as one blind submission per group: the voxel-grid chain, (maximum_iter + 1) x {stencil kNN against the slot's own map,
// linearisation, filter pass}, then classify + AddPoints (+ LRU) for every slot whose update finished, all with blockIdx.y = slot.  What the
// static batch leaves out (SURVEY 8d's B_ins) is inside the round here.  Scans the round cannot take -- a session's first scans (time origin,
// map seed), a scan whose update needs the host filter (1 <= N_eff < 23), a bounding box that needs more radix passes than were launched --
// go through the slot's engine, i.e. through lio_engine_process_scan_device's own code: a session driven here and one driven scan by scan
// through an engine with the device loop on (lio_engine_set_device_loop) run the same kernels on the same data.
static void fill_mapref(MapRef& r, const lio_map* m) {
    r.table = m->table; r.cap = m->cap; r.pending = m->pending; r.created = m->created; r.pool = m->pool; r.pool_seq = m->pool_seq; r.tie_mode = m->tie_mode; r.md = m->dev;
    r.slot_of_point = m->slot_of_point; r.stage = m->stage; r.pool_cap = m->pool_cap;
    r.touch = m->touch; r.prev_touch = m->prev_touch; r.lru_log = m->lru_log; r.log_mask = m->lru_log_cap ? m->lru_log_cap - 1 : 0;
    r.free_items = m->free_items; r.free_in = m->free_in; r.free_cap = m->free_cap;
    r.first_touch = m->first_touch; r.lru_g = m->lru_g; r.lru_rec = m->lru_rec;
    r.lru_capacity = (uint32_t)m->lru_capacity; r.lru_max_distance = m->lru_max_distance;
    r.mask = m->table_mask; r.inv_res = m->inv_res; r.res = m->res; r.key_mode = m->key_mode; r.max_voxels = (uint32_t)m->max_voxels;
    r.stencil_id = m->stencil_id;
}

lio_batch* lio_batch_create_sequences(int device, float resolution, int stencil, uint64_t max_points, uint64_t max_voxels, int n_slots, int n_groups,
                                      uint32_t max_raw, uint32_t max_ds) {
    if (n_slots < 1 || n_slots > 256 || n_groups < 1 || n_groups > 8 || max_raw == 0 || max_ds == 0 || max_points == 0 || max_voxels == 0) {
        set_error("lio_batch_create_sequences: bad argument");
        return nullptr;
    }
    const uint32_t ds_bound = max_raw < max_ds ? max_raw : max_ds;
    if (ds_bound > max_points || ds_bound > (1u << 20)) { set_error("lio_batch_create_sequences: a scan's %u downsampled points exceed the maps' insert scratch", ds_bound); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error("lio_batch_create_sequences: no HIP device %d", device); return nullptr; }
    lio_batch* b = new lio_batch();
    b->device = device;
    b->n_slots = n_slots;
    b->max_raw = max_raw;
    b->max_ds = max_ds;
    b->sequences = true;
    { const char* k = getenv("LIO_BATCH_GRAPH"); b->use_graph = (k && k[0] == '0') ? 0 : 1; }
    b->groups.resize(n_groups);
    bool ok = true;
    for (Group& g : b->groups) ok = ok && hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking) == hipSuccess;
    for (Group& g : b->groups) {
        g.job_of_slot.assign(n_slots, -1);
        g.seq_path.assign(n_slots, 0);
        const size_t desc_bytes = sizeof(SlotDesc) * (size_t)n_slots, ctrl_bytes = sizeof(EskfDev) * (size_t)n_slots;
        g.block_bytes = desc_bytes + ctrl_bytes + sizeof(MapRef) * (size_t)n_slots;
        ok = ok && hipMalloc(reinterpret_cast<void**>(&g.d_block), g.block_bytes) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&g.h_block), g.block_bytes, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipMalloc(reinterpret_cast<void**>(&g.d_seq), sizeof(SeqDev) * (size_t)n_slots) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&g.h_seq), sizeof(SeqDev) * (size_t)n_slots, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&g.h_res), sizeof(lio_batch_result) * n_slots, hipHostMallocMapped) == hipSuccess;
        ok = ok && hipHostGetDevicePointer(reinterpret_cast<void**>(&g.h_res_dev), g.h_res, 0) == hipSuccess;
        if (!ok) break;
        g.d_desc = reinterpret_cast<SlotDesc*>(g.d_block);
        g.h_desc = reinterpret_cast<SlotDesc*>(g.h_block);
        g.d_ctrl = reinterpret_cast<EskfDev*>(g.d_block + desc_bytes);
        g.h_ctrl = reinterpret_cast<EskfDev*>(g.h_block + desc_bytes);
        g.d_maps = reinterpret_cast<MapRef*>(g.d_block + desc_bytes + ctrl_bytes);
        g.h_maps = reinterpret_cast<MapRef*>(g.h_block + desc_bytes + ctrl_bytes);
        memset(g.h_block, 0, g.block_bytes);
        memset(g.h_res, 0, sizeof(lio_batch_result) * n_slots);
        memset(g.h_seq, 0, sizeof(SeqDev) * (size_t)n_slots);
        ok = hipMemset(g.d_block, 0, g.block_bytes) == hipSuccess && hipMemset(g.d_seq, 0, sizeof(SeqDev) * (size_t)n_slots) == hipSuccess;
        for (int s = 0; s < n_slots && ok; s++) {
            lio_engine* e = lio_engine_create(device, resolution, stencil, max_points, max_voxels, max_raw, max_ds);
            if (!e) { ok = false; break; }
            lio_engine_set_device_loop(e, 1);  // the scans the round cannot take run the same filter kernels
            g.eng.push_back(e);
            fill_desc(g.h_desc[s], lio_engine_scan(e), &g.d_ctrl[s], &g.h_res_dev[s]);
        }
    }
    if (!ok) {
        set_error("lio_batch_create_sequences: device setup failed: %s", hipGetErrorString(hipGetLastError()));
        lio_batch_destroy(b);
        return nullptr;
    }
    b->map = lio_engine_map(b->groups[0].eng[0]);  // (what the accessors of the static batch expect to be non-null)
    return b;
}

int lio_batch_sequences_step(lio_batch* b, lio_scan_job* jobs, int n_jobs, double* cov_out) {
    if (!b || !jobs) return LIO_E_INVALID;
    if (!b->sequences) { set_error("lio_batch_sequences_step: for a batch made by lio_batch_create_sequences"); return LIO_E_STATE; }
    if (b->broken) { set_error("lio_batch_sequences_step: an earlier step failed after part of it had run on the device; the sessions' maps hold that sweep, their engines do not -- destroy the batch"); return LIO_E_STATE; }
    const int B = b->n_slots, G = (int)b->groups.size();
    if (n_jobs != B * G) { set_error("lio_batch_sequences_step: %d jobs for %d sessions (job j is the next scan of session j; LIO_JOB_IDLE marks a session without one)", n_jobs, B * G); return LIO_E_INVALID; }
    hipSetDevice(b->device);
    int first_err = 0;
    auto note = [&](int rc) { if (rc < 0 && !first_err) first_err = rc; };
    const uint32_t ds_bound = b->max_raw < b->max_ds ? b->max_raw : b->max_ds;
    const int passes = b->pred_passes;
    // ---- submit: one round per group ----
    for (int gi = 0; gi < G; gi++) {
        Group& g = b->groups[gi];
        g.seq++;
        g.n_active = 0;
        g.max_n_raw = 0;
        g.launched_passes = passes;
        g.any_lru = 0;
        StencilArgs stencils[4];
        int stencil_ids[4], n_st = 0;
        for (int s = 0; s < B; s++) {
            SlotDesc& d = g.h_desc[s];
            d.active = 0;
            g.job_of_slot[s] = gi * B + s;
            g.seq_path[s] = -1000;
            lio_scan_job& job = jobs[gi * B + s];
            job.n_ds = job.n_pass = job.n_knn_pass = 0;
            job.rc = LIO_E_INVALID;
            if (job.flags & ~LIO_JOB_FLAGS_KNOWN) { set_error("lio_scan_job.flags = 0x%x: unknown bits (a job array that was not zero-initialised?)", job.flags); note(job.rc); continue; }
            if (job.flags & LIO_JOB_IDLE) { job.rc = 0; g.job_of_slot[s] = -1; continue; }
            if (!job.state_in || !job.cov_in || (!job.d_raw && job.n_raw)) { note(job.rc); continue; }
            if (job.flags & LIO_JOB_HOST_RAW) { set_error("lio_batch_sequences_step: LIO_JOB_HOST_RAW is a static-batch flag (a session's sweeps arrive through lio_fastlio_pcl_stage / _commit)"); note(job.rc); continue; }
            if (job.n_raw > b->max_raw) { set_error("scan of %u points exceeds max_raw %u", job.n_raw, b->max_raw); job.rc = LIO_E_CAPACITY; note(job.rc); continue; }
            lio_engine* e = g.eng[s];
            int ekf_inited = 0;
            const int path = engine_seq_prepare(e, job.n_raw, job.lidar_beg_time, &ekf_inited);
            g.seq_path[s] = path;
            if (path < 0) { job.rc = path; note(path); continue; }
            if (path == 0 || path == 2) { job.rc = path; if (job.state_out) memcpy(job.state_out, job.state_in, sizeof(double) * 26); if (cov_out) memcpy(cov_out + (size_t)(gi * B + s) * 529, job.cov_in, sizeof(double) * 529); continue; }
            if (path == 10) {  // the engine's own path: downsample + map seed (nothing is registered yet)
                lio_engine_set_state(e, job.state_in);
                lio_engine_set_cov(e, job.cov_in);
                const int rc = lio_engine_process_scan_device(e, job.d_raw, job.n_raw, job.lidar_beg_time);
                job.rc = rc;
                note(rc);
                if (rc >= 0) {
                    lio_timings tm;
                    lio_engine_timings(e, &tm);
                    job.n_ds = tm.n_ds; job.n_pass = tm.n_pass; job.n_knn_pass = tm.n_knn_pass;
                    if (job.state_out) lio_engine_get_state(e, job.state_out);
                    if (cov_out) lio_engine_get_cov(e, cov_out + (size_t)(gi * B + s) * 529);
                }
                continue;
            }
            // path 11: a slot of the round
            lio_map* m = lio_engine_map(e);
            double laser_cov = 0.001, travel = 0, last_pos_lid[3];
            int maximum_iter = 4, deg_en = 1, static_map = 0;
            float leaf_surf = 0.5f, leaf_map = 0.5f;
            engine_seq_params(e, &laser_cov, &maximum_iter, &deg_en, &leaf_surf, &leaf_map, &static_map, &travel, last_pos_lid);
            if (leaf_surf != 0.5f || maximum_iter != 4) { set_error("sequence batch: the round is built for filter_size_surf 0.5 and NUM_MAX_ITERATIONS 4"); job.rc = LIO_E_STATE; note(job.rc); g.seq_path[s] = LIO_E_STATE; continue; }
            if (m->lru_capacity && m->tomb_bound > m->table_cap / 4) {  // (map_insert_dev's rule) tombstones of evicted voxels: rebuild the table first
                const int rc = map_rebuild(m, g.stream);
                if (rc != LIO_OK) { job.rc = rc; note(rc); g.seq_path[s] = rc; continue; }
            }
            d.raw = static_cast<const float4*>(job.d_raw);
            d.n_raw = job.n_raw;
            d.nblocks = (job.n_raw + 2047u) / 2048u;
            d.active = 1;
            d.seq = g.seq;
            d.min_ds = 5;
            d.reset_cache = 0;  // a session: Nearest_Points survives from scan to scan
            fill_ctrl(g.h_ctrl[s], job.state_in, job.cov_in, laser_cov, maximum_iter, deg_en, 0);
            MapRef& r = g.h_maps[s];
            fill_mapref(r, m);
            r.stamp_base = (unsigned long long)(m->n_batches + 1) << kStampIdxBits;
            r.do_insert = static_map ? 0u : 1u;
            r.ekf_inited = (uint32_t)ekf_inited;
            r.map_leaf = leaf_map;
            r.travel_prev = travel;
            for (int i = 0; i < 3; i++) r.last_pos_lid[i] = last_pos_lid[i];
            if (m->lru_capacity) g.any_lru = 1;
            int k = 0;
            while (k < n_st && stencil_ids[k] != m->stencil_id) k++;
            if (k == n_st && n_st < 4) { stencils[n_st] = m->stencil; stencil_ids[n_st] = m->stencil_id; n_st++; }
            else if (k == n_st) { set_error("sequence batch: more than four different stencils in one round"); job.rc = LIO_E_STATE; note(job.rc); d.active = 0; g.seq_path[s] = LIO_E_STATE; continue; }
            g.h_res[s].seq = g.seq - 1;
            g.n_active++;
            if (job.n_raw > g.max_n_raw) g.max_n_raw = job.n_raw;
        }
        if (g.n_active == 0) continue;
        const bool timed = g.bt && g.bt->on;
        // the round: upload, chain, update, insert, read-back -- every argument is fixed per group (pointers into the group's blocks; what changes
        // travels in the block) except the stencils, which the neighbour search takes by value: captured once per (radix passes, stencil set, LRU
        // or not) into a graph and replayed with one hipGraphLaunch (~45 API calls otherwise)
        auto enqueue = [&](BatchTimer* bt) -> int {
            LIO_HIP_TRY(hipMemcpyAsync(g.d_block, g.h_block, g.block_bytes, hipMemcpyHostToDevice, g.stream));
            if (bt) bt->begin(0);
            int rc = vg_downsample_batch(g.stream, g.d_desc, B, b->max_raw, b->max_ds, 0.5f, passes);
            if (bt) bt->end(0);
            if (rc == LIO_OK) rc = p2plane_seq_update(g.stream, g.d_maps, g.d_desc, B, ds_bound, 5, stencils, stencil_ids, n_st, bt);
            if (bt) bt->begin(4);
            if (rc == LIO_OK) rc = p2plane_seq_insert(g.stream, g.d_maps, g.d_desc, g.d_seq, B, ds_bound, g.any_lru);
            if (bt) bt->end(4);
            if (rc != LIO_OK) return rc;
            LIO_HIP_TRY(hipMemcpyAsync(g.h_seq, g.d_seq, sizeof(SeqDev) * (size_t)B, hipMemcpyDeviceToHost, g.stream));
            return LIO_OK;
        };
        int rc = LIO_OK;
        if (!b->use_graph || timed) {
            rc = enqueue(timed ? g.bt : nullptr);
        } else {
            int sig = g.any_lru ? 1 : 0;  // what the captured launches hold by value
            for (int k = 0; k < n_st; k++) sig = sig * 131 + stencil_ids[k] + 1;
            if (g.graph_stencil != sig) {
                for (int k = 0; k < 5; k++)
                    if (g.exec[k]) { hipGraphExecDestroy(g.exec[k]); g.exec[k] = nullptr; }
                g.graph_stencil = sig;
            }
            if (!g.exec[passes]) {
                hipGraph_t graph = nullptr;
                rc = hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? LIO_OK : LIO_E_DEVICE;
                if (rc == LIO_OK) {
                    rc = enqueue(nullptr);
                    const hipError_t e2 = hipStreamEndCapture(g.stream, &graph);
                    if (rc == LIO_OK && e2 != hipSuccess) rc = LIO_E_DEVICE;
                    if (rc == LIO_OK && hipGraphInstantiate(&g.exec[passes], graph, nullptr, nullptr, 0) != hipSuccess) rc = LIO_E_DEVICE;
                    if (graph) hipGraphDestroy(graph);
                }
                if (rc == LIO_E_DEVICE) set_error("sequence batch: capturing the round failed: %s", hipGetErrorString(hipGetLastError()));
            }
            if (rc == LIO_OK && hipGraphLaunch(g.exec[passes], g.stream) != hipSuccess) { set_error("sequence batch: hipGraphLaunch: %s", hipGetErrorString(hipGetLastError())); rc = LIO_E_DEVICE; }
        }
        if (rc != LIO_OK) {
            // nothing of this call is collected: every job that was waiting for a round (this group's and the earlier groups' slots) and every job of
            // the groups not visited yet reports the failure -- a caller (lio_batch_fastlio_main) must not read a stale or value-initialised rc as success
            for (int k = 0; k <= gi; k++) (void)hipStreamSynchronize(b->groups[k].stream);
            for (int k = 0; k < G; k++)
                for (int s2 = 0; s2 < B; s2++) {
                    lio_scan_job& j2 = jobs[k * B + s2];
                    const bool in_round = k <= gi && b->groups[k].seq_path[s2] == 11;
                    if (k > gi || in_round) j2.rc = (j2.flags & LIO_JOB_IDLE) && !(j2.flags & ~LIO_JOB_FLAGS_KNOWN) ? 0 : LIO_E_DEVICE;
                }
            // the rounds of the groups before this one have already run on the device -- their sessions' maps and filters hold this sweep, the engines'
            // host side (neighbour cache bookkeeping, travel, the deferred insert's accounting) does not: the batch cannot be stepped again (a retried
            // sweep would enter those maps twice)
            if (gi > 0) b->broken = true;
            return rc;
        }
        b->n_rounds++;
    }
    // ---- collect ----
    int need_max = 1;
    for (int gi = 0; gi < G; gi++) {
        Group& g = b->groups[gi];
        if (g.n_active == 0) continue;
        LIO_HIP_TRY(hipStreamSynchronize(g.stream));
        if (g.bt && g.bt->on) g.bt->resolve();
        for (int s = 0; s < B; s++) {
            if (!g.h_desc[s].active) continue;
            lio_scan_job& job = jobs[gi * B + s];
            const lio_batch_result& r = g.h_res[s];
            const SeqDev& q = g.h_seq[s];
            lio_engine* e = g.eng[s];
            lio_scan* sc = lio_engine_scan(e);
            double* cov_dst = cov_out ? cov_out + (size_t)(gi * B + s) * 529 : nullptr;
            if (r.seq != g.seq) { set_error("sequence batch: slot %d did not report", s); job.rc = LIO_E_DEVICE; note(job.rc); continue; }
            if (r.radix_passes > need_max) need_max = r.radix_passes;
            if (r.radix_passes > g.launched_passes) {
                // the bounding box needs more radix passes than the round launched: nothing was registered or inserted (the chain reports an empty
                // cloud); the scan goes through the engine's own path, which launches what it needs
                lio_engine_set_state(e, job.state_in);
                lio_engine_set_cov(e, job.cov_in);
                const int rc = lio_engine_process_scan_device(e, job.d_raw, job.n_raw, job.lidar_beg_time);
                job.rc = rc;
                note(rc);
                if (rc >= 0) {
                    lio_timings tm;
                    lio_engine_timings(e, &tm);
                    job.n_ds = tm.n_ds; job.n_pass = tm.n_pass; job.n_knn_pass = tm.n_knn_pass;
                    if (job.state_out) lio_engine_get_state(e, job.state_out);
                    if (cov_dst) lio_engine_get_cov(e, cov_dst);
                }
                continue;
            }
            sc->have_ds = r.n_ds;
            sc->n_raw = g.h_desc[s].n_raw;
            job.n_ds = r.n_ds;
            job.n_pass = r.n_pass;
            job.n_knn_pass = r.n_knn_pass;
            if (r.err & 1) {
                set_error("downsampled scan exceeds max_ds %u", b->max_ds);
                hipMemsetAsync(&sc->dev->err, 0, 4, g.stream);
                job.rc = LIO_E_CAPACITY;
                note(job.rc);
                continue;
            }
            if (r.status == EK_SKIPPED) {  // fewer than five downsampled points: nothing is registered (laserMapping.cpp:1246)
                job.rc = 2;
                if (job.state_out) memcpy(job.state_out, job.state_in, sizeof(double) * 26);
                if (cov_dst) memcpy(cov_dst, job.cov_in, sizeof(double) * 529);
                continue;
            }
            if (r.status == EK_NEEDS_HOST) {
                const int rc = engine_seq_resume(e, r.state, job.state_in, job.cov_in, r.loop_i, r.loop_converge, r.loop_t);
                if (rc < 0) { job.rc = rc; note(rc); continue; }
                int np = 0, nk = 0;
                engine_count_passes(e, &np, &nk);
                job.n_pass += np;
                job.n_knn_pass += nk;
                if (job.state_out) lio_engine_get_state(e, job.state_out);
                if (cov_dst) lio_engine_get_cov(e, cov_dst);
                job.rc = rc;
                continue;
            }
            const int rc = engine_seq_finish(e, r.state, q.P, r.n_ds, r.n_pass, r.n_knn_pass, r.n_eff, r.degenerate, (int)q.go, q.n_add, q.go ? q.map_err : 0u, (uint32_t)r.n_ds /* map_insert_dev's n on the engine's own path */);
            if (job.state_out) memcpy(job.state_out, r.state, sizeof(double) * 26);
            if (cov_dst) memcpy(cov_dst, q.P, sizeof(double) * 529);
            job.rc = rc;
            note(rc);
        }
    }
    if (need_max > 4) need_max = 4;
    b->pred_passes = need_max;
    return first_err;
}

}  // extern "C"

// fastlio_main for every session of a sequence batch at once: the sessions' engines carry the reference's front half (lio_fastlio_init on
// lio_batch_engine(b, g, s), then lio_fastlio_imu_enqueue / lio_fastlio_pcl_enqueue* as for a single engine).  Per session: sync_packages, IMU
// initialisation, forward propagation + undistortion exactly as lio_fastlio_main does them (engine.hip: fastlio_front); the scans that reach
// registration form ONE sequence round (lio_batch_sequences_step); then the back half.  rc_out[j]: what lio_fastlio_main would have returned
// for session j (LIO_MAIN_IDLE for a session without a complete package).
int lio_batch_fastlio_main(lio_batch* b, int* rc_out) {
    if (!b || !rc_out) return LIO_E_INVALID;
    if (!b->sequences) { set_error("lio_batch_fastlio_main: for a batch made by lio_batch_create_sequences"); return LIO_E_STATE; }
    const int B = b->n_slots, G = (int)b->groups.size(), n = B * G;
    hipSetDevice(b->device);
    b->fl_jobs.assign((size_t)n, lio_scan_job{});
    b->fl_state.resize((size_t)n * 26);
    b->fl_cov.resize((size_t)n * 529);
    int first_err = 0, n_ready = 0;
    std::vector<char> ready((size_t)n, 0);
    for (int j = 0; j < n; j++) {
        lio_engine* e = b->groups[j / B].eng[j % B];
        lio_scan_job& job = b->fl_jobs[j];
        job.flags = LIO_JOB_IDLE;
        const void* d_raw = nullptr;
        uint32_t n_raw = 0;
        double beg = 0;
        const int fr = engine_fastlio_front(e, &d_raw, &n_raw, &beg, &b->fl_state[(size_t)j * 26], &b->fl_cov[(size_t)j * 529]);
        rc_out[j] = fr;
        if (fr < 0 && !first_err) first_err = fr;
        if (fr != 1000) continue;
        job.flags = 0;
        job.d_raw = d_raw;
        job.n_raw = n_raw;
        job.lidar_beg_time = beg;
        job.state_in = &b->fl_state[(size_t)j * 26];
        job.cov_in = &b->fl_cov[(size_t)j * 529];
        job.state_out = nullptr;  // (the posterior goes into the session's engine, where the reference's accessors read it)
        ready[j] = 1;
        n_ready++;
    }
    if (n_ready) {
        const int rc = lio_batch_sequences_step(b, b->fl_jobs.data(), n, nullptr);
        if (rc < 0 && !first_err) first_err = rc;
        for (int j = 0; j < n; j++) {
            if (!ready[j]) continue;
            rc_out[j] = b->fl_jobs[j].rc;
            engine_fastlio_back(b->groups[j / B].eng[j % B]);
        }
    }
    return first_err;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same device-resident loop for ONE engine (lio_engine_update, the filter update inside lio_engine_process_scan / lio_fastlio_main):
// one upload, (maximum_iter + 1) x {kNN, linearise, filter pass} on the engine's own stream, one download of the filter (state,
// covariance, pass logs), one wait -- instead of a host hand-over after every pass.  Falls back to the per-pass host loop for the rest
// of an update whose pass needs the N_eff < 23 dense branch.
struct lio_devloop {
    char* d_block = nullptr;
    char* h_block = nullptr;  // pinned: [SlotDesc][EskfDev] up, [EskfDev] down
    size_t block_bytes = 0;
    SlotDesc* d_desc = nullptr;
    SlotDesc* h_desc = nullptr;
    EskfDev* d_ctrl = nullptr;
    EskfDev* h_ctrl = nullptr;
    EskfDev* h_back = nullptr;  // pinned download target
    lio_batch_result* h_res = nullptr;
    lio_batch_result* h_res_dev = nullptr;
    // the loop as a graph, one per size class of the downsampled cloud (bucket k: grids for up to 2048 << k points): a graph captured for max_ds
    // (100 000) put 1 563 linearisation workgroups on a 7 000-point scan, 1 450 of them surplus
    static constexpr int kBuckets = 8;
    hipGraphExec_t exec[kBuckets] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t seq = 0;
    int stencil_n = 0;  // the graph holds the stencil by value and the table's address: re-captured when either changes
    const void* table = nullptr;  // (an LRU map swaps its table for a rebuilt twin now and then)
};

void devloop_destroy(lio_devloop* d) {
    if (!d) return;
    for (int k = 0; k < lio_devloop::kBuckets; k++)
        if (d->exec[k]) hipGraphExecDestroy(d->exec[k]);
    if (d->d_block) hipFree(d->d_block);
    if (d->h_block) hipHostFree(d->h_block);
    if (d->h_back) hipHostFree(d->h_back);
    if (d->h_res) hipHostFree(d->h_res);
    delete d;
}

lio_devloop* devloop_create(lio_scan* sc) {
    lio_devloop* d = new lio_devloop();
    d->block_bytes = sizeof(SlotDesc) + sizeof(EskfDev);
    bool ok = hipMalloc(reinterpret_cast<void**>(&d->d_block), d->block_bytes) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&d->h_block), d->block_bytes, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&d->h_back), sizeof(EskfDev), hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&d->h_res), sizeof(lio_batch_result), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer(reinterpret_cast<void**>(&d->h_res_dev), d->h_res, 0) == hipSuccess;
    if (!ok) { devloop_destroy(d); return nullptr; }
    d->d_desc = reinterpret_cast<SlotDesc*>(d->d_block);
    d->h_desc = reinterpret_cast<SlotDesc*>(d->h_block);
    d->d_ctrl = reinterpret_cast<EskfDev*>(d->d_block + sizeof(SlotDesc));
    d->h_ctrl = reinterpret_cast<EskfDev*>(d->h_block + sizeof(SlotDesc));
    memset(d->h_block, 0, d->block_bytes);
    memset(d->h_res, 0, sizeof(lio_batch_result));
    fill_desc(*d->h_desc, sc, d->d_ctrl, d->h_res_dev);
    return d;
}

// runs the update of the scan's current downsampled points from (x26, P529); on return *out holds the filter as the device left it
// (status EK_DONE, or EK_NEEDS_HOST with the loop state of the pass the host has to take over)
int devloop_update(lio_devloop* d, lio_map* m, lio_scan* sc, const double* x26, const double* P529, double R, int max_iter, int degenerate_detect_en,
                   const EskfDev** out) {
    SlotDesc& desc = *d->h_desc;
    desc.active = 1;
    desc.min_ds = 0;
    desc.reset_cache = 0;
    desc.n_raw = sc->n_raw;
    desc.nblocks = 0;
    desc.seq = ++d->seq;
    fill_ctrl(*d->h_ctrl, x26, P529, R, max_iter, degenerate_detect_en);
    const uint32_t bound = sc->have_ds > 0 ? (uint32_t)sc->have_ds : (sc->n_raw && sc->n_raw < sc->max_ds ? sc->n_raw : sc->max_ds);
    hipStream_t st = sc->stream;
    auto enqueue = [&](uint32_t ds_bound) -> int {
        LIO_HIP_TRY(hipMemcpyAsync(d->d_block, d->h_block, d->block_bytes, hipMemcpyHostToDevice, st));
        const int rc = p2plane_batch_update(m, st, d->d_desc, 1, ds_bound, max_iter + 1, nullptr, 0);
        if (rc != LIO_OK) return rc;
        LIO_HIP_TRY(hipMemcpyAsync(d->h_back, d->d_ctrl, sizeof(EskfDev), hipMemcpyDeviceToHost, st));
        return LIO_OK;
    };
    static const bool use_graph = []() { const char* k = getenv("LIO_BATCH_GRAPH"); return !(k && k[0] == '0'); }();
    if (use_graph && max_iter == 4) {
        if (d->stencil_n != m->stencil.n || d->table != m->table) {
            for (int k = 0; k < lio_devloop::kBuckets; k++)
                if (d->exec[k]) { hipGraphExecDestroy(d->exec[k]); d->exec[k] = nullptr; }
            d->stencil_n = m->stencil.n;
            d->table = m->table;
        }
        int bk = 0;
        while (bk < lio_devloop::kBuckets - 1 && (2048u << bk) < bound) bk++;
        uint32_t cap_b = 2048u << bk;
        if (cap_b > sc->max_ds || bk == lio_devloop::kBuckets - 1) cap_b = sc->max_ds;
        if (cap_b < bound) cap_b = bound;  // (never: bound <= max_ds)
        if (!d->exec[bk]) {
            hipGraph_t graph = nullptr;
            LIO_HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue(cap_b);
            const hipError_t e2 = hipStreamEndCapture(st, &graph);
            if (rc != LIO_OK) { if (graph) hipGraphDestroy(graph); return rc; }
            LIO_HIP_TRY(e2);
            LIO_HIP_TRY(hipGraphInstantiate(&d->exec[bk], graph, nullptr, nullptr, 0));
            hipGraphDestroy(graph);
        }
        LIO_HIP_TRY(hipGraphLaunch(d->exec[bk], st));
    } else {
        const int rc = enqueue(bound);
        if (rc != LIO_OK) return rc;
    }
    LIO_HIP_TRY(hipStreamSynchronize(st));
    *out = d->h_back;
    return LIO_OK;
}
