// lio_common.h -- internal declarations shared by the HIP translation units of liblio_hip.so.
// Not part of the ABI (that is include/lio_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lio_hip.h"
#include "eskf_dev.h"

struct LruEntry;

namespace lio {

// Loads that are meant to be in flight together: the optimiser sinks a load into the conditional block that consumes its value (and resolves the
// PHI of a conditionally loaded value by moves at the end of the predicated block), which turns `load a, load b, ... use` into load / s_waitcnt
// vmcnt(0) / use, one memory round trip after the other (tools/isa_load_chains.py shows where).  An empty asm statement that claims to rewrite
// the loaded registers pins the loads above it: written after a GROUP of unconditional loads it keeps the group together.
#if defined(__HIP_DEVICE_COMPILE__)
#define LIO_PIN4(r) asm volatile("" : "+v"((r).x), "+v"((r).y), "+v"((r).z), "+v"((r).w))
#define LIO_PIN1(r) asm volatile("" : "+v"(r))
#else
#define LIO_PIN4(r) ((void)(r))
#define LIO_PIN1(r) ((void)(r))
#endif
__host__ __device__ __forceinline__ void pin_loaded(float4& r) { LIO_PIN4(r); }
__host__ __device__ __forceinline__ void pin_loaded(uint4& r) { LIO_PIN4(r); }
__host__ __device__ __forceinline__ void pin_loaded(uint32_t& r) { LIO_PIN1(r); }
__host__ __device__ __forceinline__ void pin_loaded(float& r) { LIO_PIN1(r); }

// a double moved across lanes by one DPP control word (two 32-bit halves): quad_perm 0xB1 = [1,0,3,2], 0x4E = [2,3,0,1] give the quad sums
// (v0 + v1) + (v2 + v3) in every lane of a quad without LDS -- the first stage of the fixed-order workgroup reductions (p2plane.hip, ndt.hip)
template <int CTRL>
__device__ inline double dpp_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

void set_error(const char* fmt, ...);
void set_warning(const char* fmt, ...);  // notes that are not failures (lio_last_warning): never written by a successful call into lio_last_error

#define LIO_HIP_TRY(expr)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            ::lio::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return LIO_E_DEVICE;                                                                  \
        }                                                                                         \
    } while (0)

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr uint64_t kTombKey = 0xFFFFFFFFFFFFFFFEull;  // slot of an evicted voxel: probes walk on, inserts do not reuse it (tables are rebuilt)
constexpr uint32_t kNoIdx = 0xFFFFFFFFu;
constexpr int kMaxStencil = 75;
constexpr int kLinThreads = 64;   // linearize_kernel workgroup size (one partial-sum record per workgroup)
constexpr int kAcc = 29;          // 21 (JtJ upper) + 6 (Jtr) + sum|r| + count

// one open-addressing slot of the voxel hash grid: 16 B, one probe = one 16-B load
struct __attribute__((aligned(16))) Slot {
    unsigned long long key;  // packed 3 x 21-bit voxel coordinate, kEmptyKey when free
    uint32_t ptr;            // first point of the voxel in the pool (units of points)
    uint32_t cnt;            // points stored
};

// device-resident status / counters of a map
struct MapDev {
    unsigned long long pool_top;     // bump allocator over the point pool
    unsigned long long n_points;     // live points
    uint32_t n_voxels;
    uint32_t err;                    // bit0: table full, bit1: pool full
    uint32_t n_add;                  // staging count for map_incremental
    uint32_t n_tombs;                // evicted slots since the last table rebuild
    // LRU bookkeeping (ivox3d.h:231-256), only used when lio_map_set_lru enabled it
    unsigned long long log_head, log_tail, log_head_prev;  // touch log: entries appended / consumed / head before the current batch
    unsigned long long n_evicted;    // voxels evicted so far
    unsigned long long n_lru_interleaved;  // LRU-back voxels that the batch evicting around them also touched (see hashmap.hip)
    unsigned long long n_lru_recreated;    // ... of those, the ones the reference's point-by-point order drops and re-creates: done here too (lru_exact_*)
    unsigned long long n_lru_inexact;      // batches in which that order could not be followed (a young voxel at the back, more voxels than the quota before the batch, a full scratch list)
    int free_top[24];                // recycled pool regions by size class (floor(log2(capacity)))
    int free_in_top[24];             // regions freed by the grow kernel of the current batch (folded into free_top by lru_evict_kernel)
    // push_back order inside a voxel (ivox3d_node.hpp:87-90): every inserted point gets the running count of points offered to AddPoints before it
    // (mod 2^32, compared by signed difference); the tie-exact neighbour redo sorts a voxel's candidates by it (knn.hip, refsel.h)
    uint32_t seq_cur;                // first sequence number of the batch being inserted (set by its claim kernel)
    uint32_t seq_acc;                // ... of the next batch
    unsigned long long n_tie_boundary;    // queries whose fifth and sixth nearest were equally far: the set came from the reference's selection
    unsigned long long n_tie_unresolved;  // ... of those, the ones whose voxel lists did not fit the redo's staging area: canonical set kept
    uint32_t* touch_bits;            // diagnostic (the kNN kernel's counting variant): one bit per pool entry, cleared before every counted launch
    unsigned long long knn_cand[64 * 16];  // 64 shards, one 128-B line each (same-line atomics serialise in one L2 channel): word 0 = points resident in
                                           // the probed stencil voxels, word 1 = points the sweep loaded (diagnostic kernel variant only)
};

// rigid transforms handed to kernels by value (doubles, as the reference computes them)
struct PoseArgs {
    double qw[4];  // rot  (x,y,z,w)
    double tw[3];  // pos
    double ql[4];  // offset_R_L_I
    double tl[3];  // offset_T_L_I
};

// device-resident per-scan parameters (no host round trip between the downsample and its consumers)
struct ScanDev {
    uint32_t bbox_min[3];  // order-preserving uint encoding of float, init 0xFFFFFFFF
    uint32_t bbox_max[3];  // init 0
    uint32_t n_valid;      // finite input points
    uint32_t n_long;       // voxels queued for the wave-per-voxel centroid kernel
    uint32_t n_tie;        // kNN queries whose top-6 held an exact d2 tie (queued for the exact redo)
    uint32_t n_tie_done;   // snapshot of n_tie taken by the reporting workgroup of linearize_kernel
    uint32_t lin_ticket;   // arrival counter of linearize_kernel's workgroups (last arriver reports)
    uint32_t seq;          // sequence number of the last report
    uint32_t n_ds;         // feats_down_size
    uint32_t n_ds_prev;    // size of the neighbour cache before this scan (Nearest_Points.resize semantics)
    uint32_t cache_n;      // Nearest_Points.size(): moves only when a scan is registered (>= 5 downsampled points in fastlio_main)
    uint32_t passthrough;  // PCL int32 overflow guard hit: output = input
    uint32_t nbits;        // significant key bits for the radix sort
    uint32_t err;          // bit0: n_ds > max_ds
    int32_t minb[3];
    int32_t mul1, mul2;
    uint32_t total_cells;
    uint32_t n_monster;    // voxels with thousands of points, queued from the top of longlist for the wave-per-coordinate sums
};

struct StencilArgs {
    int n;
    signed char off[kMaxStencil][3];
};

// One scan of a batch (lio_batch_*): everything a kernel needs about it, resident in device memory.  The per-job words are rewritten by
// the host before every round (one small H2D copy for the whole batch); the rest are the buffers of the slot's lio_scan, fixed.
// Batch kernels are launched with blockIdx.y = slot and read their arguments from here (uniform scalar loads).
struct SlotDesc {
    const float4* raw;       // the job's cloud (device)
    uint32_t n_raw;
    uint32_t nblocks;        // ceil(n_raw / 2048): sort tiles of this scan
    uint32_t active;         // the slot has a job this round
    uint32_t seq;            // sequence number the result record of this round carries
    uint32_t max_ds, partial_blocks;
    uint32_t min_ds;         // scans that downsample to fewer points are not registered (5 in fastlio_main, 0 for a bare filter update)
    uint32_t reset_cache;    // the job is an independent scan: the slot's neighbour cache is forgotten before it (lio_scan_job.flags)
    ScanDev* sd;
    uint32_t *keys_a, *keys_b, *vals_a, *vals_b, *hist, *blockcnt, *hpos, *longlist, *tie_list;
    float4 *sorted, *ds_body, *ds_world, *nn_pts, *normvec;
    int32_t* nn_cnt;
    uint8_t* selected;
    double* partial;
    uint32_t* host_nds;      // mapped pinned words {n_ds, err, radix passes needed} of the slot's scan
    EskfDev* ctrl;           // device-resident filter of the slot
    lio_batch_result* result;  // mapped pinned host record of the slot
};

// Sequence mode of the batched engine (lio_batch_create_sequences): every slot is ONE SLAM session with its own map.  What the batched kernels
// need of slot s's map, and the per-round words of its map_incremental; uploaded with the descriptors before every round.
struct MapRef {
    Slot* table;
    uint32_t* cap;
    uint32_t* pending;
    float* created;
    float4* pool;
    uint32_t* pool_seq;              // per pool entry: insertion sequence number (null for maps that are not iVox maps)
    MapDev* md;
    uint32_t* slot_of_point;
    float4* stage;
    unsigned long long pool_cap;
    unsigned long long stamp_base;   // (n_batches + 1) << kStampIdxBits of the insert this round would be
    // LRU (null / 0 while the map has none)
    unsigned long long* touch;
    unsigned long long* prev_touch;
    struct ::LruEntry* lru_log;
    unsigned long long log_mask;
    uint32_t* free_items;
    uint32_t* free_in;
    unsigned long long* first_touch; // per slot: stamp_base + (index mask - index of the FIRST point of the batch in the voxel), atomicMax (null: the exact LRU order is off)
    uint32_t* lru_g;                 // scratch: the batch's voxel-creating points in point order
    uint32_t* lru_rec;               // scratch: slots of the voxels the batch drops and re-creates
    uint32_t free_cap;
    uint32_t lru_capacity;
    float lru_max_distance;
    uint32_t mask;
    float inv_res, res;
    int key_mode;
    uint32_t max_voxels;
    int tie_mode;                    // lio_map_set_tie_mode
    int stencil_id;                  // the map's stencil this round (a launch of the neighbour search serves the slots of one stencil)
    uint32_t do_insert;              // the round's scan enters the map when its update finishes on the device (map_incremental, laserMapping.cpp:1304)
    uint32_t ekf_inited;             // flg_EKF_inited of this scan (laserMapping.cpp:1201)
    float map_leaf;                  // filter_size_map_min
    // travel distance of the lidar origin before this scan (laserMapping.cpp:1288-1291); the round adds this scan's step on the device
    double travel_prev;
    double last_pos_lid[3];
};

// what the insert half of a sequence round decides / leaves per slot (device-resident; the tail is read back by the host)
struct SeqDev {
    double travel;       // travel after this scan (what AddPoints stamps new voxels with)
    uint32_t go;         // the update finished on the device (EK_DONE) and the scan enters the map this round
    uint32_t map_err;    // MapDev::err after the insert
    uint32_t n_add;      // points handed to AddPoints
    uint32_t n_voxels;
    unsigned long long n_points;
    double P[kEkN * kEkN];  // the posterior covariance (the result record in mapped memory carries the state only)
};

}  // namespace lio

struct LruEntry {  // one entry of the touch log: the voxel in `slot` was last touched at `stamp` -- unless touched again since
    unsigned long long stamp;
    uint32_t slot, pad;
};
constexpr int kStampIdxBits = 26;  // stamp = batch number << 26 | index of the point inside the batch
constexpr unsigned long long kStampIdxMask = (1ull << kStampIdxBits) - 1ull;
constexpr uint32_t kLruRecCap = 4096;  // voxels one batch may drop and re-create (more: the batch is counted in n_lru_inexact and handled as before)

struct lio_map {
    // LRU eviction (off while lru_capacity == 0)
    uint64_t lru_capacity;
    float lru_max_distance;
    unsigned long long* touch;   // per-slot stamp of the last AddPoints touch
    unsigned long long* prev_touch;  // the stamp it had before the current batch touched it
    LruEntry* lru_log;           // ring, ordered by stamp
    uint64_t lru_log_cap;        // power of two
    uint32_t* free_items;        // [24][free_cap] recycled region offsets
    uint32_t* free_in;           // [24][free_cap] regions outgrown during the current batch
    unsigned long long* first_touch;  // per slot: the FIRST point of the current batch in the voxel (see MapRef); lru_g / lru_rec: scratch of lru_evict_kernel's exact pass
    uint32_t *lru_g, *lru_rec;
    uint32_t free_cap;
    uint64_t tomb_bound;         // evictions possible since the last rebuild (host-side upper bound)
    lio::Slot* table2;           // second set of per-slot arrays: target of a table rebuild, then swapped
    uint32_t *cap2, *pending2, *remap;
    float* created2;
    unsigned long long *touch2, *prev_touch2;
    int device;
    hipStream_t stream;
    float res, inv_res;
    int key_mode;  // 0: iVox key round(p / res); 1: fast_gicp Gaussian-voxel key floor(p / res - 0.5) in f32; 2: the same in f64 (its CPU voxel map)
    uint64_t max_points, max_voxels, pool_cap;
    uint32_t table_cap, table_mask;  // power of two
    lio::Slot* table;
    uint32_t* cap;       // per-slot capacity of the voxel's pool region
    uint32_t* pending;   // per-slot staging counter of a batch insert
    float* created;      // per-slot travel distance at creation (ivox3d.h:240)
    float4* pool;
    uint32_t* pool_seq;  // insertion sequence number of every pool entry (iVox maps only: key_mode 0)
    uint32_t* touch_bits;  // one bit per pool entry for the counting variant of the kNN kernel (allocated by lio_batch_enable_kernel_timing(b, 2))
    int tie_mode;        // 1 (default): equally distant candidates at the fifth place are kept as the reference keeps them; 0: smallest (d2, x, y, z)
    lio::MapDev* dev;
    lio::MapDev* host_dev;  // pinned mirror
    // map_incremental enqueued on the map's own stream and not yet looked at by the host (map_incremental_async / map_settle)
    hipEvent_t ev_classified = nullptr, ev_inserted = nullptr;
    bool insert_pending = false;
    uint32_t settled_n_add = 0;
    unsigned long long* tile_sum;  // prebuilt-map layout scan scratch
    uint32_t* slot_of_point;  // batch insert scratch
    uint64_t slot_of_point_cap;
    float4* stage;            // staging for host->device inserts and map_incremental
    uint64_t stage_cap;
    lio::StencilArgs stencil;
    int stencil_id;
    int own_stream;
    uint64_t n_batches;
    uint64_t bytes;
};

namespace lio {
struct KernelTimer;  // capi.hip
}
struct lio_scan {
    lio::KernelTimer* kt;
    int force_degeneracy;  // evaluate the degeneracy sums even when the eigenvalue bound makes them moot
    int device;
    hipStream_t stream;
    uint32_t max_raw, max_ds;
    uint32_t n_raw;
    const float4* raw;   // points to raw_own or to a caller-owned device buffer
    float4* raw_own;
    float4* ds_body;
    float4* ds_world;
    float4* nn_pts;      // SoA: 5 planes of max_ds float4
    int32_t* nn_cnt;
    uint8_t* selected;
    float4* normvec;
    uint32_t *keys_a, *keys_b, *vals_a, *vals_b;
    uint32_t* hist;      // radix histograms [nblocks][256]
    uint32_t* blockcnt;  // head counts per tile
    uint32_t* hpos;      // first sorted position of every occupied voxel
    uint32_t* longlist;  // voxels with long runs
    uint32_t* tie_list;  // query indices with exact d2 ties
    float4* sorted;      // raw points gathered into (voxel, input index) order
    double* partial;     // per-block partial sums
    uint32_t partial_blocks;
    lio::ScanDev* dev;
    lio::ScanDev* host_dev;       // pinned mirror
    lio_normal_eq* d_result;
    uint32_t* host_nds;           // pinned, mapped: {n_ds, err, radix passes the scan needed} written by vg_heads_kernel
    int pred_passes;              // radix passes the last waited-for downsample needed (4 until known)
    lio::SlotDesc* d_batch_desc = nullptr;  // lio_scan_voxel_downsample_batch with this scan first: the descriptor rows of the call
    lio::SlotDesc* h_batch_desc = nullptr;  // (pinned)
    uint32_t batch_desc_cap = 0;
    uint32_t* host_nds_dev;
    lio_normal_eq* h_result;      // pinned, mapped
    lio_normal_eq* h_result_dev;  // device-side alias of h_result (linearize_kernel's last workgroup writes the record there)
    uint32_t seq_expected;        // sequence number the next report will carry
    int have_ds;
    uint32_t resize_min;  // scans that downsample to fewer points leave the neighbour cache alone (5 on an engine's scan: laserMapping.cpp:1250-1274)
    uint64_t bytes;
};

namespace lio {
// kernels' host-side launchers (defined in the .hip files)
// ---- IMU front half (undistort.hip) ----
constexpr int kMaxImuPoses = 128;
struct ImuPoseDev {  // Pose6D of common_lib.h:31-39 in f64, as UndistortPcl fills it (IMU_Processing.hpp:284,337,357)
    double off;      // offset_time from the scan start, s
    double acc[3], gyr[3], vel[3], pos[3], R[9];
};
struct UndistortArgs {
    double pos_e[3], rot_e[4], ril[4], til[3];  // state at the scan end
    double blind2;
    int n_poses, filter_num, undistort;
};
struct DeltaArgs {  // undistortPoints(delta_pose, ...): delta translation, angle * axis of the delta rotation (f32), scan period
    float t[3];
    float aa[3];
    double scan_period;
};
constexpr int kMaxPoseList = 64;
struct PoseListArgs {  // undistortPoints(poses, points): interval ends (unsigned us since the header stamp) and the motion of every pose since poses[0]
    int n_poses;
    unsigned long long limit[kMaxPoseList];
    DeltaArgs d[kMaxPoseList];
};
int undistort_poses_launch(hipStream_t stream, const float4* d_in, const uint32_t* d_stamp_us, uint32_t n, float4* d_out, const PoseListArgs& args,
                           uint32_t* d_tile_max);
int undistort_delta_launch(hipStream_t stream, const float4* d_in, const uint32_t* d_stamp_us, uint32_t n, float4* d_out, const DeltaArgs& args);
int undistort_launch(hipStream_t stream, const float4* d_in, const uint32_t* d_stamp_us, uint32_t n, float4* d_out, const ImuPoseDev* d_poses,
                     const UndistortArgs& args, unsigned long long* d_block_min /* ceil(n / 256) words of scratch */);

int vg_downsample(lio_scan* s, float leaf, int passes /* radix passes to launch, 1..4 */);
int vg_downsample_batch(hipStream_t st, const SlotDesc* d_slots, int n_slots, uint32_t max_raw, uint32_t max_ds, float leaf, int passes);
int knn_batch_launch(lio_map* m, hipStream_t st, const SlotDesc* d_slots, int n_slots, uint32_t grid_x, int count_touched);
// live kernel timing of the batched chain (bench.py's roofline leg): HIP events on the stream the kernels are launched on, per class
struct BatchTimer {
    static constexpr int kClasses = 5;  // 0 downsample chain, 1 stencil kNN, 2 linearise, 3 filter pass, 4 map_incremental (sequence mode)
    static constexpr int kPool = 256;
    hipEvent_t ev[kClasses][kPool][2];
    int used[kClasses] = {0, 0, 0, 0, 0};
    double us[kClasses] = {0, 0, 0, 0, 0};
    uint32_t launches[kClasses] = {0, 0, 0, 0, 0};
    bool on = false, created = false;
    hipStream_t stream = nullptr;
    void begin(int c) { if (on && used[c] < kPool) hipEventRecord(ev[c][used[c]][0], stream); }
    void end(int c) { if (on && used[c] < kPool) { hipEventRecord(ev[c][used[c]][1], stream); used[c]++; } }
    void resolve() {  // after the stream was waited for
        for (int c = 0; c < kClasses; c++) {
            for (int i = 0; i < used[c]; i++) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev[c][i][0], ev[c][i][1]) == hipSuccess) { us[c] += (double)ms * 1000.0; launches[c]++; }
            }
            used[c] = 0;
        }
    }
};
int p2plane_batch_update(lio_map* m, hipStream_t st, const SlotDesc* d_slots, int n_slots, uint32_t ds_bound, int n_passes, BatchTimer* bt, int count_touched);
// sequence mode: the same loop with every slot's neighbour search against ITS map (one launch per stencil among `stencils`), then the
// map_incremental of every slot whose update finished on the device, then the read-back records
int knn_seq_launch(hipStream_t st, const MapRef* d_maps, const SlotDesc* d_slots, int n_slots, uint32_t grid_x, const StencilArgs* stencils, const int* stencil_ids,
                   int n_stencils);
int p2plane_seq_update(hipStream_t st, const MapRef* d_maps, const SlotDesc* d_slots, int n_slots, uint32_t ds_bound, int n_passes, const StencilArgs* stencils,
                       const int* stencil_ids, int n_stencils, BatchTimer* bt);
int p2plane_seq_insert(hipStream_t st, const MapRef* d_maps, const SlotDesc* d_slots, SeqDev* d_seq, int n_slots, uint32_t ds_bound, int any_lru);
int map_insert_seq(hipStream_t st, const MapRef* d_maps, const SeqDev* d_seq, int n_slots, uint32_t bound, int any_lru);
int map_rebuild(lio_map* m, hipStream_t stream);
}
struct lio_comm;
namespace lio {
int p2plane_batch_share(hipStream_t st, const SlotDesc* d_descs, int n_slots, int n_maps, uint32_t ds_bound);
int p2plane_batch_update_joint(lio_map** maps, int n_maps, ::lio_comm* comm, int world, int (*gather_hook)(void*, const double*, double*, uint32_t, void*), void* gather_ctx, hipStream_t st, const SlotDesc* d_descs, int n_slots, uint32_t ds_bound,
                               int n_passes, double* d_local32, double* d_gathered, BatchTimer* bt, const MapRef* d_rowmaps = nullptr);
int scan_begin_rows(hipStream_t st, const SlotDesc* d_descs, int n_rows);
size_t ds_exchange_slot_bytes(uint32_t cap);
int p2plane_batch_exchange_ds(::lio_comm* comm, int world, int (*gather_hook)(void*, const double*, double*, uint32_t, void*), void* gather_ctx, hipStream_t st,
                              const SlotDesc* d_descs, int n_slots, int per, int s0, int s1, uint32_t cap, char* d_send, char* d_all);
int scan_begin(lio_scan* s);
int scan_set_nds(lio_scan* s, uint32_t n);
void kt_begin(lio_scan* s, int which);
void kt_end(lio_scan* s, int which);
int map_enable_touch_bits(lio_map* m);
lio_map* map_create_mode(int device, float resolution, uint64_t max_points, uint64_t max_voxels, int stencil, int key_mode);
int map_insert_dev(lio_map* m, hipStream_t stream, const float4* d_pts, uint64_t n, const uint32_t* d_n, double travel);
int map_knn_plane(lio_map* m, lio_scan* s, const PoseArgs& pose, int redo_knn);
int knn_batch(lio_map* m, const float4* d_q, uint32_t n, float4* d_out, int32_t* d_cnt, uint32_t* d_tie);
int map_knn_exact(lio_map* m, lio_scan* s, const PoseArgs& pose, uint32_t n_tie_host);
int p2plane_reduce(lio_map* m, lio_scan* s, const PoseArgs& pose, int redo_knn);
int p2plane_degeneracy(lio_scan* s);
int incremental_classify(lio_map* m, lio_scan* s, const PoseArgs& pose, float map_leaf, int ekf_inited, int seed_all);
PoseArgs make_pose(const double pose_wi[7], const double ext_il[7]);
}  // namespace lio

// internal entry points defined in capi.hip's extern "C" block (not part of include/lio_hip.h)
extern "C" {
int p2plane_linearize_begin(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], int redo_knn);
int p2plane_linearize_end(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], int redo_knn, lio_normal_eq* out);
int scan_share_ds(lio_scan* dst, lio_scan* src, uint32_t n);
int scan_forget_cache(lio_scan* s);
int map_clear(lio_map* m);
// map_incremental without the host wait: the points to add are chosen on the scan's stream (classify), the insert chain (+ LRU) and the
// status read-back run on the MAP's stream behind an event, and the host looks at the outcome in map_settle -- which everything that reads
// the map from another stream calls first (the next scan's neighbour search: by then its downsample has run beside the insert)
int map_incremental_async(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], float map_leaf, int ekf_inited, double travel);
int map_settle(lio_map* m);
int map_settle_if_done(lio_map* m);
}
