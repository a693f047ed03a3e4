/*
 * lio_hip.h -- C ABI of the MI355X-native LIO scan-matching core (liblio_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of w111liang222/lidar-slam-detection:
 * the per-scan  voxel-grid downsample -> map kNN -> point-to-plane residual + Jacobian
 * accumulation -> iterated ESKF update -> map insert  loop of the FastLIO frontend, and
 * the callers either side of it:
 *   lio_map_* / lio_scan_* / lio_p2plane_* ... the kernels' level (iVox, VoxelGrid, h_share_model)
 *   lio_engine_* ........................... fastlio_main after IMU processing, the 23-DoF filter
 *   lio_fastlio_* .......................... the reference's FastLIO entry points one to one (IMU front half included)
 *   lio_engines_process_batch, lio_engine_set_reduce_hook ... throughput mode, joint registration across GPUs
 *   lio_ndt_* / lio_pose_estimator_* / lio_localmap_* ........ the localisation mode's matcher, its filter, its local map
 *   lio_state_* / lio_eskf_update_cb ....... host-only helpers that pin the filter algebra
 * Every entry point names the reference interface it replaces (paths relative to
 * /root/reference/slam/mapping/fastlio unless stated otherwise).
 *
 * Conventions
 *   - plain C, opaque handles, caller-owned buffers, no exceptions cross the ABI;
 *   - return value: >= 0 ok (often a count), < 0 one of LIO_E_*;
 *   - points are XYZI float quadruples (16 B, "float4"); quaternions are (x, y, z, w);
 *   - "_device" variants take pointers that are already resident in HBM on the handle's
 *     device (e.g. a torch tensor's data_ptr()); the others take host pointers;
 *   - a handle owns one HIP stream; a handle is not thread-safe; distinct handles may be
 *     driven from distinct host threads / on distinct GPUs.
 *   - there is NO CPU fallback: without a usable HIP device every create() returns NULL.
 */
#ifndef LIO_HIP_H_
#define LIO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIO_OK 0
#define LIO_E_INVALID (-1)     /* bad argument / handle */
#define LIO_E_CAPACITY (-2)    /* a fixed capacity (points, voxels, ds points) would be exceeded */
#define LIO_E_DEVICE (-3)      /* HIP runtime error (message via lio_last_error) */
#define LIO_E_STATE (-4)       /* call order violated (e.g. linearize before a scan is set) */

typedef struct lio_map lio_map;       /* iVox-equivalent hash-grid map resident in HBM */
typedef struct lio_scan lio_scan;     /* one scan in flight: raw, downsampled, neighbour cache, gates */
typedef struct lio_engine lio_engine; /* per-scan driver = the body of fastlio_main after IMU processing */

const char* lio_last_error(void);
/* notes of calls that SUCCEEDED (thread-local, like lio_last_error, which only failures write): e.g. lio_batch_create when GPU_MAX_HW_QUEUES
 * leaves its rounds in flight sharing hardware queues */
const char* lio_last_warning(void);
int lio_device_count(void);
/* The ABI revision this header describes; lio_abi_version() returns the revision the loaded library was built from.  A caller built against an
 * older header can keep running as long as it checks that the library's revision is >= its own: revisions only APPEND (struct tails, flag bits,
 * entry points).  History: 4 = round 4 (lio_batch_times grew by insert_us / insert_launches / pad: a caller that allocates the round-3 struct and
 * calls a revision-4 library is written past -- rebuild, or check the revision and allocate sizeof(lio_batch_times) of THIS header;
 * lio_timings.n_added may be -1 = "insert still in flight, not read back"; lio_engine_timings may return a deferred LIO_E_CAPACITY);
 * 5 = round 5 (LIO_JOB_HOST_RAW, lio_pinned_alloc / lio_pinned_free, lio_abi_version itself);
 * 6 = round 6 (lio_map_set_tie_mode / lio_map_tie_stats: candidates exactly as far as the fifth nearest are now kept as the reference keeps them). */
#define LIO_ABI_VERSION 6
int lio_abi_version(void);
/* page-locked host memory for clouds handed over with LIO_JOB_HOST_RAW (or lio_scan_upload): copies from it run at the link's rate and
 * overlap with kernels; NULL on failure.  Any hipHostMalloc'ed / hipHostRegister'ed range serves as well. */
void* lio_pinned_alloc(uint64_t bytes);
void lio_pinned_free(void* p);
/* bytes of HBM currently held by a map / scan handle (capacity planning on the 288 GB part) */
uint64_t lio_map_bytes(const lio_map*);

/* ---------------------------------------------------------------------------------------------
 * Map: replaces faster_lio::IVox<3, DEFAULT, PointType>  (include/ivox3d/ivox3d.h:59-120)
 *   voxel key = round-half-away(p / resolution) per axis (ivox3d.h:258-261)
 *   stencils  = 1 (CENTER), 7 (NEARBY6), 19 (NEARBY18), 27 (NEARBY26), 75 ("NEARBY74", 5x5x3)
 *               (ivox3d.h:179-210)
 * ------------------------------------------------------------------------------------------- */
/* ctor: IVox(Options) as configured at src/laserMapping.cpp:1060-1064.  max_points bounds the point
 * pool (the pool is 2x that to leave slack for voxel growth), max_voxels the number of live voxels. */
lio_map* lio_map_create(int device, float resolution, uint64_t max_points, uint64_t max_voxels, int stencil);
void lio_map_destroy(lio_map*);
/* IVox::SetNearByType (used at src/laserMapping.cpp:1241-1243) */
int lio_map_set_stencil(lio_map*, int stencil);
/* IVox::AddPoints(points, travel_distance)  (ivox3d.h:231-256).  Points are appended to their voxels;
 * new voxels are stamped with `travel`.  The LRU list of the reference (least recently touched voxel dropped
 * above `capacity` voxels once it is older than `max_distance`) is opt-in: lio_map_set_lru below; without it
 * nothing is ever dropped and exceeding max_points / max_voxels returns LIO_E_CAPACITY. */
int lio_map_insert(lio_map*, const float* world_xyzi, uint64_t n, double travel);
int lio_map_insert_device(lio_map*, const void* d_world_xyzi, uint64_t n, double travel);
/* back to an empty map without giving its memory back: what `ivox = std::make_shared<IVoxType>(ivox_options)` does in fastlio_init (src/laserMapping.cpp:1064) when the
 * reference starts over -- here also the way a checker puts a map of known content in place (clear, then one lio_map_insert of that content: points of a
 * voxel keep the order of the inserted array, as IVoxNode::InsertPoint's push_back does).  The LRU list, if on, starts over with the map. */
int lio_map_clear(lio_map*);
/* IVox::Options capacity_ / max_distance_ (ivox3d.h:46-52; 100000 voxels / 100 m at src/laserMapping.cpp:1060-1064): turn
 * on the LRU list of IVox::AddPoints (ivox3d.h:231-256) -- after every inserted point the least recently touched voxel is
 * dropped while the map holds more than `capacity_voxels` voxels and that voxel was created more than `max_distance` of
 * travel ago.  Call before the first insert; `max_voxels` of lio_map_create stays the hard limit and must be larger (the map
 * overshoots its capacity while nothing is old enough to go).  Off by default (nothing is ever dropped).
 * lio_map_lru_stats: voxels evicted so far, and an UPPER BOUND on the back-of-list voxels that were touched by the very
 * batch that was evicting around them.  If such a voxel's first point of the batch comes after its turn to go, the reference's
 * point-by-point order drops it and creates it again holding the batch's points alone: since ABI revision 6 so does the device
 * (the pops of a batch are replayed in order, csrc/hashmap.hip lru_exact_*; LIO_LRU_EXACT=0 in the environment when the map is
 * made: counted only, the voxel keeps its points).  lio_map_lru_exact_stats: voxels dropped and re-created that way, and the
 * batches in which the order could not be followed -- the map above its capacity before the batch, or a voxel younger than
 * max_distance at the back of the list, where pops no longer coincide with creations -- and the batch was handled as a whole. */
int lio_map_set_lru(lio_map*, uint64_t capacity_voxels, double max_distance);
int lio_map_lru_stats(lio_map*, uint64_t* n_evicted, uint64_t* n_interleaved);
int lio_map_lru_exact_stats(lio_map*, uint64_t* n_recreated, uint64_t* n_batches_not_followed);
/* Which five, when the fifth and the sixth nearest candidate of a query are EXACTLY equally far (f32 d2).  IVox::GetClosestPoint cuts every stencil
 * voxel's in-range points to five and then the whole list to five with std::nth_element on the distance alone (ivox3d_node.hpp:107-127,
 * ivox3d.h:156-164): which of the equally distant candidates survives is what libstdc++'s introselect does to that particular sequence (stencil
 * order, push_back order inside a voxel).  mode 1 (default): the same survivor -- the map keeps every point's push_back rank, and the exact redo
 * of tied queries runs the same selection on the same sequence (csrc/refsel.h); mode 0: the five smallest in (d2, x, y, z), the definition of
 * rounds 1-5.  The two differ only on such ties (about one query in 1e6 on sensor data); the returned lists are in the canonical order either way.
 * mode 2: the lists exactly as the reference returns them, ORDER included (nearest first, the rest as introselect leaves them) -- every query of
 * every search is redone by the reference's selection, tens of times slower than the search itself: a parity mode (the plane fit's QR sees its rows
 * in the reference's order, so the whole path follows the reference's build to the rounding of the f64 sums), not a production mode.
 * lio_map_tie_stats: queries whose set the selection decided so far, and how many of those could not be resolved (a stencil voxel of more
 * than 2560 points: the canonical set was kept) -- 0 unless a map holds voxels of thousands of points. */
int lio_map_set_tie_mode(lio_map*, int mode);
int lio_map_tie_stats(lio_map*, uint64_t* n_boundary_ties, uint64_t* n_unresolved);
/* capacity planning: slots of the point pool handed out so far by the bump allocator (recycled regions of evicted / outgrown
 * voxels are re-used first and do not move it) and the pool's size, both in points of 16 B */
int lio_map_pool_stats(lio_map*, uint64_t* pool_top, uint64_t* pool_cap);
/* IVox::NumValidGrids (ivox3d.h:173-176) and the total number of stored points */
int lio_map_stats(lio_map*, uint64_t* n_points, uint64_t* n_voxels);
/* running total of map points visited by stencil kNN queries (the C-bar * N_ds statistic of the roofline model) */
uint64_t lio_map_knn_candidates(lio_map*);
/* running total of map points whose 16 bytes the kNN sweep actually asked for (it prunes stencil voxels that cannot hold one of the five
 * nearest, exactly) -- counted only by the diagnostic kernel variant that lio_batch_enable_kernel_timing(b, 2) selects; 0 otherwise */
uint64_t lio_map_knn_touched(lio_map*);
/* ... and of those the DISTINCT ones per launch, summed over the counted launches (a bitmap over the point pool, cleared before every counted launch):
 * the bytes a launch cannot avoid moving once -- 16 B for every distinct candidate point -- whatever its caches do */
uint64_t lio_map_knn_unique(lio_map*);
/* all stored points, voxel by voxel in unspecified order; returns the count or -(needed) */
int64_t lio_map_dump(lio_map*, float* out_xyzi, uint64_t cap_points);
/* IVox::GetClosestPoint(pt, out, 5, 5.0) for a batch of world-frame queries (ivox3d.h:139-171):
 * out_pts is n x 5 x 4 floats -- the reference's five (lio_map_set_tie_mode) in the canonical order (d2, x, y, z) ascending, out_cnt[n] the number
 * found (0..5).  Test/diagnostic entry; the per-scan path uses lio_p2plane_linearize. */
int lio_map_knn(lio_map*, const float* world_xyzi, uint32_t n, float* out_pts, int32_t* out_cnt);

/* ---------------------------------------------------------------------------------------------
 * Scan: replaces the file-scope per-scan buffers feats_undistort / feats_down_body /
 * feats_down_world / Nearest_Points / point_selected_surf / normvec / res_last
 * (src/laserMapping.cpp:86-124) and downSizeFilterSurf (src/laserMapping.cpp:127,1206-1207).
 * The neighbour cache persists across scans exactly like Nearest_Points does.
 * ------------------------------------------------------------------------------------------- */
lio_scan* lio_scan_create(int device, uint32_t max_raw, uint32_t max_ds);
void lio_scan_destroy(lio_scan*);
/* forget the neighbour cache and gate flags, as fastlio_init() does for Nearest_Points / point_selected_surf
 * (src/laserMapping.cpp:1045-1047) */
int lio_scan_reset(lio_scan*);
int lio_scan_upload(lio_scan*, const float* body_xyzi, uint32_t n_raw);          /* host -> HBM */
int lio_scan_set_device(lio_scan*, const void* d_body_xyzi, uint32_t n_raw);     /* already in HBM (not copied) */
/* pcl::VoxelGrid<PointType>::filter with setLeafSize(leaf, leaf, leaf) (PCL 1.9.1 voxel_grid.hpp,
 * called at src/laserMapping.cpp:1206-1207): centroid of every occupied voxel, ascending voxel index.
 * Asynchronous; *n_ds (may be NULL) is only filled when sync != 0. */
/* undistortPoints(const Eigen::Matrix4f& delta_pose, PointCloudAttrPtr&, double scan_period)   slam/common/slam_utils.cpp:163-191,
 * called by the localisation mode before the downsample (hdl_localization_nodelet.cpp:224-227) with delta_pose = start^-1 * stop of the
 * filter's prediction: constant-velocity motion compensation of the uploaded cloud, p' = Exp(r * log(delta)) p with r = stamp / period,
 * all in f32 as there.  delta_pose is row-major; stamp_us[i] is PointAttr::stamp (us from the scan start), a host array
 * (stamps_on_device = 0) or a device array.  Runs on the scan's stream; the result replaces the scan's raw cloud (a caller-owned
 * device cloud given by lio_scan_set_device is not written).  lio_scan_download_raw returns the point count. */
int lio_scan_undistort_delta(lio_scan*, const uint32_t* stamp_us, int stamps_on_device, const float delta_pose[16], double scan_period);
int lio_scan_download_raw(lio_scan*, float* out_xyzi, uint32_t cap);
/* undistortPoints(std::vector<PoseType>& poses, PointCloudAttrPtr&)   slam/common/slam_utils.cpp:193-228 (graph back end and map export: the
 * cloud compensated with a LIST of poses, e.g. the IMU poses of a frame): poses[i] = (absolute stamp us, motion T since poses[0], row-major
 * 4 x 4), i = 0 is the scan start; a point uses the first pose interval, from the one its predecessor used onwards, that ends
 * (pose_stamp[i] - header_stamp, unsigned) not before its stamp; points past the last pose, and everything after them, are left alone.
 * The interval ends must not decrease (LIO_E_INVALID otherwise); at most 64 poses. */
int lio_scan_undistort_poses(lio_scan*, const uint32_t* stamp_us, int stamps_on_device, uint64_t header_stamp_us, const uint64_t* pose_stamp_us,
                             const double* pose_T, uint32_t n_poses);
int lio_scan_voxel_downsample(lio_scan*, float leaf, int sync, uint32_t* n_ds);
/* the same filter over n scans (each holding its raw cloud: lio_scan_upload / lio_scan_set_device) with ONE set of launches -- the batched
 * chain of lio_batch, blockIdx.y = scan -- for callers that hold many clouds at once (the candidate key frames of the map-merge tools,
 * overlap_merge.hpp:158-179; relocalisation; offline re-registration): lio_ndt_align_batch takes the scans as they leave here.  Waits for the
 * result; n_ds[i] (may be NULL) = points of scan i.  Per scan identical to lio_scan_voxel_downsample */
int lio_scan_voxel_downsample_batch(lio_scan** scans, int n, float leaf, uint32_t* n_ds);
/* bypass the filter: use these points as feats_down_body (tests, staged pipelines) */
int lio_scan_set_ds(lio_scan*, const float* ds_body_xyzi, uint32_t n_ds);
int lio_scan_num_ds(lio_scan*);                                                   /* syncs the stream */
int lio_scan_download_ds(lio_scan*, float* out_xyzi, uint32_t cap);              /* feats_down_body */
int lio_scan_download_world(lio_scan*, float* out_xyzi, uint32_t cap);           /* feats_down_world */
/* per-point results of the last linearize: any pointer may be NULL.
 * selected[n_ds] (point_selected_surf), normvec[n_ds*4] (n, pd2), nn_cnt[n_ds], nn_pts[n_ds*5*4] */
int lio_scan_download_match(lio_scan*, uint8_t* selected, float* normvec, int32_t* nn_cnt, float* nn_pts);

/* live per-kernel timing with HIP events recorded on the handle's stream around every launch of the three
 * per-pass kernels (bench.py's roofline leg).  Off by default; costs two event records per launch when on. */
typedef struct lio_kernel_times {
    double knn_us, linearize_us, finalize_us;          /* summed device time */
    uint32_t knn_launches, linearize_launches, finalize_launches;
    uint32_t pad;
} lio_kernel_times;
/* on = bit mask of the kernel classes to time: 1 kNN (what bench.py's timed region uses: two event records per
 * kNN launch), 2 linearize (+ report), 0 = off */
int lio_scan_enable_kernel_timing(lio_scan*, int on);
int lio_scan_kernel_times(lio_scan*, lio_kernel_times* out, int reset);

/* normal equations of one pass, all f64, reduced in a fixed order (run-to-run identical) */
typedef struct lio_normal_eq {
    double JtJ[36];      /* sum row6 row6^T, row6 = [n, (R_il p + t_il) x (R_wi^T n)]  (src/laserMapping.cpp:909-931) */
    double Jtr[6];       /* sum row6 * (-pd2)                                               */
    double nnT[9];       /* sum n n^T  (HTH of src/laserMapping.cpp:937)                     */
    double eigvec[9];    /* eigenvectors of nnT as columns, ascending eigenvalue (row-major 3x3) */
    double eigval[3];
    double contri[3];    /* per eigenvector: sum |n^.v| over rows with |n^.v| > 0.1736 (src/laserMapping.cpp:946-964); +inf when
                            the bound eigval[i] - 0.1736^2 n_eff >= 250 already proves "not degenerate" and the pass was skipped */
    double strong[3];    /* ... > 0.7070 */
    double sum_abs_res;  /* total_residual (src/laserMapping.cpp:884) */
    uint32_t n_eff;      /* effct_feat_num */
    uint32_t n_ds;       /* feats_down_size */
    uint32_t n_knn_candidates_lo, n_knn_candidates_hi; /* 64-bit count of in-stencil points visited (kNN passes) */
    uint32_t n_tie;      /* kNN queries redone with the exact (d2, x, y, z) comparison because of an exact d2 tie */
    uint32_t seq;        /* sequence number of this record (the host-side wait spins on it) */
} lio_normal_eq;

/* One evaluation of h_share_model_geometric (src/laserMapping.cpp:813-932) without the host-side
 * degeneracy projection: body->world transform, [redo_knn: stencil kNN into the neighbour cache],
 * esti_plane (include/common_lib.h:236-268), residual gate, and the f64 accumulation of the
 * 6-column Jacobian blocks.  pose_wi = (t_wi[3], q_wi[4]);  ext_il = (t_il[3], q_il[4]). */
int lio_p2plane_linearize(lio_map*, lio_scan*, const double pose_wi[7], const double ext_il[7], int redo_knn,
                          lio_normal_eq* out);
/* how lio_p2plane_linearize treats the degeneracy sums: 0 = auto (evaluated only when the eigenvalue bound does not
 * decide, default), 1 = always, 2 = never (the caller evaluates them itself, e.g. after a cross-GPU reduction) */
int lio_scan_set_degeneracy_mode(lio_scan*, int mode);
/* the degeneracy sums of the last linearize against the given eigenvectors (columns of a row-major 3x3) */
int lio_p2plane_degeneracy(lio_scan*, const double V[9], double contri[3], double strong[3]);
/* rows of the last linearize for the (rare) N_eff < 23 branch of the filter
 * (esekfom.hpp:1715-1744): h_x[n_eff*6] (first six columns) and h[n_eff], selected points in index order */
int lio_p2plane_rows(lio_scan*, const double pose_wi[7], const double ext_il[7], double* h_x6, double* h, uint32_t cap_rows);
/* map_incremental (src/laserMapping.cpp:523-576) with the final state: world transform, need_add test
 * against the cached neighbours, insert.  Returns the number of points inserted. */
int lio_map_incremental(lio_map*, lio_scan*, const double pose_wi[7], const double ext_il[7], float map_leaf,
                        int ekf_inited, double travel);
/* first-scan seeding (src/laserMapping.cpp:1227-1238): insert every downsampled point */
int lio_map_seed(lio_map*, lio_scan*, const double pose_wi[7], const double ext_il[7], double travel);

/* ---------------------------------------------------------------------------------------------
 * Engine: replaces the part of fastlio_main() that follows p_imu->Process
 * (src/laserMapping.cpp:1189-1304) together with esekf::update_iterated_dyn_share_modified
 * (include/IKFoM_toolkit/esekfom/esekfom.hpp:1619-1931) and the constants of fastlio_init
 * (src/laserMapping.cpp:1025-1124).  Host C++ (filter algebra, 23 DoF) over the calls above.
 * State vector (26 doubles): pos3 rot4 R_il4 t_il3 vel3 bg3 ba3 grav3  (use-ikfom.hpp:12-21).
 * ------------------------------------------------------------------------------------------- */
lio_engine* lio_engine_create(int device, float resolution, int stencil, uint64_t max_points, uint64_t max_voxels,
                              uint32_t max_raw, uint32_t max_ds);
/* an engine (state + covariance + scan buffers + stream) that registers scans against a map owned by someone
 * else.  Several such engines may run concurrently from different host threads on one map: the map is then
 * read-only (static-map mode is forced; map_incremental is skipped).  Destroying the engine leaves the map. */
lio_engine* lio_engine_create_shared(lio_map* shared_map, uint32_t max_raw, uint32_t max_ds);
void lio_engine_destroy(lio_engine*);
lio_map* lio_engine_map(lio_engine*);
lio_scan* lio_engine_scan(lio_engine*);
int lio_engine_set_state(lio_engine*, const double s26[26]);
int lio_engine_get_state(lio_engine*, double s26[26]);
int lio_engine_set_cov(lio_engine*, const double P[529]);
int lio_engine_get_cov(lio_engine*, double P[529]);
/* flg_EKF_inited, flg_first_scan, travel_distance, first_lidar_time (src/laserMapping.cpp:97-121) */
int lio_engine_set_flags(lio_engine*, int ekf_inited, int first_scan, double travel, double first_lidar_time);
double lio_engine_travel(lio_engine*);
int lio_engine_is_degenerate(lio_engine*);
/* kf.update_iterated_dyn_share_modified(LASER_POINT_COV) on the scan's current ds points; returns #passes */
int lio_engine_update(lio_engine*);
typedef struct lio_pass_log {
    int32_t knn, n_eff, valid, degenerate;
    double sum_abs_res;
    double JtJ[36]; /* after the degeneracy projection, as consumed by the filter */
    double Jtr[6];
    double dx[23];
} lio_pass_log;
int lio_engine_pass_log(lio_engine*, int i, lio_pass_log* out);
/* the per-scan body of fastlio_main: returns 0 first-scan latch, 1 map seeded, 2 too few points,
 * 3 state updated + map_incremental ENQUEUED, < 0 error.  The insert chain (map_incremental's AddPoints + the LRU list) runs on the map's
 * own stream and is not waited for: the call returns when the state is final, the next scan's upload / motion compensation / downsample
 * run beside it, and whatever reads the map next (the next neighbour search, any lio_map_* call) is ordered behind it.  A map that
 * overflowed (LIO_E_CAPACITY) is therefore reported later, ONCE, by whichever call looks first: the next lio_engine_process_scan /
 * lio_fastlio_main (before it launches anything of its own scan; lio_last_error says the failure is the previous scan's),
 * lio_engine_timings, or lio_engine_flush -- call lio_engine_flush after the last scan of a run, or the last insert's outcome is never seen.
 * With lio_engine_enable_timing(1) the insert is waited for (its time is one of the stage timings). */
int lio_engine_process_scan(lio_engine*, const float* raw_body_xyzi, uint32_t n_raw, double lidar_beg_time);
int lio_engine_process_scan_device(lio_engine*, const void* d_raw_body_xyzi, uint32_t n_raw, double lidar_beg_time);
/* per-stage device time of the last process_scan in microseconds (hipEvent based; the reference's
 * equivalent timers are commented out at src/laserMapping.cpp:1314-1342) */
typedef struct lio_timings {
    float downsample_us, knn_us, linearize_us, insert_us, total_device_us;
    float host_solve_us, total_wall_us;
    int32_t n_knn_pass, n_pass, n_ds, n_eff_last;
    int32_t n_added; /* points map_incremental added; -1 while the scan's insert is still running (see above; lio_engine_flush waits for it) */
    uint64_t knn_candidates; /* in-stencil points visited over all kNN passes (C-bar * N_ds * n_knn) */
    float undistort_us;      /* lio_fastlio_main only: pose upload + point filter + motion compensation kernels */
    float imu_host_us;       /* lio_fastlio_main only: host forward propagation (esekf::predict per IMU sample) */
} lio_timings;
int lio_engine_timings(lio_engine*, lio_timings* out);
int lio_engine_enable_timing(lio_engine*, int on);
/* waits for the map_incremental the last lio_engine_process_scan / lio_fastlio_main enqueued and returns its outcome (LIO_OK, LIO_E_CAPACITY
 * when the map overflowed, LIO_E_DEVICE); nothing pending: LIO_OK.  The reference inserts synchronously inside fastlio_main
 * (laserMapping.cpp:1304): this is where the deferred half of that call ends. */
int lio_engine_flush(lio_engine*);
/* ---------------------------------------------------------------------------------------------------------------
 * The IMU front half: the reference's FastLIO entry points, one to one.  After lio_fastlio_init the engine is driven
 * exactly like the reference's module: sensor threads enqueue, the LIO thread calls lio_fastlio_main in a loop
 * (slam/mapping/fastlio/src/fastlio.cpp:185-210,262-276).
 *   lio_fastlio_init ............ fastlio_init          src/laserMapping.cpp:1025-1124 (extR row-major 3x3); also turns on the
 *                                 map's LRU list with the reference's 100000 voxels / 100 m when the engine owns an empty
 *                                 map created with max_voxels > 100000
 *   lio_fastlio_is_init ......... fastlio_is_init       src/laserMapping.cpp:740-743
 *   lio_fastlio_imu_enqueue ..... fastlio_imu_enqueue   src/laserMapping.cpp:397-415 (acc in m/s^2, divided by 9.81 inside)
 *   lio_fastlio_ins_enqueue ..... fastlio_ins_enqueue   src/laserMapping.cpp:417-441
 *   lio_fastlio_pcl_enqueue ..... fastlio_pcl_enqueue   src/laserMapping.cpp:311-330 + Preprocess::velodyne_handler
 *                                 src/preprocess.cpp:395-451: xyzi float4 + PointAttr::stamp (us relative to the header
 *                                 stamp, common/mapping_types.h:26-29); header_stamp in seconds
 *   lio_fastlio_main ............ fastlio_main          src/laserMapping.cpp:1126-1310: sync_packages (:445-520),
 *                                 ImuProcess::Process (src/IMU_Processing.hpp:408-450: IMU_init :164-236, UndistortPcl
 *                                 :238-406 with esekf::predict esekfom.hpp:279-383), then the scan-matching path
 *   lio_fastlio_odometry ........ fastlio_odometry      src/laserMapping.cpp:692-712 (two row-major 4x4)
 *   lio_fastlio_state ........... fastlio_state         src/laserMapping.cpp:714-738 (20 doubles)
 * lio_fastlio_main returns one of LIO_MAIN_* (the reference returns false for IDLE and true otherwise) or a negative
 * error.  The enqueue calls may come from other threads than lio_fastlio_main (the reference's mtx_buffer). */
#define LIO_MAIN_FIRST_SCAN 0 /* first scan only latches first_lidar_time */
#define LIO_MAIN_SEEDED 1     /* map was empty: seeded with this scan */
#define LIO_MAIN_SKIPPED 2    /* too few points */
#define LIO_MAIN_UPDATED 3    /* state updated, map extended */
#define LIO_MAIN_IMU_INIT 4   /* IMU initialisation still collecting samples (no cloud registered) */
#define LIO_MAIN_IDLE 5       /* sync_packages found nothing to do */
int lio_fastlio_init(lio_engine*, const double extT[3], const double extR[9], int filter_num, int max_point_num, double scan_period,
                     int undistort);
int lio_fastlio_is_init(lio_engine*);
int lio_fastlio_imu_enqueue(lio_engine*, double stamp, const double gyr[3], const double acc_ms2[3]);
/* Where the iterate loop of lio_engine_update / lio_engine_process_scan / lio_fastlio_main runs.  0 (default): on the host -- one
 * device linearisation and one hand-over per pass (esekfom.hpp:1619-1931 driven from the CPU, 8 us of host algebra per pass).  1: on the
 * device (the batched engine's loop for this one scan: one submission, one wait; the calling thread is free meanwhile, the scan takes
 * ~15 % longer).  Results identical (tests/test_gpu_parity.py).  Environment LIO_DEVICE_LOOP=1 sets the default of new engines. */
int lio_engine_set_device_loop(lio_engine*, int on);

/* fastlio_ins_enqueue (src/laserMapping.cpp:417-441) after the ENU -> ego -> IMU rotation the reference applies there:
 * vel_imu = Lidar_R_wrt_IMU * Tve^-1 * (Ve, Vn, Vu), third component zeroed by the caller as :436 does.  Only IMU
 * initialisation reads it (IMU_Processing.hpp:201-204; the wheel-speed rows are compiled out, wheelspeed_en == false) */
int lio_fastlio_ins_enqueue(lio_engine*, double stamp, const double vel_imu[3]);
/* wheelspeed_en (src/laserMapping.cpp:83, a constant false in the reference: its binaries never take the branch).  When enabled, a scan
 * whose last INS sample (lio_fastlio_ins_enqueue) is within 10 ms of the scan's end gets the three velocity rows of
 * h_share_model_wheelspeed (:794-811) appended to its point-to-plane rows in every pass of the update (:994-1012: weight 1e-4, 1e-3 when
 * degenerate, times the rows before them).  Such updates run the iterate loop from the host. */
int lio_fastlio_set_wheelspeed(lio_engine*, int enable);
int lio_fastlio_pcl_enqueue(lio_engine*, const float* xyzi, const uint32_t* stamp_us, uint32_t n, double header_stamp);
/* the same without an intermediate copy (boundary marshalling, numpy_to_pointcloud + preprocessPoints of the reference: slam/src/py_utils.cpp:
 * 149-181, slam/common/slam_base.h:83-85): _stage hands out the pinned staging buffers of the next scan (room for n points), the caller writes
 * the points (already in the INS frame) and stamps straight into them, _commit starts their trip to the device and queues the scan.  One scan
 * staged at a time; the pointers are dead after _commit. */
int lio_fastlio_pcl_stage(lio_engine*, uint32_t n, float** xyzi, uint32_t** stamp_us);
int lio_fastlio_pcl_commit(lio_engine*, uint32_t n, double header_stamp);
/* device-resident scan: the two buffers must stay valid until the lio_fastlio_main call that consumes them returned */
int lio_fastlio_pcl_enqueue_device(lio_engine*, const void* d_xyzi, const void* d_stamp_us, uint32_t n, double header_stamp);
int lio_fastlio_main(lio_engine*);
int lio_fastlio_odometry(lio_engine*, double odom_s[16], double odom_e[16]);
int lio_fastlio_state(lio_engine*, double out[20]);
/* test visibility: p_imu->start_state_point as a 26-double state, feats_undistort (dropped points are NaN), and one
 * esekf::predict step on the host (esekfom.hpp:279-383; Q = diagonal of the 12 x 12 process noise: ng, na, nbg, nba;
 * acc in m/s^2); and one update_iterated_dyn_share_modified (esekfom.hpp:1619-1931) on the host with a caller-supplied
 * measurement model fn(ctx, state26, converge, &n, rows6 n x 6, h n, cap) -> valid, for pinning the filter algebra */
int lio_fastlio_start_state(lio_engine*, double s26[26]);
int lio_fastlio_download_undistorted(lio_engine*, float* out_xyzi, uint32_t cap, uint32_t* n);
typedef int (*lio_meas_fn)(void* ctx, const double* s26, int converge, int* n, double* rows6, double* h, int cap);
int lio_eskf_update_cb(const double s26[26], const double P[529], double R, int max_iter, lio_meas_fn fn, void* ctx, int cap, double s26_out[26],
                       double P_out[529]);
/* ... with the wheel-speed rows (src/laserMapping.cpp:794-811, 994-1012) appended to the model's rows in every pass: ins_vel = the INS velocity in
 * the IMU frame (NULL: none), degenerate = the is_degenerate flag the weight depends on */
int lio_eskf_update_ws_cb(const double s26[26], const double P[529], double R, int max_iter, lio_meas_fn fn, void* ctx, int cap, const double* ins_vel,
                          int degenerate, double s26_out[26], double P_out[529]);
int lio_state_predict(const double s26[26], const double P[529], double dt, const double Q[12], const double acc[3], const double gyro[3],
                      double s26_out[26], double P_out[529]);
/* test visibility, host-only: the DEVICE-resident form of update_iterated_dyn_share_modified (csrc/eskf_dev.h: the filter pass as
 * data-parallel phases, one workgroup per scan on the GPU; here the same source compiled for the host) driven by caller-supplied sums:
 *   fn(ctx, state26, converge, acc29) -> valid: acc29 = the 21 upper-triangle entries of J^T J row by row, 6 of J^T h, sum |r|, N_eff
 *   dfn(ctx, V, cs): the six degeneracy sums (contri[3], strong[3]) against the eigenvectors V (columns of a row-major 3 x 3); called
 *                    only when the eigenvalue bound does not decide
 * Returns the number of pass logs written (<= cap_logs); *status = 1 finished, 2 a pass saw N_eff < 23 (the dense branch of
 * esekfom.hpp:1715-1744 runs on the host: the state is the one BEFORE that pass). */
typedef int (*lio_sums_fn)(void* ctx, const double* s26, int converge, double* acc29);
typedef void (*lio_degeneracy_fn)(void* ctx, const double* V9, double* cs6);
int lio_eskf_update_sums_cb(const double s26[26], const double P[529], double R, int max_iter, int degenerate_detect_en, lio_sums_fn fn,
                            lio_degeneracy_fn dfn, void* ctx, double s26_out[26], double P_out[529], lio_pass_log* logs, int cap_logs, int* status);

/* Joint registration across GPUs (BASELINE.json config 5: sub-maps one per GPU, all-gather of the per-shard
 * J^T J / J^T r sums).  When a hook is set the engine calls it after every device linearisation with its LOCAL sums and
 * continues with whatever the hook leaves in the buffer (the GLOBAL sums, identical on every rank):
 *   first call,  n = 29: buf[0..20] J^T J upper triangle (row-major), buf[21..26] J^T r, buf[27] sum|r|, buf[28] n_eff
 *   second call, n = 6 (only when the degeneracy bound on the GLOBAL eigenvalues does not decide): contri[3], strong[3]
 * Every rank then runs the same 23-DoF update on the same numbers.  With a hook the N_eff < 23 branch of the filter
 * uses the information form (rows live on different GPUs). */
typedef void (*lio_reduce_fn)(void* ctx, double* buf, int n);
int lio_engine_set_reduce_hook(lio_engine*, lio_reduce_fn fn, void* ctx);
/* The same natively, over RCCL (librccl = the ROCm build of NCCL; xGMI between the GPUs of a node), for hosts that are not Python: one process
 * per GPU.  lio_comm_unique_id: rank 0 makes the 128-byte id and ships it to the other ranks by whatever the application has (MPI, a socket,
 * torch.distributed's store); lio_comm_init: ncclCommInitRank on `device` (a world of one needs no id and no RCCL).
 * lio_allgather_normal_eq: DEVICE buffers in and out -- every rank contributes one 32-double record (21 J^T J upper triangle, 6 J^T r, sum |r|,
 * N_eff, 3 spare), receives all of them rank-major (world x 32) and, if d_sum32 is given, their sum in fixed rank order (a one-wave kernel:
 * every rank forms the identical bits).  `stream` (a hipStream_t, NULL = the communicator's own) orders it with the caller's kernels.
 * lio_engine_set_joint: joint registration of ONE scan against sub-maps spread over engines and GPUs -- `e` drives the filter; after every
 * linearisation its sums are joined, in this order, by those of the `others` (further sub-maps resident on this GPU, each engine with its own
 * map) and then by the other ranks' through `comm` (NULL = single process).  Replaces a hook set by lio_engine_set_reduce_hook.
 * lio_engine_joint_register: uploads the cloud to every local engine, runs the registration; prior in, posterior out. */
typedef struct lio_comm lio_comm;
int lio_comm_unique_id(uint8_t id[128]);
lio_comm* lio_comm_init(int device, int rank, int world, const uint8_t id[128]);
void lio_comm_destroy(lio_comm*);
int lio_comm_rank(const lio_comm*);
int lio_comm_world(const lio_comm*);
int lio_allgather_normal_eq(lio_comm*, const double* d_local32, double* d_gathered, double* d_sum32, void* stream);
/* the same for the records of many scans at once (the batched engine's joint mode: one collective per pass for all scans of a round):
 * n_records x 32 doubles per rank in, world x n_records x 32 out (rank-major), no sum -- the consumer adds the ranks' records in rank order */
int lio_allgather_records(lio_comm*, const double* d_local, double* d_gathered, uint32_t n_records, void* stream);
int lio_comm_stats(lio_comm*, uint64_t* n_collectives, double* total_us);  /* collectives issued by joint registrations and their host-observed time */
int lio_engine_set_joint(lio_engine* e, lio_engine** others, int n_others, lio_comm* comm);
int lio_engine_joint_register(lio_engine* e, const float* raw_body_xyzi, uint32_t n_raw, double lidar_beg_time, double state26[26], double cov[529]);
int lio_engine_joint_register_device(lio_engine* e, const void* d_raw_body_xyzi, uint32_t n_raw, double lidar_beg_time, double state26[26], double cov[529]);
/* Throughput mode: register a batch of independent scans with `n_engines` engines running concurrently (one host
 * thread + HIP stream per engine, jobs handed out through an atomic counter).  Every job = set_state(state_in) +
 * set_cov(cov_in) + lio_engine_process_scan_device(d_raw, n_raw, lidar_beg_time); outputs are filled per job.
 * Intended for engines created with lio_engine_create_shared on one read-only map (independent scans of several
 * sensors / sequences, relocalisation candidates, map-merge alignments). */
#define LIO_JOB_KEEP_CACHE 1u
#define LIO_JOB_IDLE 2u         /* lio_batch_sequences_step: the session of this job has no scan this round (rc = 0, nothing else is looked at) */
#define LIO_JOB_HOST_RAW 4u     /* lio_batch_process / lio_engines_process_batch: d_raw points to HOST memory (pinned for full PCIe rate: lio_pinned_alloc);
                                   the library copies the cloud to HBM on the round's stream, overlapped with the other rounds in flight -- the copy
                                   the reference's boundary starts with (slam/src/py_utils.cpp:149-181, slam_wrapper.cpp:64-84).  The host buffer must
                                   stay valid and unchanged until the call that took the job returns (the copy is asynchronous).  Not accepted by
                                   lio_batch_sequences_step (rc = LIO_E_INVALID).  Appended in round 5 */
#define LIO_JOB_FLAGS_KNOWN 7u  /* every other bit must be zero: a job with unknown bits is rejected (rc = LIO_E_INVALID), so that an uninitialised
                                   word cannot silently pick a behaviour */
typedef struct lio_scan_job {
    const void* d_raw;          /* device pointer, XYZI float4 */
    uint32_t n_raw;
    uint32_t flags;             /* ABI note: until round 3 this word was padding and ignored.  ZERO-INITIALISE the job array (memset / = {0}).
                                   0 (default): the jobs are INDEPENDENT scans -- the engine / slot that takes the job first forgets its neighbour
                                   cache, as fastlio_init does for Nearest_Points (src/laserMapping.cpp:1045-1047), so the result does not depend
                                   on which scan that engine registered before; LIO_JOB_KEEP_CACHE: a job of a sequence -- the cache of the
                                   previous scan survives (Nearest_Points across fastlio_main calls, stale where a search finds nothing): callers that
                                   feed CONSECUTIVE scans of one sensor through a batch API and want fastlio_main's carry-over must set it */
    double lidar_beg_time;
    const double* state_in;     /* 26 doubles */
    const double* cov_in;       /* 529 doubles */
    double* state_out;          /* 26 doubles, may be NULL */
    int32_t rc;                 /* return code of process_scan */
    int32_t n_ds, n_pass, n_knn_pass;
} lio_scan_job;
int lio_engines_process_batch(lio_engine** engines, int n_engines, lio_scan_job* jobs, int n_jobs);
/* Throughput mode, batched: B scans per launch.  A batch owns `n_groups` groups of `n_slots` slots (scan buffers + a device-resident
 * filter each); lio_batch_process hands the jobs to the groups round-robin from ONE host thread: a round = one small upload (states,
 * covariances, descriptors), the voxel-grid chain and (maximum_iter + 1) x {stencil kNN where the filter asks for it, linearisation,
 * filter pass} enqueued blind on the group's stream -- every launch serves all slots (blockIdx.y = slot), the iterate loop of
 * update_iterated_dyn_share_modified (esekfom.hpp:1619-1931) runs on the device (csrc/eskf_dev.h), and the host sees one result record per
 * scan in mapped memory.  Same job semantics and results as lio_engines_process_batch on engines created with lio_engine_create_shared
 * (static map: map_incremental is skipped); a pass that needs the N_eff < 23 dense branch of the filter is finished on the host. */
typedef struct lio_batch lio_batch;
typedef struct lio_batch_result {
    double state[26];
    int32_t status;        /* 1 finished on the device, 2 a pass saw 1 <= N_eff < 23: state / loop_* are those BEFORE that pass */
    int32_t n_pass, n_knn_pass, n_ds, n_eff, degenerate, radix_passes, err;
    int32_t loop_i, loop_t, loop_converge, pad;
    uint32_t seq, pad2;
} lio_batch_result;
lio_batch* lio_batch_create(lio_map* shared_map, int n_slots, int n_groups, uint32_t max_raw, uint32_t max_ds);
/* Joint registration, batched (BASELINE.json config 5, the map-merge shape of slam/localization/include/overlap_merge.hpp:46-48,158-179: many key
 * frames x candidates, each an independent alignment): every job of lio_batch_process is registered against ALL `sub_maps` (resident on this GPU)
 * and, through `comm` (NULL: single process), against the sub-maps of the other ranks -- every rank calls lio_batch_process with the SAME job list
 * (same clouds resident on its own GPU, same priors) and receives the same posteriors, bit for bit.  A round of B scans is enqueued blind on the
 * round's stream: downsample once per scan, then per pass {neighbour search, linearisation} per local sub-map, the rank's B records of 32 doubles,
 * ONE all-gather of [B x 32] doubles for the whole round (RCCL on that stream, no host in between), the filter pass on the sums taken in rank
 * order (csrc/eskf_dev.h on the device).  A pass with N_eff < 23 uses the information form (the rows live on several GPUs), a scan whose pass needs
 * the degeneracy sums (src/laserMapping.cpp:946-964) is redone by the host-driven path (lio_engine_joint_register_device of the slot's engines).
 * Results equal lio_engine_joint_register's on the same sub-maps, job by job (tests/test_dist.py). */
lio_batch* lio_batch_create_joint(lio_map** sub_maps, int n_sub_maps, lio_comm* comm, int n_slots, int n_groups, uint32_t max_raw, uint32_t max_ds);
/* the collective of the joint mode through a caller-supplied function instead of RCCL (embedding in a host that has its own transport; and the
 * way the world > 1 logic is exercised where RCCL cannot be -- two ranks on ONE GPU over gloo, tests/test_dist.py): called on the submitting
 * thread once per round and pass with this rank's [n_records x 32] doubles in device memory, it must leave every rank's records, rank-major
 * ([rank][record][32]), in d_gathered before it returns (work already enqueued on `stream` precedes the call; what follows is enqueued after).
 * For a batch created by lio_batch_create_joint WITHOUT a communicator; a scan that needs the host-driven path (degeneracy sums) fails with
 * LIO_E_STATE in this mode. */
typedef int (*lio_gather_fn)(void* ctx, const double* d_local, double* d_gathered, uint32_t n_records, void* stream);
int lio_batch_set_gather_hook(lio_batch*, lio_gather_fn fn, void* ctx, int rank, int world);
/* With more than one rank a joint round's voxel-grid chain (src/laserMapping.cpp:1206-1208, once per scan) runs on ONE rank per scan -- rank r owns the
 * slots [r * ceil(B / world), ...) -- and the downsampled clouds reach the others in one all-gather per round of fixed-size slot chunks (the same
 * transport as the records; with a hook, n_records = chunk bytes / 256).  The chunk holds 1.25 x the largest cloud the batch has registered so far
 * (derived from the results, hence equal on every rank); a cloud that does not fit voids the round's result for that job on every rank and the job runs
 * again with full-size chunks.  LIO_JOINT_SPLIT_DS=0 in the environment: every rank downsamples every scan, as up to ABI revision 5 (bit-identical
 * results either way: tests/test_dist.py).  Statistics: points per chunk now in force (0 before the first round), jobs that ran again, bytes one rank
 * contributes to a round's all-gather. */
int lio_batch_exchange_stats(lio_batch*, uint32_t* chunk_points, uint64_t* jobs_rerun, uint64_t* bytes_per_rank_and_round);
/* Sequence mode: throughput WITH map_incremental.  n_groups x n_slots independent SLAM sessions (replay of recorded drives, offline mapping of many
 * sequences), one per slot, each with ITS OWN map (lio_engine_create's arguments, per session).  lio_batch_sequences_step takes exactly one job per
 * session -- job j is the NEXT scan of session j = (group j / n_slots, slot j % n_slots): state_in / cov_in the propagated prior as for every
 * lio_scan_job, LIO_JOB_IDLE for a session without a scan this round -- and runs, per group, fastlio_main from the downsample to map_incremental
 * (src/laserMapping.cpp:1193-1304) as ONE blind submission: voxel-grid chain, (maximum_iter + 1) x {stencil kNN against the slot's own map,
 * linearisation, filter pass on the device}, then for every slot whose update finished classify + IVox::AddPoints (+ the LRU list) -- every launch
 * serves all slots (blockIdx.y = slot).  SURVEY 8d's B_ins, which the static batch leaves out, is inside the round.  The file-scope state of
 * laserMapping.cpp (first_lidar_time, flg_EKF_inited, travel, the stencil switch at 10 x INIT_TIME) lives in the slot's engine
 * (lio_batch_engine(b, group, slot); its map: lio_engine_map).  Scans a round cannot take go through that engine's own code -- a session's first
 * scan (only its time is kept, rc 0) and its map seed (rc 1), a scan whose pass needs the dense N_eff < 23 branch, a bounding box that needs more
 * radix passes than were launched -- so a session driven here and the same scans pushed one by one through lio_engine_set_state / set_cov /
 * lio_engine_process_scan_device on an engine with lio_engine_set_device_loop(e, 1) give the same bits (tests/test_sequence_batch_gpu.py).
 * rc per job as lio_engine_process_scan; state_out and cov_out[j * 529 ..] (may be NULL) receive the posterior (the prior where nothing was
 * registered).  The pass logs of a scan registered inside a round stay on the device (lio_engine_pass_log of the slot's engine is empty).
 * A step that returns a device error after some of its groups' rounds already ran leaves those sessions' maps one sweep ahead of their engines: the
 * batch then refuses further steps (LIO_E_STATE) -- destroy it; a step that fails before any round ran can be repeated. */
lio_batch* lio_batch_create_sequences(int device, float resolution, int stencil, uint64_t max_points, uint64_t max_voxels, int n_slots, int n_groups,
                                      uint32_t max_raw, uint32_t max_ds);
int lio_batch_sequences_step(lio_batch*, lio_scan_job* jobs, int n_jobs, double* cov_out);
/* lio_fastlio_main (= fastlio_main, src/laserMapping.cpp:1160-1310) for every session of a sequence batch at once: replay of many recorded drives
 * through the reference's own entry points.  Each session's engine carries the front half -- lio_fastlio_init(lio_batch_engine(b, g, s), ...), then
 * lio_fastlio_imu_enqueue / lio_fastlio_ins_enqueue / lio_fastlio_pcl_enqueue[_device] on it as for a single engine.  One call = one fastlio_main per
 * session: sync_packages, IMU initialisation, forward propagation and undistortion per session as lio_fastlio_main does them, ONE sequence round for
 * the scans that reach registration, the back half per session.  rc_out[j] (n_groups x n_slots entries) = what lio_fastlio_main would have returned
 * for session j (LIO_MAIN_IDLE without a complete package); lio_fastlio_odometry / _state of the session's engine read the result.  Same bits as
 * lio_fastlio_main on a per-session engine with the device loop on (tests/test_sequence_batch_gpu.py). */
int lio_batch_fastlio_main(lio_batch*, int* rc_out);
void lio_batch_destroy(lio_batch*);
int lio_batch_process(lio_batch*, lio_scan_job* jobs, int n_jobs);
/* live kernel timing of the batched chain with HIP events on the groups' streams (bench.py's roofline leg): per class the summed device
 * time and the number of timed launches (a "downsample" launch = the whole voxel-grid chain of one round; a kNN / linearise / filter-pass
 * launch = one kernel serving all slots of a round).  Off by default. */
typedef struct lio_batch_times {
    double downsample_us, knn_us, linearize_us, step_us;
    uint32_t downsample_launches, knn_launches, linearize_launches, step_launches;
    double insert_us;            /* sequence mode: the map_incremental half of a round (classify + AddPoints + LRU + read-back records); appended in round 4 */
    uint32_t insert_launches, pad;
} lio_batch_times;
/* on: 0 off, 1 timing, 2 (or 3) timing with the counting variant of the kNN kernel (lio_map_knn_touched) -- times of that variant are not
 * the product's */
int lio_batch_enable_kernel_timing(lio_batch*, int on);
int lio_batch_kernel_times(lio_batch*, lio_batch_times* out, int reset);
/* test visibility: the engine behind slot `slot` of group `group` (its scan buffers, pass log of a host continuation) */
lio_engine* lio_batch_engine(lio_batch*, int group, int slot);
/* on: process_scan skips map_incremental -- scan-to-map registration against a prebuilt static map
 * (BASELINE.json configs 2 and 4); off (default): the reference's mapping behaviour */
int lio_engine_set_static_map(lio_engine*, int on);

/* ---------------------------------------------------------------------------------------------
 * Localization matcher: replaces fast_gicp::NDTCuda<PointXYZI, PointXYZI> as select_registration_method("NDT_CUDA")
 * configures it (slam/backend/hdl_graph_slam/src/hdl_graph_slam/registrations.cpp:105-118: P2D, resolution 1.0,
 * DIRECT7) behind pcl::Registration's setInputTarget / setInputSource / align
 * (slam/localization/hdl_localization/src/hdl_localization/pose_estimator.cpp:246-247).
 * The source cloud is a lio_scan (upload + lio_scan_voxel_downsample with leaf = map_resolution, as
 * hdl_localization_nodelet.cpp:333-344 does); transforms are row-major 4x4 doubles.
 * ------------------------------------------------------------------------------------------- */
typedef struct lio_ndt lio_ndt;
/* NDTCuda() + setResolution + setNeighborSearchMethod(DIRECT1 / DIRECT7 / DIRECT27  ->  search_method 1 / 7 / 27) */
lio_ndt* lio_ndt_create(int device, float resolution, int search_method, uint64_t max_points, uint64_t max_voxels, uint32_t max_source_points);
void lio_ndt_destroy(lio_ndt*);
/* setInputTarget -> NDTCudaCore::set_target_cloud -> create_target_voxelmap (ndt_cuda.cu:105-141): Gaussian voxel map
 * (mean, covariance), PLANE regularisation, inverse covariance.  Replaces any previous target. */
int lio_ndt_set_target(lio_ndt*, const float* xyzi, uint64_t n);
int lio_ndt_set_target_device(lio_ndt*, const void* d_xyzi, uint64_t n);
int lio_ndt_num_voxels(lio_ndt*);
/* diagnostic: statistics of the voxel that contains p; returns its point count (0 = no such voxel) */
int lio_ndt_voxel_at(lio_ndt*, const float p[3], float mean[3], float cinv[9]);
/* NDTCuda::linearize (update_corr = 1, with_derivatives = 1: update_correspondences + compute_error with H, b) and
 * NDTCuda::compute_error (update_corr = 0, with_derivatives = 0: error on the cached pairs)  (ndt_cuda_impl.hpp:82-90) */
int lio_ndt_linearize(lio_ndt*, lio_scan* source, const double T[16], int update_corr, int with_derivatives, double H[36], double b[6],
                      double* err, uint32_t* n_corr);
/* pcl::Registration::getFitnessScore(max_range) as pose_estimator.cpp:262 calls it (max_range 25 = a SQUARED distance): mean
 * squared distance from the transformed source points to their nearest target point, over those within range; *score is
 * DBL_MAX when none is (PCL's value).  Exact nearest neighbours, from the target points the voxel grid retains. */
int lio_ndt_fitness_score(lio_ndt*, lio_scan* source, const double T[16], double max_range, double* score, uint32_t* n_inliers);
/* calc_fitness_score(cloud1, cloud2, relpose, max_range) of the map-merge / loop-closure tools, slam/localization/include/overlap_merge.hpp:
 * 206-263: the target is cloud1 after `filter` (:196-204: sqrt(x^2 + y^2) < xy_range && z > min_z -- applied by the caller before
 * lio_ndt_set_target), the source is cloud2 (all its points: lio_scan_set_ds), transformed by relpose (pcl::transformPointCloud, f32) and
 * then put through the same filter here.  *score = mean squared nearest-neighbour distance over the source points whose neighbour is
 * within max_range (squared, as PCL's kd-tree returns it), *inlier_ratio = their share of the filtered source; (DBL_MAX, 0) if none.
 * The reference's constants: xy_range 100.0, min_z 0.5. */
int lio_ndt_overlap_score(lio_ndt*, lio_scan* source, const double relpose[16], double max_range, double xy_range, double min_z, double* score,
                          double* inlier_ratio);
/* live timing of the matcher's dominant kernel (ndt_cost_kernel: correspondences + cost [+ H, b] of one evaluation) with HIP events on the
 * stream it runs on -- bench.py's roofline leg of BASELINE config 4.  Summed over the evaluations since the last reset: device time, launches
 * (of which with a correspondence update), voxel correspondences evaluated, source points.  Off by default (one event wait per evaluation). */
typedef struct lio_ndt_times {
    double cost_us;
    uint64_t launches, update_launches, pairs, source_points;
} lio_ndt_times;
int lio_ndt_enable_kernel_timing(lio_ndt*, int on);
int lio_ndt_kernel_times(lio_ndt*, lio_ndt_times* out, int reset);
typedef struct lio_ndt_params {
    int32_t max_iterations;           /* setMaximumIterations (64) */
    int32_t lm_max_iterations;        /* lm_max_iterations_ (10) */
    double rotation_epsilon_deg;      /* setRotationEpsilon (0.1) */
    double transformation_epsilon;    /* setTransformationEpsilon (0.01) */
    double lm_init_lambda_factor;     /* 1e-9 */
    double max_process_time_ms;       /* setMaxProcessTime; <= 0: no wall-clock cut-off */
} lio_ndt_params;
void lio_ndt_default_params(lio_ndt_params*);
/* pcl::Registration::align(guess) -> LsqRegistration::computeTransformation with step_lm
 * (lsq_registration_impl.hpp:71-109,163-208); out = final transformation, *converged = hasConverged().  step_lm's compute_error(xi) and the
 * linearize(xi) that follows an accepted step are fetched in one launch (see lio_ndt_align_batch): identical results, one device hand-over per
 * LM iteration instead of two. */
int lio_ndt_align(lio_ndt*, lio_scan* source, const double guess[16], const lio_ndt_params* params, double out[16], int* iterations,
                  int* converged);

/* Batched alignments: the map-merge / loop-closure / relocalisation tools evaluate several candidates per key frame (slam/localization/include/
 * overlap_merge.hpp:158-179: 64 key frames x <= 3 candidates, every one an independent registration->align) -- here B of them per launch against
 * one or several targets: slot = (target, source scan, guess), the Levenberg-Marquardt loop of LsqRegistration (lsq_registration_impl.hpp:71-208) resident on the
 * device, a round = {cost evaluation of the slots that linearise, of the slots that try a step, LM kernel}, every launch serving all slots; the
 * host looks at the slots' states every six rounds.  A trial evaluation is speculative (round 4): one launch computes the trial's cost on the
 * pairs cached at the linearisation point AND the linearisation at the trial pose, which an accepted step continues from -- one round per LM
 * iteration instead of two, the same numbers (lio_ndt_align does the same, LIO_NDT_SPEC=0 turns it off there).  Same schedule, stopping rules
 * and results as lio_ndt_align job by job (to the rounding of the device's libm); max_process_time_ms does not apply.  `evaluations` counts
 * launches that served the job (1 + the LM trials).  Sources: lio_scan objects holding their downsampled clouds (lio_scan_voxel_downsample /
 * lio_scan_set_ds), at most max_source_points each. */
typedef struct lio_align_job {
    lio_ndt* target;         /* NULL: the matcher the call is made on; else another target of the same resolution / search method / device
                                (a new key frame against each of its candidate frames: setInputTarget per candidate in the reference) */
    lio_scan* source;
    const double* guess;     /* row-major 4 x 4 */
    double out[16];          /* final transformation */
    int32_t iterations, converged, evaluations, rc;
} lio_align_job;
int lio_ndt_align_batch(lio_ndt*, lio_align_job* jobs, int n_jobs, const lio_ndt_params* params);

/* ---------------------------------------------------------------------------------------------
 * Generalized-ICP: replaces fast_gicp::FastGICP<PointXYZI, PointXYZI> as select_registration_method("FAST_GICP") configures it
 * (slam/backend/hdl_graph_slam/src/hdl_graph_slam/registrations.cpp:33-42) -- the fine matcher of the map merge / loop closure tools
 * (slam/localization/include/overlap_merge.hpp:54-58,158-179: coarse NDT -> fine GICP -> fitness) and, through the same cost function, the
 * matcher family of slam/thirdparty/fast_gicp/include/fast_gicp/gicp/impl/{fast_gicp_impl.hpp:118-303, fast_vgicp_impl.hpp:72-204}:
 *   create ......... FastGICP() + setCorrespondenceRandomness(k); grid_resolution = cell size of the search grid (results do not depend on it)
 *   set_target / set_source  setInputTarget / setInputSource: exact k nearest neighbours of every point, covariance, PLANE regularisation
 *   linearize ...... update_correspondences (nearest target point within max_corr_dist, Mahalanobis (C_B + R C_A R^T)^-1) + linearize:
 *                    H = sum J^T M J, b = sum J^T M e, err = sum e^T M e (update_corr = 0: the cached pairs; with_derivatives = 0: compute_error)
 *   align .......... pcl::Registration::align(guess) -> LsqRegistration's LM loop; params NULL = the reference's FAST_GICP settings
 *   download / correspondences  test visibility: the clouds in internal (grid) order with their regularised covariances (xx, xy, xz, yy, yz, zz);
 *                    the target index of every source point (-1 = none), in that order
 * Transforms are row-major 4 x 4 doubles. */
typedef struct lio_gicp lio_gicp;
lio_gicp* lio_gicp_create(int device, float grid_resolution, uint32_t max_points, int k_correspondences);
void lio_gicp_destroy(lio_gicp*);
int lio_gicp_set_target(lio_gicp*, const float* xyzi, uint32_t n);
int lio_gicp_set_source(lio_gicp*, const float* xyzi, uint32_t n);
/* The voxelised variant, fast_gicp::FastVGICP (fast_vgicp_impl.hpp:72-204; select_registration_method("FAST_VGICP"), registrations.cpp:56-66:
 * the reference's matcher on machines without CUDA): voxel_resolution > 0 turns the target into Gaussian voxels (mean position, mean of the
 * points' regularised 20-NN covariances: ADDITIVE mode, fast_vgicp_voxel.hpp:95-110,129-167) and lio_gicp_linearize / _align into its
 * update_correspondences / linearize / compute_error: a source point corresponds to the voxel its transformed position falls in
 * (search_method 1 = DIRECT1, the reference's default; 7, 27: the neighbours too), weight sqrt(points in the voxel); max_corr_dist is not
 * used.  0 = back to the kd-tree form.  The resolution should be exactly representable in f32 (the reference's 1.0 is). */
int lio_gicp_set_voxel_mode(lio_gicp*, double voxel_resolution, int search_method);
/* diagnostic: the Gaussian voxel of the target holding point p -> number of points (0: none), mean[3], covariance (xx, xy, xz, yy, yz, zz) */
int lio_gicp_voxel_at(lio_gicp*, const float p[3], double mean[3], double cov6[6]);
int lio_gicp_download(lio_gicp*, int which, float* xyzi, double* cov6, uint32_t cap);
int lio_gicp_correspondences(lio_gicp*, int32_t* corr, uint32_t cap);
int lio_gicp_linearize(lio_gicp*, const double T[16], double max_corr_dist, int update_corr, int with_derivatives, double H[36], double b[6], double* err,
                       uint32_t* n_corr);
int lio_gicp_align(lio_gicp*, const double guess[16], const lio_ndt_params* params, double max_corr_dist, double out[16], int* iterations, int* converged);

/* ---------------------------------------------------------------------------------------------------------------
 * The localisation loop around the matcher: hdl_localization::PoseEstimator (slam/localization/hdl_localization/src/
 * pose_estimator.cpp) = a 23-state unscented Kalman filter (include/kkl/alg/unscented_kalman_filter.hpp:42-262 over
 * include/hdl_localization/pose_system.hpp:14-115; f32 like the reference), host C++:
 *   create ........ PoseEstimator::PoseEstimator      pose_estimator.cpp:22-66 (imu_ext row-major 4 x 4, quaternion (w, x, y, z))
 *   predict ....... predict(stamp) / predict(stamp, acc, gyro)   :142-186 (acc, gyro NULL = no IMU); returns 1 if a step was made
 *   match ......... match(observation, ..., stamp, cloud, gps = none, ...)  :188-300: lio_ndt_align from the filter's pose, the
 *                   5 m / 10 deg gate, quaternion hemisphere; returns 1 / 0 = the reference's bool
 *   correct ....... correct(stamp, observation)       :348-360
 *   matrix / get .. matrix(), ukf->mean / cov
 *   guess / observe  the two host halves of match() around the alignment (:196-247 / :250-302), with the GNSS observation fused in
 *                   (fusion_pose :420-433); match_gps = guess -> lio_ndt_align -> observe; match_gps_only = the scan-less match :304-346
 *   get_timed_pose  get_timed_pose + the INS state queue (:104-141), re-predicted by correct (:366-381); predict_nostate :70-86
 * The fitness score of the warm-up phase is lio_ndt_fitness_score.  predict / predict_nostate / get_timed_pose / correct hold the handle's
 * mutex, as the reference's data_mutex does (the INS callback thread against the scan thread); the other calls belong to the scan thread. */
typedef struct lio_pose_estimator lio_pose_estimator;
typedef struct lio_gps_observation {  /* what match() reads of an RTKType: T (map-frame pose, row-major), precision, dimension (2 / 3 / 6) */
    double T[16];
    double precision;
    int32_t dimension;
} lio_gps_observation;
lio_pose_estimator* lio_pose_estimator_create(const float imu_ext[16], uint64_t stamp_us, const float pos[3], const float quat_wxyz[4],
                                              double cool_time_duration);
void lio_pose_estimator_destroy(lio_pose_estimator*);
int lio_pose_estimator_predict(lio_pose_estimator*, uint64_t stamp_us, const float acc[3], const float gyro[3]);
int lio_pose_estimator_match(lio_pose_estimator*, lio_ndt* target, lio_scan* source, const lio_ndt_params* params, float observation[7],
                             int* iterations);
int lio_pose_estimator_match_gps(lio_pose_estimator*, lio_ndt* target, lio_scan* source, const lio_ndt_params* params, const lio_gps_observation* gps,
                                 float observation[7], float observation_cov[49], int* iterations);
int lio_pose_estimator_guess(lio_pose_estimator*, const lio_gps_observation* gps, float init_guess[16]);
int lio_pose_estimator_observe(lio_pose_estimator*, const float init_guess[16], const float aligned[16], int converged, const lio_gps_observation* gps,
                               float observation[7], float observation_cov[49]);
int lio_pose_estimator_match_gps_only(lio_pose_estimator*, const lio_gps_observation* gps, float observation[7], float observation_cov[49]);
int lio_pose_estimator_get_timed_pose(lio_pose_estimator*, uint64_t stamp_us, const double acc_g[3], const double gyro_dps[3], double pose[16]);
int lio_pose_estimator_predict_nostate(lio_pose_estimator*, uint64_t stamp_us, double pose[16]);
int lio_pose_estimator_correct(lio_pose_estimator*, uint64_t stamp_us, const float observation[7]);
uint64_t lio_pose_estimator_get_dt(lio_pose_estimator*);  /* get_dt() :389-391, us */
int lio_pose_estimator_get(lio_pose_estimator*, float mean23[23], float cov529[529]);
int lio_pose_estimator_set(lio_pose_estimator*, const float mean23[23], const float cov529[529]);
int lio_pose_estimator_matrix(lio_pose_estimator*, float T[16]);

/* ---------------------------------------------------------------------------------------------------------------
 * Local-map assembly for localisation on the device: Localization::runUpdateLocalMap (slam/localization/src/
 * localization.cpp:303-373).  Key-frame clouds (map frame) are stored once in HBM; lio_localmap_update does one pass of the
 * loop body -- skip if the pose moved less than update_distance (10 m) since the last update, radius search (30 m) over the
 * key-frame positions nearest first, thinning by key_frame_distance, concatenation until max_local_points (200000), VoxelGrid
 * with `leaf`, new NDT target -- with device-to-device copies only.  Returns 0 nothing to do, 1 target replaced, 2 out of map
 * (:364-367), 3 nearest key frame >= 20 m away (:352-354); the target is dropped in cases 2 and 3. */
typedef struct lio_localmap lio_localmap;
lio_localmap* lio_localmap_create(int device, uint64_t max_total_points, uint32_t max_local_points, uint32_t max_keyframe_points);
void lio_localmap_destroy(lio_localmap*);
int lio_localmap_add_keyframe(lio_localmap*, const float* world_xyzi, uint32_t n, const float position[3]);
int lio_localmap_num_keyframes(lio_localmap*);
int lio_localmap_update(lio_localmap*, lio_ndt* target, const double pose_xyz[3], double update_distance, double radius, double key_frame_distance,
                        float leaf, int* n_keyframes, uint32_t* n_points);
int lio_localmap_download(lio_localmap*, float* out_xyzi, uint32_t cap);

/* manifold helpers exposed for known-answer tests (mtk SO3/S2 boxplus/boxminus, SOn.hpp:233-245, S2.hpp:136-167) */
void lio_state_boxplus(const double s26[26], const double d23[23], double out26[26]);
void lio_state_boxminus(const double a26[26], const double b26[26], double d23[23]);

#ifdef __cplusplus
}
#endif
#endif /* LIO_HIP_H_ */
