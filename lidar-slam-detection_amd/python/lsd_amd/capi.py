"""ctypes binding of liblio_hip.so (include/lio_hip.h) -- the MI355X-native LIO scan-matching core.

There is no CPU fallback: importing this module without the built library raises, and every handle
constructor raises when no HIP device is usable.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# the application's job (the library does not touch the environment): HIP's hardware-queue count, read by the runtime at its first call.  With the
# default of 4 the rounds in flight of lio_batch share queues and mostly serialise; lio_batch_create notes a lower value in lio_last_warning
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
LIB_PATH = os.environ.get("LIO_HIP_LIB") or os.path.join(_HERE, "liblio_hip.so")  # LIO_HIP_LIB: a variant build (tools/experiments)

LIO_OK, LIO_E_INVALID, LIO_E_CAPACITY, LIO_E_DEVICE, LIO_E_STATE = 0, -1, -2, -3, -4
MAIN_FIRST_SCAN, MAIN_SEEDED, MAIN_SKIPPED, MAIN_UPDATED, MAIN_IMU_INIT, MAIN_IDLE = 0, 1, 2, 3, 4, 5  # lio_fastlio_main

# every symbol include/lio_hip.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "lio_last_warning",
    "lio_last_error", "lio_device_count", "lio_map_bytes",
    "lio_map_create", "lio_map_destroy", "lio_map_set_lru", "lio_map_lru_stats", "lio_map_lru_exact_stats", "lio_map_set_tie_mode", "lio_map_tie_stats", "lio_map_clear", "lio_map_pool_stats", "lio_map_set_stencil", "lio_map_insert", "lio_map_insert_device", "lio_map_stats",
    "lio_abi_version", "lio_pinned_alloc", "lio_pinned_free",
    "lio_map_dump", "lio_map_knn", "lio_map_knn_candidates", "lio_map_knn_touched", "lio_map_knn_unique",
    "lio_scan_create", "lio_scan_destroy", "lio_scan_reset", "lio_scan_upload", "lio_scan_set_device", "lio_scan_undistort_delta", "lio_scan_undistort_poses", "lio_scan_download_raw", "lio_scan_voxel_downsample", "lio_scan_voxel_downsample_batch", "lio_scan_set_ds",
    "lio_scan_num_ds", "lio_scan_download_ds", "lio_scan_download_world", "lio_scan_download_match",
    "lio_p2plane_linearize", "lio_scan_set_degeneracy_mode", "lio_p2plane_degeneracy", "lio_engine_set_reduce_hook", "lio_p2plane_rows", "lio_map_incremental", "lio_map_seed",
    "lio_engine_create", "lio_engine_create_shared", "lio_engine_destroy", "lio_engine_map", "lio_engine_scan", "lio_engine_set_state", "lio_engine_get_state",
    "lio_engine_set_cov", "lio_engine_get_cov", "lio_engine_set_flags", "lio_engine_travel", "lio_engine_is_degenerate",
    "lio_engine_update", "lio_engine_pass_log", "lio_engine_process_scan", "lio_engine_process_scan_device", "lio_engine_timings", "lio_engine_flush",
    "lio_engine_enable_timing", "lio_comm_unique_id", "lio_comm_init", "lio_comm_destroy", "lio_comm_rank", "lio_comm_world", "lio_allgather_normal_eq", "lio_comm_stats", "lio_engine_set_joint", "lio_engine_joint_register", "lio_engine_joint_register_device", "lio_allgather_records", "lio_engines_process_batch", "lio_batch_create", "lio_batch_create_joint", "lio_batch_create_sequences", "lio_batch_sequences_step", "lio_batch_fastlio_main", "lio_batch_set_gather_hook", "lio_batch_exchange_stats", "lio_batch_destroy", "lio_batch_process", "lio_batch_engine", "lio_batch_enable_kernel_timing", "lio_batch_kernel_times", "lio_engine_set_static_map", "lio_scan_enable_kernel_timing", "lio_scan_kernel_times",
    "lio_state_boxplus", "lio_state_boxminus",
    "lio_localmap_create", "lio_localmap_destroy", "lio_localmap_add_keyframe", "lio_localmap_num_keyframes", "lio_localmap_update",
    "lio_localmap_download",
    "lio_pose_estimator_create", "lio_pose_estimator_destroy", "lio_pose_estimator_predict", "lio_pose_estimator_match", "lio_pose_estimator_match_gps", "lio_pose_estimator_guess", "lio_pose_estimator_observe",
    "lio_pose_estimator_match_gps_only", "lio_pose_estimator_get_timed_pose", "lio_pose_estimator_predict_nostate",
    "lio_pose_estimator_correct", "lio_pose_estimator_get_dt", "lio_pose_estimator_get", "lio_pose_estimator_set", "lio_pose_estimator_matrix",
    "lio_fastlio_init", "lio_fastlio_is_init", "lio_fastlio_imu_enqueue", "lio_engine_set_device_loop", "lio_fastlio_ins_enqueue", "lio_fastlio_set_wheelspeed", "lio_fastlio_pcl_enqueue", "lio_fastlio_pcl_enqueue_device", "lio_fastlio_pcl_stage", "lio_fastlio_pcl_commit",
    "lio_fastlio_main", "lio_fastlio_odometry", "lio_fastlio_state", "lio_fastlio_start_state", "lio_fastlio_download_undistorted",
    "lio_state_predict", "lio_eskf_update_cb", "lio_eskf_update_ws_cb", "lio_eskf_update_sums_cb",
    "lio_ndt_create", "lio_ndt_destroy", "lio_ndt_set_target", "lio_ndt_set_target_device", "lio_ndt_num_voxels", "lio_ndt_fitness_score", "lio_ndt_overlap_score", "lio_ndt_voxel_at",
    "lio_ndt_linearize", "lio_ndt_default_params", "lio_ndt_align", "lio_ndt_align_batch", "lio_ndt_enable_kernel_timing", "lio_ndt_kernel_times",
    "lio_gicp_create", "lio_gicp_destroy", "lio_gicp_set_target", "lio_gicp_set_source", "lio_gicp_set_voxel_mode", "lio_gicp_voxel_at", "lio_gicp_download", "lio_gicp_correspondences", "lio_gicp_linearize", "lio_gicp_align",
]


class NormalEq(C.Structure):
    _fields_ = [("JtJ", C.c_double * 36), ("Jtr", C.c_double * 6), ("nnT", C.c_double * 9), ("eigvec", C.c_double * 9),
                ("eigval", C.c_double * 3), ("contri", C.c_double * 3), ("strong", C.c_double * 3), ("sum_abs_res", C.c_double),
                ("n_eff", C.c_uint32), ("n_ds", C.c_uint32), ("n_knn_candidates_lo", C.c_uint32), ("n_knn_candidates_hi", C.c_uint32),
                ("n_tie", C.c_uint32), ("seq", C.c_uint32)]


class BatchTimes(C.Structure):
    _fields_ = [("downsample_us", C.c_double), ("knn_us", C.c_double), ("linearize_us", C.c_double), ("step_us", C.c_double),
                ("downsample_launches", C.c_uint32), ("knn_launches", C.c_uint32), ("linearize_launches", C.c_uint32), ("step_launches", C.c_uint32),
                ("insert_us", C.c_double), ("insert_launches", C.c_uint32), ("pad", C.c_uint32)]


class PassLog(C.Structure):
    _fields_ = [("knn", C.c_int32), ("n_eff", C.c_int32), ("valid", C.c_int32), ("degenerate", C.c_int32), ("sum_abs_res", C.c_double),
                ("JtJ", C.c_double * 36), ("Jtr", C.c_double * 6), ("dx", C.c_double * 23)]


SUMS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double))
DEG_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))
MEAS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)


class Timings(C.Structure):
    _fields_ = [("downsample_us", C.c_float), ("knn_us", C.c_float), ("linearize_us", C.c_float), ("insert_us", C.c_float),
                ("total_device_us", C.c_float), ("host_solve_us", C.c_float), ("total_wall_us", C.c_float), ("n_knn_pass", C.c_int32),
                ("n_pass", C.c_int32), ("n_ds", C.c_int32), ("n_eff_last", C.c_int32), ("n_added", C.c_int32),
                ("knn_candidates", C.c_uint64), ("undistort_us", C.c_float), ("imu_host_us", C.c_float)]


class ScanJob(C.Structure):
    _fields_ = [("d_raw", C.c_void_p), ("n_raw", C.c_uint32), ("flags", C.c_uint32), ("lidar_beg_time", C.c_double),
                ("state_in", C.POINTER(C.c_double)), ("cov_in", C.POINTER(C.c_double)), ("state_out", C.POINTER(C.c_double)),
                ("rc", C.c_int32), ("n_ds", C.c_int32), ("n_pass", C.c_int32), ("n_knn_pass", C.c_int32)]


class GpsObservation(C.Structure):  # lio_gps_observation
    _fields_ = [("T", C.c_double * 16), ("precision", C.c_double), ("dimension", C.c_int32)]


class NdtParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("lm_max_iterations", C.c_int32), ("rotation_epsilon_deg", C.c_double),
                ("transformation_epsilon", C.c_double), ("lm_init_lambda_factor", C.c_double), ("max_process_time_ms", C.c_double)]


class AlignJob(C.Structure):  # lio_align_job
    _fields_ = [("target", C.c_void_p), ("source", C.c_void_p), ("guess", C.POINTER(C.c_double)), ("out", C.c_double * 16), ("iterations", C.c_int32), ("converged", C.c_int32),
                ("evaluations", C.c_int32), ("rc", C.c_int32)]


class NdtTimes(C.Structure):  # lio_ndt_times
    _fields_ = [("cost_us", C.c_double), ("launches", C.c_uint64), ("update_launches", C.c_uint64), ("pairs", C.c_uint64), ("source_points", C.c_uint64)]


class KernelTimes(C.Structure):
    _fields_ = [("knn_us", C.c_double), ("linearize_us", C.c_double), ("finalize_us", C.c_double), ("knn_launches", C.c_uint32),
                ("linearize_launches", C.c_uint32), ("finalize_launches", C.c_uint32), ("pad", C.c_uint32)]


REDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p)  # lio_gather_fn: (ctx, d_local, d_gathered, n_records, stream)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback for the LIO hot path.")
    L = C.CDLL(LIB_PATH)
    vp, f32p, f64p, i32p, u8p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    u32, u64, dbl, flt, cint = C.c_uint32, C.c_uint64, C.c_double, C.c_float, C.c_int

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("lio_last_error", C.c_char_p)
    sig("lio_last_warning", C.c_char_p)
    sig("lio_device_count", cint)
    sig("lio_abi_version", cint)
    sig("lio_pinned_alloc", vp, u64)
    sig("lio_pinned_free", None, vp)
    sig("lio_map_bytes", u64, vp)
    sig("lio_map_create", vp, cint, flt, u64, u64, cint)
    sig("lio_map_destroy", None, vp)
    sig("lio_map_set_stencil", cint, vp, cint)
    sig("lio_map_insert", cint, vp, f32p, u64, dbl)
    sig("lio_map_insert_device", cint, vp, vp, u64, dbl)
    sig("lio_map_stats", cint, vp, C.POINTER(u64), C.POINTER(u64))
    sig("lio_map_dump", C.c_int64, vp, f32p, u64)
    sig("lio_map_set_lru", cint, vp, u64, dbl)
    sig("lio_map_lru_stats", cint, vp, C.POINTER(u64), C.POINTER(u64))
    sig("lio_map_lru_exact_stats", cint, vp, C.POINTER(u64), C.POINTER(u64))
    sig("lio_map_clear", cint, vp)
    sig("lio_map_set_tie_mode", cint, vp, cint)
    sig("lio_map_tie_stats", cint, vp, C.POINTER(u64), C.POINTER(u64))
    sig("lio_map_pool_stats", cint, vp, C.POINTER(u64), C.POINTER(u64))
    sig("lio_map_knn_candidates", u64, vp)
    sig("lio_map_knn_touched", u64, vp)
    sig("lio_map_knn_unique", u64, vp)
    sig("lio_map_knn", cint, vp, f32p, u32, f32p, i32p)
    sig("lio_scan_create", vp, cint, u32, u32)
    sig("lio_scan_destroy", None, vp)
    sig("lio_scan_reset", cint, vp)
    sig("lio_scan_upload", cint, vp, f32p, u32)
    sig("lio_scan_set_device", cint, vp, vp, u32)
    sig("lio_scan_undistort_delta", cint, vp, vp, cint, f32p, C.c_double)
    sig("lio_scan_undistort_poses", cint, vp, vp, cint, u64, C.POINTER(u64), f64p, u32)
    sig("lio_scan_download_raw", cint, vp, f32p, u32)
    sig("lio_scan_voxel_downsample", cint, vp, flt, cint, C.POINTER(u32))
    sig("lio_scan_voxel_downsample_batch", cint, C.POINTER(vp), cint, flt, C.POINTER(u32))
    sig("lio_scan_set_ds", cint, vp, f32p, u32)
    sig("lio_scan_num_ds", cint, vp)
    sig("lio_scan_download_ds", cint, vp, f32p, u32)
    sig("lio_scan_download_world", cint, vp, f32p, u32)
    sig("lio_scan_download_match", cint, vp, u8p, f32p, i32p, f32p)
    sig("lio_p2plane_linearize", cint, vp, vp, f64p, f64p, cint, C.POINTER(NormalEq))
    sig("lio_scan_set_degeneracy_mode", cint, vp, cint)
    sig("lio_p2plane_degeneracy", cint, vp, f64p, f64p, f64p)
    sig("lio_engine_set_reduce_hook", cint, vp, REDUCE_FN, vp)
    sig("lio_p2plane_rows", cint, vp, f64p, f64p, f64p, f64p, u32)
    sig("lio_map_incremental", cint, vp, vp, f64p, f64p, flt, cint, dbl)
    sig("lio_map_seed", cint, vp, vp, f64p, f64p, dbl)
    sig("lio_engine_create", vp, cint, flt, cint, u64, u64, u32, u32)
    sig("lio_engine_create_shared", vp, vp, u32, u32)
    sig("lio_engine_destroy", None, vp)
    sig("lio_engine_map", vp, vp)
    sig("lio_engine_scan", vp, vp)
    sig("lio_engine_set_state", cint, vp, f64p)
    sig("lio_engine_get_state", cint, vp, f64p)
    sig("lio_engine_set_cov", cint, vp, f64p)
    sig("lio_engine_get_cov", cint, vp, f64p)
    sig("lio_engine_set_flags", cint, vp, cint, cint, dbl, dbl)
    sig("lio_engine_travel", dbl, vp)
    sig("lio_engine_is_degenerate", cint, vp)
    sig("lio_engine_update", cint, vp)
    sig("lio_engine_pass_log", cint, vp, cint, C.POINTER(PassLog))
    sig("lio_engine_process_scan", cint, vp, f32p, u32, dbl)
    sig("lio_engine_process_scan_device", cint, vp, vp, u32, dbl)
    sig("lio_engine_timings", cint, vp, C.POINTER(Timings))
    sig("lio_engine_flush", cint, vp)
    sig("lio_engine_enable_timing", cint, vp, cint)
    sig("lio_engines_process_batch", cint, C.POINTER(vp), cint, C.POINTER(ScanJob), cint)
    sig("lio_comm_unique_id", cint, C.POINTER(C.c_uint8))
    sig("lio_comm_init", vp, cint, cint, cint, C.POINTER(C.c_uint8))
    sig("lio_comm_destroy", None, vp)
    sig("lio_comm_rank", cint, vp)
    sig("lio_comm_world", cint, vp)
    sig("lio_allgather_normal_eq", cint, vp, vp, vp, vp, vp)
    sig("lio_comm_stats", cint, vp, C.POINTER(u64), f64p)
    sig("lio_engine_set_joint", cint, vp, C.POINTER(vp), cint, vp)
    sig("lio_engine_joint_register", cint, vp, f32p, u32, dbl, f64p, f64p)
    sig("lio_engine_joint_register_device", cint, vp, vp, u32, dbl, f64p, f64p)
    sig("lio_allgather_records", cint, vp, vp, vp, u32, vp)
    sig("lio_batch_create_joint", vp, C.POINTER(vp), cint, vp, cint, cint, u32, u32)
    sig("lio_batch_set_gather_hook", cint, vp, GATHER_FN, vp, cint, cint)
    sig("lio_batch_exchange_stats", cint, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
    sig("lio_batch_create", vp, vp, cint, cint, u32, u32)
    sig("lio_batch_create_sequences", vp, cint, C.c_float, cint, u64, u64, cint, cint, u32, u32)
    sig("lio_batch_sequences_step", cint, vp, C.POINTER(ScanJob), cint, C.POINTER(C.c_double))
    sig("lio_batch_fastlio_main", cint, vp, C.POINTER(cint))
    sig("lio_batch_destroy", None, vp)
    sig("lio_batch_process", cint, vp, C.POINTER(ScanJob), cint)
    sig("lio_batch_engine", vp, vp, cint, cint)
    sig("lio_batch_enable_kernel_timing", cint, vp, cint)
    sig("lio_batch_kernel_times", cint, vp, C.POINTER(BatchTimes), cint)
    sig("lio_engine_set_static_map", cint, vp, cint)
    sig("lio_localmap_create", vp, cint, u64, u32, u32)
    sig("lio_localmap_destroy", None, vp)
    sig("lio_localmap_add_keyframe", cint, vp, f32p, u32, f32p)
    sig("lio_localmap_num_keyframes", cint, vp)
    sig("lio_localmap_update", cint, vp, vp, f64p, dbl, dbl, dbl, flt, C.POINTER(cint), C.POINTER(u32))
    sig("lio_localmap_download", cint, vp, f32p, u32)
    sig("lio_pose_estimator_create", vp, f32p, u64, f32p, f32p, dbl)
    sig("lio_pose_estimator_destroy", None, vp)
    sig("lio_pose_estimator_predict", cint, vp, u64, f32p, f32p)
    sig("lio_pose_estimator_match", cint, vp, vp, vp, C.POINTER(NdtParams), f32p, C.POINTER(cint))
    sig("lio_pose_estimator_match_gps", cint, vp, vp, vp, C.POINTER(NdtParams), C.POINTER(GpsObservation), f32p, f32p, C.POINTER(cint))
    sig("lio_pose_estimator_guess", cint, vp, C.POINTER(GpsObservation), f32p)
    sig("lio_pose_estimator_observe", cint, vp, f32p, f32p, cint, C.POINTER(GpsObservation), f32p, f32p)
    sig("lio_pose_estimator_match_gps_only", cint, vp, C.POINTER(GpsObservation), f32p, f32p)
    sig("lio_pose_estimator_get_timed_pose", cint, vp, u64, f64p, f64p, f64p)
    sig("lio_pose_estimator_predict_nostate", cint, vp, u64, f64p)
    sig("lio_pose_estimator_correct", cint, vp, u64, f32p)
    sig("lio_pose_estimator_get_dt", u64, vp)
    sig("lio_pose_estimator_get", cint, vp, f32p, f32p)
    sig("lio_pose_estimator_set", cint, vp, f32p, f32p)
    sig("lio_pose_estimator_matrix", cint, vp, f32p)
    sig("lio_fastlio_init", cint, vp, f64p, f64p, cint, cint, dbl, cint)
    sig("lio_fastlio_is_init", cint, vp)
    sig("lio_fastlio_imu_enqueue", cint, vp, dbl, f64p, f64p)
    sig("lio_engine_set_device_loop", cint, vp, cint)
    sig("lio_fastlio_set_wheelspeed", cint, vp, cint)
    sig("lio_fastlio_ins_enqueue", cint, vp, dbl, f64p)
    sig("lio_fastlio_pcl_enqueue", cint, vp, f32p, C.POINTER(u32), u32, dbl)
    sig("lio_fastlio_pcl_enqueue_device", cint, vp, vp, vp, u32, dbl)
    sig("lio_fastlio_main", cint, vp)
    sig("lio_fastlio_odometry", cint, vp, f64p, f64p)
    sig("lio_fastlio_state", cint, vp, f64p)
    sig("lio_fastlio_start_state", cint, vp, f64p)
    sig("lio_fastlio_download_undistorted", cint, vp, f32p, u32, C.POINTER(u32))
    sig("lio_eskf_update_cb", cint, f64p, f64p, dbl, cint, MEAS_FN, vp, cint, f64p, f64p)
    sig("lio_eskf_update_ws_cb", cint, f64p, f64p, dbl, cint, MEAS_FN, vp, cint, f64p, cint, f64p, f64p)
    sig("lio_state_predict", cint, f64p, f64p, dbl, f64p, f64p, f64p, f64p, f64p)
    sig("lio_eskf_update_sums_cb", cint, f64p, f64p, dbl, cint, cint, SUMS_FN, DEG_FN, vp, f64p, f64p, C.POINTER(PassLog), cint, C.POINTER(cint))
    sig("lio_eskf_update_cb", cint, f64p, f64p, dbl, cint, MEAS_FN, vp, cint, f64p, f64p)
    sig("lio_scan_enable_kernel_timing", cint, vp, cint)
    sig("lio_scan_kernel_times", cint, vp, C.POINTER(KernelTimes), cint)
    sig("lio_ndt_create", vp, cint, flt, cint, u64, u64, u32)
    sig("lio_ndt_destroy", None, vp)
    sig("lio_ndt_set_target", cint, vp, f32p, u64)
    sig("lio_ndt_set_target_device", cint, vp, vp, u64)
    sig("lio_ndt_fitness_score", cint, vp, vp, f64p, dbl, f64p, C.POINTER(u32))
    sig("lio_ndt_overlap_score", cint, vp, vp, f64p, dbl, dbl, dbl, f64p, f64p)
    sig("lio_ndt_num_voxels", cint, vp)
    sig("lio_ndt_voxel_at", cint, vp, f32p, f32p, f32p)
    sig("lio_ndt_linearize", cint, vp, vp, f64p, cint, cint, f64p, f64p, f64p, C.POINTER(u32))
    sig("lio_ndt_default_params", None, C.POINTER(NdtParams))
    sig("lio_ndt_align", cint, vp, vp, f64p, C.POINTER(NdtParams), f64p, C.POINTER(cint), C.POINTER(cint))
    sig("lio_ndt_align_batch", cint, vp, C.POINTER(AlignJob), cint, C.POINTER(NdtParams))
    sig("lio_ndt_enable_kernel_timing", cint, vp, cint)
    sig("lio_ndt_kernel_times", cint, vp, C.POINTER(NdtTimes), cint)
    sig("lio_gicp_create", vp, cint, flt, u32, cint)
    sig("lio_gicp_destroy", None, vp)
    sig("lio_gicp_set_target", cint, vp, f32p, u32)
    sig("lio_gicp_set_source", cint, vp, f32p, u32)
    sig("lio_gicp_set_voxel_mode", cint, vp, C.c_double, cint)
    sig("lio_gicp_voxel_at", cint, vp, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double))
    sig("lio_gicp_download", cint, vp, cint, f32p, f64p, u32)
    sig("lio_gicp_correspondences", cint, vp, i32p, u32)
    sig("lio_gicp_linearize", cint, vp, f64p, dbl, cint, cint, f64p, f64p, f64p, C.POINTER(u32))
    sig("lio_gicp_align", cint, vp, f64p, C.POINTER(NdtParams), dbl, f64p, C.POINTER(cint), C.POINTER(cint))
    sig("lio_state_boxplus", None, f64p, f64p, f64p)
    sig("lio_state_boxminus", None, f64p, f64p, f64p)
    _lib = L
    return L


class LioError(RuntimeError):
    pass


def check(rc, what=""):
    if rc < 0:
        msg = lib().lio_last_error()
        raise LioError(f"{what}: error {rc}: {msg.decode() if msg else ''}")
    return rc


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))
