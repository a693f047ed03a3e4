"""ctypes access to oracle/_ref/libref_slam_utils.so: the reference's OWN slam/common/slam_utils.cpp compiled whole from
/root/reference by `make -C oracle ref` (oracle/ref_slam_utils.cpp).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_slam_utils.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        f32p, f64p, u32p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.ref_undistort_delta.argtypes = [f32p, f32p, u32p, C.c_int, C.c_double, f32p]
        L.ref_undistort_poses.argtypes = [u64p, f64p, C.c_int, f32p, u32p, C.c_int, C.c_uint64, f32p]
        L.ref_transform_from_rpyt.argtypes = [C.c_double] * 6 + [f64p]
        L.ref_interpolate_transform.argtypes = [f64p, f64p, C.c_double, f64p]
        L.ref_distance_filter.argtypes = [f32p, C.c_int, C.c_double, C.c_double, f32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def undistort_delta(xyzi, stamp_us, delta_pose, scan_period=0.1):
    """undistortPoints(const Eigen::Matrix4f&, PointCloudAttrPtr&, double), slam_utils.cpp:163-191"""
    p, st = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4), np.ascontiguousarray(stamp_us, np.uint32)
    d = np.ascontiguousarray(delta_pose, np.float32).reshape(16)
    out = np.zeros_like(p)
    lib().ref_undistort_delta(_p(d, C.c_float), _p(p, C.c_float), _p(st, C.c_uint32), len(p), float(scan_period), _p(out, C.c_float))
    return out


def undistort_poses(xyzi, stamp_us, header_us, pose_stamps_us, pose_T):
    """undistortPoints(std::vector<PoseType>&, PointCloudAttrPtr&), slam_utils.cpp:193-228"""
    p, st = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4), np.ascontiguousarray(stamp_us, np.uint32)
    ps, pt = np.ascontiguousarray(pose_stamps_us, np.uint64), np.ascontiguousarray(pose_T, np.float64).reshape(-1, 16)
    out = np.zeros_like(p)
    lib().ref_undistort_poses(_p(ps, C.c_uint64), _p(pt, C.c_double), len(ps), _p(p, C.c_float), _p(st, C.c_uint32), len(p), int(header_us), _p(out, C.c_float))
    return out


def transform_from_rpyt(x, y, z, yaw, pitch, roll):
    T = np.zeros(16)
    lib().ref_transform_from_rpyt(x, y, z, yaw, pitch, roll, _p(T, C.c_double))
    return T.reshape(4, 4)


def interpolate_transform(A, B, ratio):
    a, b, T = np.ascontiguousarray(A, np.float64).reshape(16), np.ascontiguousarray(B, np.float64).reshape(16), np.zeros(16)
    lib().ref_interpolate_transform(_p(a, C.c_double), _p(b, C.c_double), float(ratio), _p(T, C.c_double))
    return T.reshape(4, 4)


def distance_filter(xyzi, min_range, max_range):
    p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
    out = np.zeros_like(p)
    n = lib().ref_distance_filter(_p(p, C.c_float), len(p), float(min_range), float(max_range), _p(out, C.c_float))
    return out[:n]
