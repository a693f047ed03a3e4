"""ctypes access to oracle/_ref/libref_hdl_localization.so: the reference's own localisation loop (hdl_localization_nodelet.cpp + pose_estimator.cpp,
compiled whole) LINKED against the product's liblio_hip.so through the NdtHip class INTEGRATION.md section 3a shows -- or, in another process, over the
reference's own fast_gicp::NDTCuda (libref_ndt_cuda.so).  Needs a GPU to run.  One variant per process (the nodelet keeps its two matcher objects in
file-scope statics).  Test infrastructure."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_hdl_localization.so")
_lib = None


def available():
    return os.path.exists(_PATH) and os.path.exists(os.path.join(_HERE, "_ref", "libref_slam_utils.so"))


def lib():
    global _lib
    if _lib is None:
        C.CDLL(os.path.join(_HERE, "liblio_oracle.so"), mode=C.RTLD_GLOBAL)  # orc_voxel_downsample for the nodelet's pcl::VoxelGrid (the shim)
        L = C.CDLL(_PATH)
        f64p, f32p, u32p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.hloc_init.argtypes = [C.c_int, C.c_double, C.c_double, f64p]
        L.hloc_set_initpose.argtypes = [C.c_uint64, f64p]
        L.hloc_set_map.argtypes = [f32p, C.c_int]
        L.hloc_imu.argtypes = [C.c_double, f64p, f64p]
        L.hloc_ins.argtypes = [C.c_uint64, f64p, C.c_double, C.c_int]
        L.hloc_frame.argtypes = [f32p, u32p, C.c_int, C.c_uint64, f64p]
        L.hloc_timed_pose.argtypes = [C.c_uint64, f64p]
        _lib = L
    return _lib


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


class HdlLocalization:
    def __init__(self, use_reference_matcher=False, resolution=0.2, scan_period=0.1, imu_ext=np.eye(4)):
        e = np.ascontiguousarray(imu_ext, np.float64)
        lib().hloc_init(int(use_reference_matcher), float(resolution), float(scan_period), _p(e))

    def close(self):
        lib().hloc_deinit()

    def set_initpose(self, stamp_us, T):
        t = np.ascontiguousarray(T, np.float64)
        lib().hloc_set_initpose(int(stamp_us), _p(t))

    def set_map(self, xyzi):
        if xyzi is None:
            lib().hloc_set_map(None, 0)
            return
        p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        lib().hloc_set_map(_p(p, C.c_float), len(p))

    def imu(self, stamp_s, acc, gyr):
        a, g = np.ascontiguousarray(acc, np.float64), np.ascontiguousarray(gyr, np.float64)
        lib().hloc_imu(float(stamp_s), _p(a), _p(g))

    def ins(self, stamp_us, T, precision=1.0, dimension=6):
        t = np.ascontiguousarray(T, np.float64)
        lib().hloc_ins(int(stamp_us), _p(t), float(precision), int(dimension))

    def frame(self, xyzi, stamp_us, header_stamp_us):
        """frame_callback: returns (LocType 0 OK / 1 ERROR / 2 OTHER, pose 4 x 4)"""
        p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        s = np.ascontiguousarray(stamp_us, np.uint32)
        T = np.zeros(16)
        r = lib().hloc_frame(_p(p, C.c_float), _p(s, C.c_uint32), len(p), int(header_stamp_us), _p(T))
        return r, T.reshape(4, 4)

    def timed_pose(self, stamp_us):
        T = np.zeros(16)
        ok = lib().hloc_timed_pose(int(stamp_us), _p(T))
        return bool(ok), T.reshape(4, 4)
