"""ctypes access to oracle/_ref/libref_hdl_fastlio.so: the reference's own class Mapping::HDL_FastLIO compiled whole and LINKED against the
product's liblio_hip.so through the Option 0 binding INTEGRATION.md shows (see ref_hdl_fastlio.cpp).  Needs a GPU to run.  Test infrastructure."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_hdl_fastlio.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        f64p, f32p, u32p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.hdl_create.argtypes = [C.c_char_p, f64p, f64p, C.c_double]
        L.hdl_feed_imu.argtypes = [C.c_double, f64p, f64p]
        L.hdl_frame.argtypes = [C.c_char_p, f32p, u32p, C.c_int, C.c_uint64, f64p, f64p]
        _lib = L
    return _lib


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


class HdlFastLio:
    def __init__(self, lidar="0-lidar", T_static=np.eye(4), T_imu=np.eye(4), scan_period=0.1):
        self.name = lidar.encode()
        a, b = np.ascontiguousarray(T_static, np.float64), np.ascontiguousarray(T_imu, np.float64)
        n = lib().hdl_create(self.name, _p(a), _p(b), float(scan_period))
        assert n == 2, n  # IMU + the lidar

    def close(self):
        lib().hdl_destroy()

    def feed_imu(self, stamp, gyr, acc_ms2):
        g, a = np.ascontiguousarray(gyr, np.float64), np.ascontiguousarray(acc_ms2, np.float64)
        lib().hdl_feed_imu(float(stamp), _p(g), _p(a))

    def frame(self, xyzi, stamp_us, header_stamp_us):
        """feedPointData + getPose: returns (pose 4x4 = odom2map * odometry at the scan start, delta odometry over the scan, #predicted IMU poses)"""
        p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        s = np.ascontiguousarray(stamp_us, np.uint32)
        T, D = np.zeros(16), np.zeros(16)
        n = lib().hdl_frame(self.name, _p(p, C.c_float), _p(s, C.c_uint32), len(p), int(header_stamp_us), _p(T), _p(D))
        return T.reshape(4, 4), D.reshape(4, 4), n

    def is_init(self):
        return bool(lib().hdl_is_init())
