"""ctypes access to oracle/_ref/libref_pose_estimator.so: the reference's OWN hdl_localization::PoseEstimator compiled whole from
/root/reference (oracle/ref_pose_estimator.cpp) with a mock scan matcher behind it.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_pose_estimator.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        f32p, f64p, u64 = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_uint64
        L.ref_pe_create.restype = C.c_void_p
        L.ref_pe_create.argtypes = [f32p, u64, f32p, f32p, C.c_double]
        L.ref_pe_destroy.argtypes = [C.c_void_p]
        L.ref_pe_predict.argtypes = [C.c_void_p, u64, f32p, f32p]
        L.ref_pe_predict_nostate.argtypes = [C.c_void_p, u64, f64p]
        L.ref_pe_get_timed_pose.argtypes = [C.c_void_p, u64, f64p, f64p, f64p]
        L.ref_pe_match.argtypes = [C.c_void_p, u64, f32p, C.c_int, C.c_double, f64p, C.c_double, C.c_int, f32p, f32p, f32p, f64p]
        L.ref_pe_match_gps_only.argtypes = [C.c_void_p, f64p, C.c_double, C.c_int, f32p, f32p]
        L.ref_pe_correct.argtypes = [C.c_void_p, u64, f32p]
        L.ref_pe_matrix.argtypes = [C.c_void_p, f32p]
        L.ref_pe_state.argtypes = [C.c_void_p, f32p, f32p]
        L.ref_pe_queue.argtypes = [C.c_void_p, C.POINTER(u64), f32p, C.c_int]
        L.ref_pe_get_dt.argtypes = [C.c_void_p]
        L.ref_pe_get_dt.restype = u64
        L.ref_pe_last_correction_time.argtypes = [C.c_void_p]
        L.ref_pe_last_correction_time.restype = u64
        _lib = L
    return _lib


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


class RefPoseEstimator:
    def __init__(self, pos, quat_wxyz, stamp_us=0, imu_ext=np.eye(4), cool_time=1.0):
        a, b, c = _f(imu_ext).reshape(16), _f(pos), _f(quat_wxyz)
        self.h = lib().ref_pe_create(_p(a), int(stamp_us), _p(b), _p(c), float(cool_time))

    def close(self):
        if getattr(self, "h", None):
            lib().ref_pe_destroy(self.h)
        self.h = None

    def predict(self, stamp_us, acc=None, gyro=None):
        if acc is None:
            lib().ref_pe_predict(self.h, int(stamp_us), None, None)
        else:
            a, g = _f(acc), _f(gyro)
            lib().ref_pe_predict(self.h, int(stamp_us), _p(a), _p(g))

    def predict_nostate(self, stamp_us):
        T = np.zeros(16)
        lib().ref_pe_predict_nostate(self.h, int(stamp_us), _p(T, C.c_double))
        return T.reshape(4, 4)

    def get_timed_pose(self, stamp_us, acc_g, gyro_dps):
        a, g, T = np.ascontiguousarray(acc_g, np.float64), np.ascontiguousarray(gyro_dps, np.float64), np.zeros(16)
        ok = lib().ref_pe_get_timed_pose(self.h, int(stamp_us), _p(a, C.c_double), _p(g, C.c_double), _p(T, C.c_double))
        return bool(ok), T.reshape(4, 4)

    def match(self, stamp_us, aligned, converged=True, gps=None, fitness=0.0):
        """the mock matcher returns `aligned`; -> (ok, observation, observation_cov, the guess the matcher was given, fitness read)"""
        al = _f(aligned).reshape(16)
        obs, cov, guess, fit = np.zeros(7, np.float32), np.zeros(49, np.float32), np.zeros(16, np.float32), C.c_double(0)
        if gps is None:
            ok = lib().ref_pe_match(self.h, int(stamp_us), _p(al), int(converged), float(fitness), None, 0.0, 0, _p(obs), _p(cov), _p(guess), C.byref(fit))
        else:
            T = np.ascontiguousarray(gps[0], np.float64).reshape(16)
            ok = lib().ref_pe_match(self.h, int(stamp_us), _p(al), int(converged), float(fitness), _p(T, C.c_double), float(gps[1]), int(gps[2]), _p(obs), _p(cov),
                                    _p(guess), C.byref(fit))
        return bool(ok), obs, cov.reshape(7, 7), guess.reshape(4, 4), fit.value

    def match_gps_only(self, gps):
        obs, cov = np.zeros(7, np.float32), np.zeros(49, np.float32)
        if gps is None:
            ok = lib().ref_pe_match_gps_only(self.h, None, 0.0, 0, _p(obs), _p(cov))
        else:
            T = np.ascontiguousarray(gps[0], np.float64).reshape(16)
            ok = lib().ref_pe_match_gps_only(self.h, _p(T, C.c_double), float(gps[1]), int(gps[2]), _p(obs), _p(cov))
        return bool(ok), obs, cov.reshape(7, 7)

    def correct(self, stamp_us, observation):
        z = _f(observation)
        lib().ref_pe_correct(self.h, int(stamp_us), _p(z))

    def matrix(self):
        T = np.zeros(16, np.float32)
        lib().ref_pe_matrix(self.h, _p(T))
        return T.reshape(4, 4)

    def get(self):
        m, c = np.zeros(23, np.float32), np.zeros(529, np.float32)
        lib().ref_pe_state(self.h, _p(m), _p(c))
        return m, c.reshape(23, 23)

    def queue(self, cap=256):
        st, me = np.zeros(cap, np.uint64), np.zeros((cap, 23), np.float32)
        n = lib().ref_pe_queue(self.h, _p(st, C.c_uint64), _p(me), cap)
        return st[:n], me[:n]

    def get_dt(self):
        return int(lib().ref_pe_get_dt(self.h))
