"""ctypes access to oracle/_ref/libref_ukf.so: the reference's own UKF + pose system (hdl_localization), compiled from
/root/reference by `make -C oracle ref`.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_ukf.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        f32p = C.POINTER(C.c_float)
        L.ref_ukf_create.argtypes = [f32p, f32p, f32p]
        L.ref_ukf_create.restype = C.c_void_p
        L.ref_ukf_destroy.argtypes = [C.c_void_p]
        L.ref_ukf_predict.argtypes = [C.c_void_p, C.c_double, f32p]
        L.ref_ukf_correct.argtypes = [C.c_void_p, f32p]
        L.ref_ukf_get.argtypes = [C.c_void_p, f32p, f32p]
        _lib = L
    return _lib


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Ukf:
    def __init__(self, imu_ext, pos, quat_wxyz):
        a, b, c = _f(imu_ext).reshape(-1), _f(pos), _f(quat_wxyz)
        self.h = lib().ref_ukf_create(_p(a), _p(b), _p(c))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_ukf_destroy(self.h)
            self.h = None

    def predict(self, dt, control=None):
        if control is None:
            lib().ref_ukf_predict(self.h, float(dt), None)
        else:
            c = _f(control)
            lib().ref_ukf_predict(self.h, float(dt), _p(c))

    def correct(self, obs):
        z = _f(obs)
        lib().ref_ukf_correct(self.h, _p(z))

    def get(self):
        m, c = np.zeros(23, np.float32), np.zeros(529, np.float32)
        lib().ref_ukf_get(self.h, _p(m), _p(c))
        return m, c.reshape(23, 23)
