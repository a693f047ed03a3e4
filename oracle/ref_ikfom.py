"""ctypes access to oracle/_ref/libref_ikfom.so: the reference's OWN IKFoM filter (esekfom.hpp predict /
update_iterated_dyn_share_modified, MTK manifold types, use-ikfom.hpp process model) compiled from /root/reference by
`make -C oracle ref`.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_ikfom.so")
_lib = None
MEAS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        f64p = C.POINTER(C.c_double)
        L.ref_kf_predict.argtypes = [f64p, f64p, C.c_double, f64p, f64p, f64p, f64p, f64p]
        L.ref_process_noise_cov.argtypes = [f64p]
        L.ref_kf_update.argtypes = [f64p, f64p, C.c_double, C.c_int, MEAS_FN, C.c_void_p, C.c_int, f64p, f64p]
        L.ref_state_boxplus.argtypes = [f64p, f64p, f64p]
        L.ref_state_boxminus.argtypes = [f64p, f64p, f64p]
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def predict(s, P, dt, Q12, acc, gyro):
    s, P, q, a, g = _d(s), _d(P).reshape(-1), _d(Q12), _d(acc), _d(gyro)
    so, Po = np.zeros(26), np.zeros(529)
    lib().ref_kf_predict(_p(s), _p(P), float(dt), _p(q), _p(a), _p(g), _p(so), _p(Po))
    return so, Po.reshape(23, 23)


def process_noise_cov():
    Q = np.zeros(144)
    lib().ref_process_noise_cov(_p(Q))
    return Q.reshape(12, 12)


def update(s, P, R, meas_fn, max_iter=4, cap=4096):
    s, P = _d(s), _d(P).reshape(-1)
    so, Po = np.zeros(26), np.zeros(529)
    lib().ref_kf_update(_p(s), _p(P), float(R), max_iter, meas_fn, None, cap, _p(so), _p(Po))
    return so, Po.reshape(23, 23)


def state_boxplus(s, d):
    s, d = _d(s), _d(d)
    o = np.zeros(26)
    lib().ref_state_boxplus(_p(s), _p(d), _p(o))
    return o


def state_boxminus(a, b):
    a, b = _d(a), _d(b)
    o = np.zeros(23)
    lib().ref_state_boxminus(_p(a), _p(b), _p(o))
    return o
