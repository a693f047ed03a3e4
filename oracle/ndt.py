"""ctypes binding of oracle/libndt_oracle.so (CPU oracle of the NDT-P2D localization matcher) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libndt_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE, "libndt_oracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(_SO)
        f32p, f64p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.ndt_create.argtypes = [C.c_float, C.c_int]
        L.ndt_create.restype = C.c_void_p
        L.ndt_destroy.argtypes = [C.c_void_p]
        L.ndt_set_params.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.ndt_set_target.argtypes = [C.c_void_p, f32p, C.c_int]
        L.ndt_num_voxels.argtypes = [C.c_void_p]
        L.ndt_voxel_at.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p]
        L.ndt_set_source.argtypes = [C.c_void_p, f32p, C.c_int]
        L.ndt_linearize.argtypes = [C.c_void_p, f64p, f64p, f64p, f64p]
        L.ndt_compute_error.argtypes = [C.c_void_p, f64p]
        L.ndt_compute_error.restype = C.c_double
        L.ndt_align.argtypes = [C.c_void_p, f64p, f64p, i32p]
        L.ndt_se3_exp.argtypes = [f64p, f64p]
        L.ndt_regularize_plane.argtypes = [f32p, f32p, f32p]
        L.ndt_eig3_direct.argtypes = [f32p, f32p, f32p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Ndt:
    def __init__(self, resolution=1.0, method=7):
        self.h = lib().ndt_create(resolution, method)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ndt_destroy(self.h)
            self.h = None

    def set_params(self, max_iter=64, rot_eps_deg=0.1, trans_eps=0.01):
        lib().ndt_set_params(self.h, max_iter, rot_eps_deg, trans_eps)

    def set_target(self, pts):
        p = _f32(pts).reshape(-1, 4)
        lib().ndt_set_target(self.h, _p(p, C.c_float), len(p))

    @property
    def num_voxels(self):
        return lib().ndt_num_voxels(self.h)

    def voxel_at(self, p):
        p = _f32(p)
        mean, cov, cinv = np.zeros(3, np.float32), np.zeros(9, np.float32), np.zeros(9, np.float32)
        n = lib().ndt_voxel_at(self.h, _p(p, C.c_float), _p(mean, C.c_float), _p(cov, C.c_float), _p(cinv, C.c_float))
        return n, mean, cov.reshape(3, 3), cinv.reshape(3, 3)

    def set_source(self, pts):
        p = _f32(pts).reshape(-1, 4)
        lib().ndt_set_source(self.h, _p(p, C.c_float), len(p))

    def linearize(self, T):
        T = _f64(T).reshape(4, 4)
        H, b, e = np.zeros(36), np.zeros(6), C.c_double(0)
        n = lib().ndt_linearize(self.h, _p(T, C.c_double), _p(H, C.c_double), _p(b, C.c_double), C.byref(e))
        return dict(n_corr=n, H=H.reshape(6, 6), b=b, err=e.value)

    def compute_error(self, T):
        T = _f64(T).reshape(4, 4)
        return lib().ndt_compute_error(self.h, _p(T, C.c_double))

    def align(self, guess):
        g = _f64(guess).reshape(4, 4)
        out = np.zeros((4, 4))
        it = C.c_int(0)
        conv = lib().ndt_align(self.h, _p(g, C.c_double), _p(out, C.c_double), C.byref(it))
        return out, bool(conv), it.value


def se3_exp(a):
    a = _f64(a)
    T = np.zeros((4, 4))
    lib().ndt_se3_exp(_p(a, C.c_double), _p(T, C.c_double))
    return T


def regularize_plane(cov):
    c = _f32(cov).reshape(3, 3)
    out, inv = np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32)
    lib().ndt_regularize_plane(_p(c, C.c_float), _p(out, C.c_float), _p(inv, C.c_float))
    return out, inv


def eig3_direct(cov):
    c = _f32(cov).reshape(3, 3)
    w, V = np.zeros(3, np.float32), np.zeros((3, 3), np.float32)
    lib().ndt_eig3_direct(_p(c, C.c_float), _p(w, C.c_float), _p(V, C.c_float))
    return w, V
