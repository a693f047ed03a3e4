"""ctypes binding of oracle/_ref/libref_harness.so -- the reference's own esti_plane and iVox, compiled from
/root/reference by `make -C oracle ref`.  TEST INFRASTRUCTURE ONLY (tests/, tools/make_golden.py)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libref_harness.so")


def available():
    return os.path.exists(SO)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(SO)
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.ref_esti_plane.argtypes = [f32p, C.c_float, f32p]
        L.ref_esti_plane.restype = C.c_int
        L.ref_calc_dist.argtypes = [f32p, f32p]
        L.ref_calc_dist.restype = C.c_float
        L.ref_ivox_create.argtypes = [C.c_float, C.c_int, C.c_uint64, C.c_double]
        L.ref_ivox_create.restype = C.c_void_p
        L.ref_ivox_destroy.argtypes = [C.c_void_p]
        L.ref_ivox_set_stencil.argtypes = [C.c_void_p, C.c_int]
        L.ref_ivox_add.argtypes = [C.c_void_p, f32p, C.c_int, C.c_double]
        L.ref_ivox_num_voxels.argtypes = [C.c_void_p]
        L.ref_ivox_num_voxels.restype = C.c_uint64
        L.ref_ivox_knn.argtypes = [C.c_void_p, f32p, C.c_int, f32p, i32p]
        f64p = C.POINTER(C.c_double)
        L.ref_se3_exp.argtypes = [f64p, f64p]
        L.ref_eig3_direct.argtypes = [f32p, f32p, f32p]
        L.ref_regularize_plane.argtypes = [f32p, f32p, f32p]
        L.ref_so3_Exp.argtypes = [f64p, C.c_double, f64p]
        L.ref_undistort_point.argtypes = [f64p, f64p, f64p, f64p, f64p, C.c_double, f32p, f64p, f64p, f64p, f64p, f32p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def esti_plane(five_xyzi, thr=0.1):
    p = _f32(five_xyzi).reshape(5, 4)
    out = np.zeros(4, np.float32)
    ok = lib().ref_esti_plane(_p(p, C.c_float), float(thr), _p(out, C.c_float))
    return bool(ok), out


def calc_dist(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().ref_calc_dist(_p(a, C.c_float), _p(b, C.c_float)))


def se3_exp(a):
    a = np.ascontiguousarray(a, np.float64)
    T = np.zeros((4, 4))
    lib().ref_se3_exp(_p(a, C.c_double), _p(T, C.c_double))
    return T


def eig3_direct(cov):
    c = _f32(cov).reshape(3, 3)
    w, V = np.zeros(3, np.float32), np.zeros((3, 3), np.float32)
    lib().ref_eig3_direct(_p(c, C.c_float), _p(w, C.c_float), _p(V, C.c_float))
    return w, V


def so3_Exp(w, dt):
    w = np.ascontiguousarray(w, np.float64)
    R = np.zeros(9)
    lib().ref_so3_Exp(_p(w, C.c_double), float(dt), _p(R, C.c_double))
    return R.reshape(3, 3)


def undistort_point(R_imu, vel, pos, acc, gyr, dt, p, end_pos, end_rot, ril, til):
    a = [np.ascontiguousarray(v, np.float64).ravel() for v in (R_imu, vel, pos, acc, gyr)]
    b = [np.ascontiguousarray(v, np.float64).ravel() for v in (end_pos, end_rot, ril, til)]
    pp = _f32(p)
    out = np.zeros(3, np.float32)
    lib().ref_undistort_point(*[_p(v, C.c_double) for v in a], float(dt), _p(pp, C.c_float), *[_p(v, C.c_double) for v in b], _p(out, C.c_float))
    return out


def regularize_plane(cov):
    c = _f32(cov).reshape(3, 3)
    out, inv = np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32)
    lib().ref_regularize_plane(_p(c, C.c_float), _p(out, C.c_float), _p(inv, C.c_float))
    return out, inv


class IVox:
    def __init__(self, res=0.5, stencil=19, capacity=1 << 40, max_distance=100.0):
        self.h = lib().ref_ivox_create(res, stencil, capacity, max_distance)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_ivox_destroy(self.h)
            self.h = None

    def set_stencil(self, s):
        lib().ref_ivox_set_stencil(self.h, s)

    def add(self, pts, travel=0.0):
        p = _f32(pts).reshape(-1, 4)
        lib().ref_ivox_add(self.h, _p(p, C.c_float), len(p), float(travel))

    @property
    def num_voxels(self):
        return int(lib().ref_ivox_num_voxels(self.h))

    def knn(self, q):
        q = _f32(q).reshape(-1, 4)
        out = np.zeros((len(q), 5, 4), np.float32)
        cnt = np.zeros(len(q), np.int32)
        lib().ref_ivox_knn(self.h, _p(q, C.c_float), len(q), _p(out, C.c_float), _p(cnt, C.c_int))
        return out, cnt


def canonical(nn, cnt, q):
    """sort each neighbour list by the canonical total order (d2, x, y, z) (the reference only promises
    'element 0 is the nearest, the rest in no particular order', ivox3d.h:160-165)"""
    nn = nn.copy()
    for i in range(len(nn)):
        k = int(cnt[i])
        if k > 1:
            d = nn[i, :k, :3] - q[i, :3]
            d = d.astype(np.float32)
            d2 = (d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])).astype(np.float32)  # Eigen tree order
            order = np.lexsort((nn[i, :k, 2], nn[i, :k, 1], nn[i, :k, 0], d2))
            nn[i, :k] = nn[i, :k][order]
    return nn
