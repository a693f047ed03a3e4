"""ctypes access to oracle/_ref/libref_gicp.so: the reference's fine matcher fast_gicp::FastGICP<PointXYZI, PointXYZI> compiled from
/root/reference by `make -C oracle ref` (oracle/ref_gicp.cpp), configured as select_registration_method("FAST_GICP").  CPU only.  Test
infrastructure only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_gicp.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        f32p, f64p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.ref_gicp_create.restype = C.c_void_p
        L.ref_gicp_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_float]
        L.ref_gicp_destroy.argtypes = [C.c_void_p]
        L.ref_gicp_set_target.argtypes = [C.c_void_p, f32p, C.c_int, f64p]
        L.ref_gicp_set_source.argtypes = [C.c_void_p, f32p, C.c_int, f64p]
        L.ref_gicp_linearize.restype = C.c_double
        L.ref_gicp_linearize.argtypes = [C.c_void_p, f64p, f64p, f64p, i32p, f32p, f64p]
        L.ref_gicp_compute_error.restype = C.c_double
        L.ref_gicp_compute_error.argtypes = [C.c_void_p, f64p]
        L.ref_gicp_align.argtypes = [C.c_void_p, f32p, f32p, i32p]
        L.ref_gicp_transform_f.argtypes = [f64p, f32p, f32p]
        L.ref_vgicp_create.restype = C.c_void_p
        L.ref_vgicp_create.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_float]
        L.ref_vgicp_destroy.argtypes = [C.c_void_p]
        L.ref_vgicp_set_target.argtypes = [C.c_void_p, f32p, C.c_int]
        L.ref_vgicp_set_source.argtypes = [C.c_void_p, f32p, C.c_int]
        L.ref_vgicp_linearize.restype = C.c_double
        L.ref_vgicp_linearize.argtypes = [C.c_void_p, f64p, f64p, f64p, i32p]
        L.ref_vgicp_compute_error.restype = C.c_double
        L.ref_vgicp_compute_error.argtypes = [C.c_void_p, f64p]
        L.ref_vgicp_voxel_at.argtypes = [C.c_void_p, f32p, f64p, f64p]
        L.ref_vgicp_align.argtypes = [C.c_void_p, f32p, f32p, i32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def transform_f(T, xyz):
    """trans.cast<float>() * [x y z 1] as FastGICP::update_correspondences evaluates it (f32)."""
    T = np.ascontiguousarray(T, np.float64)
    xyz = np.ascontiguousarray(xyz, np.float32)
    out = np.zeros(3, np.float32)
    lib().ref_gicp_transform_f(_p(T, C.c_double), _p(xyz, C.c_float), _p(out, C.c_float))
    return out


class RefGicp:
    def __init__(self, k=20, max_corr_dist=2.0, transformation_epsilon=0.01, max_iterations=64, num_threads=4, kdtree_cell=1.0):
        self.h = lib().ref_gicp_create(k, max_corr_dist, transformation_epsilon, max_iterations, num_threads, kdtree_cell)
        self.n_src = 0

    def close(self):
        if self.h:
            lib().ref_gicp_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _set(self, fn, xyzi):
        xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        cov = np.zeros((len(xyzi), 3, 3))
        fn(self.h, _p(xyzi, C.c_float), len(xyzi), _p(cov, C.c_double))
        return cov

    def set_target(self, xyzi):
        return self._set(lib().ref_gicp_set_target, xyzi)

    def set_source(self, xyzi):
        self.n_src = len(np.asarray(xyzi).reshape(-1, 4))
        return self._set(lib().ref_gicp_set_source, xyzi)

    def linearize(self, T):
        """-> (err, H 6x6, b 6, corr int32[n], sq_dist f32[n], mahalanobis [n,3,3] (rows without a correspondence zero))"""
        T = np.ascontiguousarray(T, np.float64)
        H, b = np.zeros((6, 6)), np.zeros(6)
        corr, sq, maha = np.zeros(self.n_src, np.int32), np.zeros(self.n_src, np.float32), np.zeros((self.n_src, 3, 3))
        e = lib().ref_gicp_linearize(self.h, _p(T, C.c_double), _p(H, C.c_double), _p(b, C.c_double), _p(corr, C.c_int), _p(sq, C.c_float), _p(maha, C.c_double))
        return e, H, b, corr, sq, maha

    def compute_error(self, T):
        T = np.ascontiguousarray(T, np.float64)
        return lib().ref_gicp_compute_error(self.h, _p(T, C.c_double))

    def align(self, guess):
        g = np.ascontiguousarray(guess, np.float32)
        T = np.zeros((4, 4), np.float32)
        it = C.c_int(0)
        conv = lib().ref_gicp_align(self.h, _p(g, C.c_float), _p(T, C.c_float), C.byref(it))
        return T, it.value, bool(conv)


class RefVgicp:
    """fast_gicp::FastVGICP<PointXYZI, PointXYZI> as select_registration_method("FAST_VGICP") configures it (registrations.cpp:56-66)"""

    def __init__(self, k=20, resolution=1.0, search_method=1, transformation_epsilon=0.1, rotation_epsilon=0.1, max_iterations=64, num_threads=4, kdtree_cell=1.0):
        self.h = lib().ref_vgicp_create(k, resolution, search_method, transformation_epsilon, rotation_epsilon, max_iterations, num_threads, kdtree_cell)

    def close(self):
        if self.h:
            lib().ref_vgicp_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_target(self, xyzi):
        xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        lib().ref_vgicp_set_target(self.h, _p(xyzi, C.c_float), len(xyzi))

    def set_source(self, xyzi):
        xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        lib().ref_vgicp_set_source(self.h, _p(xyzi, C.c_float), len(xyzi))

    def linearize(self, T):
        T = np.ascontiguousarray(T, np.float64)
        H, b, nc = np.zeros((6, 6)), np.zeros(6), C.c_int(0)
        e = lib().ref_vgicp_linearize(self.h, _p(T, C.c_double), _p(H, C.c_double), _p(b, C.c_double), C.byref(nc))
        return e, H, b, nc.value

    def compute_error(self, T):
        T = np.ascontiguousarray(T, np.float64)
        return lib().ref_vgicp_compute_error(self.h, _p(T, C.c_double))

    def voxel_at(self, p):
        p = np.ascontiguousarray(p, np.float32)
        m, c = np.zeros(3), np.zeros((3, 3))
        n = lib().ref_vgicp_voxel_at(self.h, _p(p, C.c_float), _p(m, C.c_double), _p(c, C.c_double))
        return n, m, c

    def align(self, guess):
        g = np.ascontiguousarray(guess, np.float32)
        T = np.zeros((4, 4), np.float32)
        it = C.c_int(0)
        conv = lib().ref_vgicp_align(self.h, _p(g, C.c_float), _p(T, C.c_float), C.byref(it))
        return T, it.value, bool(conv)
