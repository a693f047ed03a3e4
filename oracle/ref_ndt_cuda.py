"""ctypes access to oracle/_ref/libref_ndt_cuda.so: the reference's OWN GPU matcher core (fast_gicp::cuda::NDTCudaCore and its
CUDA / Thrust kernels) compiled for gfx950 by `make -C oracle ref` (oracle/ref_ndt_cuda.hip).  Needs a GPU to run.  Test
infrastructure only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_ndt_cuda.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        f32p, f64p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.ref_ndt_create.restype = C.c_void_p
        L.ref_ndt_create.argtypes = [C.c_double, C.c_int]
        L.ref_ndt_destroy.argtypes = [C.c_void_p]
        L.ref_ndt_set_target.argtypes = [C.c_void_p, f32p, C.c_int]
        L.ref_ndt_set_source.argtypes = [C.c_void_p, f32p, C.c_int]
        L.ref_ndt_num_voxels.argtypes = [C.c_void_p]
        L.ref_ndt_voxels.argtypes = [C.c_void_p, i32p, i32p, f32p, f32p, C.c_int]
        L.ref_ndt_update_correspondences.argtypes = [C.c_void_p, f64p]
        L.ref_ndt_compute_error.argtypes = [C.c_void_p, f64p, f64p, f64p]
        L.ref_ndt_compute_error.restype = C.c_double
        L.ref_ndt_valid_pairs.argtypes = [C.c_void_p]
        L.ref_ndtreg_create.restype = C.c_void_p
        L.ref_ndtreg_create.argtypes = [C.c_double, C.c_int, C.c_double]
        L.ref_ndtreg_destroy.argtypes = [C.c_void_p]
        L.ref_ndtreg_set_target.argtypes = [C.c_void_p, f32p, C.c_int]
        L.ref_ndtreg_set_source.argtypes = [C.c_void_p, f32p, C.c_int]
        L.ref_ndtreg_align.argtypes = [C.c_void_p, f32p, f32p, i32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class NdtCudaCore:
    def __init__(self, resolution=1.0, search_method=7):
        self.h = lib().ref_ndt_create(float(resolution), int(search_method))

    def close(self):
        if getattr(self, "h", None):
            lib().ref_ndt_destroy(self.h)
        self.h = None

    def set_target(self, xyzi):
        p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        lib().ref_ndt_set_target(self.h, _p(p, C.c_float), len(p))

    def set_source(self, xyzi):
        p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        lib().ref_ndt_set_source(self.h, _p(p, C.c_float), len(p))

    @property
    def num_voxels(self):
        return lib().ref_ndt_num_voxels(self.h)

    def voxels(self):
        """(coord n x 3 int, point count, mean n x 3, covariance n x 3 x 3 after the PLANE regularisation)"""
        cap = self.num_voxels + 16
        co, nn, me, cv = np.zeros((cap, 3), np.int32), np.zeros(cap, np.int32), np.zeros((cap, 3), np.float32), np.zeros((cap, 9), np.float32)
        n = lib().ref_ndt_voxels(self.h, _p(co, C.c_int), _p(nn, C.c_int), _p(me, C.c_float), _p(cv, C.c_float), cap)
        assert n <= cap
        return co[:n], nn[:n], me[:n], cv[:n].reshape(-1, 3, 3)

    def linearize(self, T):
        """NDTCuda::linearize: update_correspondences(T) + compute_error(T, H, b)"""
        t = np.ascontiguousarray(T, np.float64).reshape(16)
        lib().ref_ndt_update_correspondences(self.h, _p(t, C.c_double))
        H, b = np.zeros(36), np.zeros(6)
        err = lib().ref_ndt_compute_error(self.h, _p(t, C.c_double), _p(H, C.c_double), _p(b, C.c_double))
        return dict(n_corr=lib().ref_ndt_valid_pairs(self.h), H=H.reshape(6, 6), b=b, err=err)

    def compute_error(self, T):
        t = np.ascontiguousarray(T, np.float64).reshape(16)
        return lib().ref_ndt_compute_error(self.h, _p(t, C.c_double), None, None)


class NdtCudaRegistration:
    """fast_gicp::NDTCuda<PointXYZI, PointXYZI> configured as select_registration_method("NDT_CUDA") does (registrations.cpp:107-118):
    setInputTarget / setInputSource / align(guess) with LsqRegistration's Levenberg-Marquardt loop"""

    def __init__(self, resolution=1.0, search_method=7, max_process_time_ms=-1):
        self.h = lib().ref_ndtreg_create(float(resolution), int(search_method), float(max_process_time_ms))

    def close(self):
        if getattr(self, "h", None):
            lib().ref_ndtreg_destroy(self.h)
        self.h = None

    def set_target(self, xyzi):
        p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        lib().ref_ndtreg_set_target(self.h, _p(p, C.c_float), len(p))

    def set_source(self, xyzi):
        p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        lib().ref_ndtreg_set_source(self.h, _p(p, C.c_float), len(p))

    def align(self, guess):
        g, T, it = np.ascontiguousarray(guess, np.float32).reshape(16), np.zeros(16, np.float32), C.c_int(0)
        conv = lib().ref_ndtreg_align(self.h, _p(g, C.c_float), _p(T, C.c_float), C.byref(it))
        return T.reshape(4, 4).astype(np.float64), bool(conv), it.value
