"""ctypes wrapper of oracle/_ref/libref_voxelgrid_cov.so: the reference tree's own pclomp::VoxelGridCovariance (PCL's VoxelGridCovariance, vendored
with ndt_omp) compiled from where it lies -- bounding box, overflow guard, voxel keys and per-leaf f32 sums of pcl::VoxelGrid's applyFilter,
statement for statement (oracle/ref_voxelgrid_cov.cpp).  TEST INFRASTRUCTURE."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_voxelgrid_cov.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        _lib.ref_vgc_filter.restype = C.c_int
        _lib.ref_vgc_filter.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return _lib


def leaves(xyzi, leaf, is_dense=True):
    """every leaf of the reference's filter in ascending key order: (keys int64 (m,), counts int32 (m,), centroids f32 (m, 4), min_b (3,), div_b (3,)),
    or None when the overflow guard fired (or no point is finite)"""
    p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
    n = len(p)
    cap = max(n, 1)
    keys, cnt, cen = np.zeros(cap, np.int64), np.zeros(cap, np.int32), np.zeros((cap, 4), np.float32)
    mb, db = np.zeros(3, np.int32), np.zeros(3, np.int32)
    m = lib().ref_vgc_filter(p.ctypes.data_as(C.POINTER(C.c_float)), n, C.c_float(leaf), int(bool(is_dense)), keys.ctypes.data_as(C.POINTER(C.c_int64)),
                             cnt.ctypes.data_as(C.POINTER(C.c_int)), cen.ctypes.data_as(C.POINTER(C.c_float)), cap, mb.ctypes.data_as(C.POINTER(C.c_int)),
                             db.ctypes.data_as(C.POINTER(C.c_int)))
    if m == -1:
        return None
    if m < 0:
        raise RuntimeError(f"ref_vgc_filter returned {m}")
    return keys[:m].copy(), cnt[:m].copy(), cen[:m].copy(), mb, db
