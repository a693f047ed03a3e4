"""ctypes access to oracle/_ref/libref_localmap.so: the reference's OWN text of the local-map assembly loop body (localization.cpp:305-312,
325-372) and of OverlapDetector::filter / calc_fitness_score (overlap_merge.hpp:213-263), cut out where it lies by `make -C oracle ref` and
compiled inside oracle/ref_localmap.cpp (pcl::VoxelGrid = the oracle's restatement, kd-trees = exact stand-ins).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_localmap.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        C.CDLL(os.path.join(_HERE, "liblio_oracle.so"), mode=C.RTLD_GLOBAL)  # orc_voxel_downsample for the VoxelGrid shim
        L = C.CDLL(_PATH)
        f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.ref_lm_reset.argtypes = [C.c_double, C.c_double]
        L.ref_lm_add_keyframe.argtypes = [f32p, C.c_int, f32p]
        L.ref_lm_update.argtypes = [f64p]
        L.ref_lm_local_map.argtypes = [f32p, C.c_int]
        L.ref_overlap_fitness.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p, C.c_double, f64p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class RefLocalMap:
    """one Localization instance's local-map thread: key frames (map-frame clouds + positions), then loop turns for dequeued poses"""

    def __init__(self, resolution=0.2, key_frame_distance=1.0):
        lib().ref_lm_reset(float(resolution), float(key_frame_distance))

    def add_keyframe(self, world_xyzi, position):
        p, c = np.ascontiguousarray(world_xyzi, np.float32).reshape(-1, 4), np.ascontiguousarray(position, np.float32)
        return lib().ref_lm_add_keyframe(_p(p, C.c_float), len(p), _p(c, C.c_float))

    def update(self, pose_xyz):
        """0 nothing to do, 1 local map replaced, 2 out of map, 3 nearest key frame >= 20 m away (the codes of lio_localmap_update)"""
        x = np.ascontiguousarray(pose_xyz, np.float64)
        return lib().ref_lm_update(_p(x, C.c_double))

    def local_map(self, cap=400_000):
        out = np.zeros((cap, 4), np.float32)
        n = lib().ref_lm_local_map(_p(out, C.c_float), cap)
        return None if n < 0 else out[:n].copy()


def overlap_fitness(cloud1, cloud2, relpose, max_range):
    """OverlapDetector::calc_fitness_score(cloud1, cloud2, relpose (cast to Matrix4f), max_range) -> (score, inlier ratio)"""
    a, b = np.ascontiguousarray(cloud1, np.float32).reshape(-1, 4), np.ascontiguousarray(cloud2, np.float32).reshape(-1, 4)
    T = np.ascontiguousarray(relpose, np.float32).reshape(16)
    out = np.zeros(2)
    lib().ref_overlap_fitness(_p(a, C.c_float), len(a), _p(b, C.c_float), len(b), _p(T, C.c_float), float(max_range), _p(out, C.c_double))
    return float(out[0]), float(out[1])
