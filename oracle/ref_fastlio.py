"""ctypes access to oracle/_ref/libref_fastlio.so: the reference's OWN FastLIO translation units (laserMapping.cpp,
IMU_Processing.hpp, preprocess.cpp, iVox, IKFoM) compiled whole from /root/reference by `make -C oracle ref`; only
pcl::VoxelGrid is routed to the oracle's restatement (oracle/ref_shims/pcl/filters/voxel_grid.h).  One global instance per
process, as in the reference.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_fastlio.so")  # parity build: scalar Eigen, no FMA contraction, kNN loop on one thread
_PATH_RELEASE = os.path.join(_HERE, "_ref", "libref_fastlio_release.so")  # the reference's own CMake flags: -O3 -DNDEBUG, MP_EN on 8 threads
_lib = None
_which = None


def available(release=False):
    return os.path.exists(_PATH_RELEASE if release else _PATH)


def use_release_build():
    """select the build with the reference's CMake flags (for timing); one build per process -- both define the same globals"""
    global _which
    assert _lib is None or _which == _PATH_RELEASE
    _which = _PATH_RELEASE


def lib():
    global _lib, _which
    if _lib is None:
        C.CDLL(os.path.join(_HERE, "liblio_oracle.so"), mode=C.RTLD_GLOBAL)  # orc_voxel_downsample for the VoxelGrid shim
        _which = _which or _PATH
        L = C.CDLL(_which)
        L.ref_fl_map_add.argtypes = [C.POINTER(C.c_float), C.c_int]
        L.ref_fl_map_dump.argtypes = [C.POINTER(C.c_float), C.c_int]
        L.ref_fl_register.argtypes = [C.POINTER(C.c_float), C.c_int] + [C.POINTER(C.c_double)] * 4
        f64p, f32p, u32p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.ref_fl_init.argtypes = [f64p, f64p, C.c_int, C.c_int, C.c_double, C.c_int]
        L.ref_fl_imu_enqueue.argtypes = [C.c_double, f64p, f64p]
        L.ref_fl_ins_enqueue.argtypes = [C.c_int, C.c_uint64] + [C.c_double] * 6 + [C.c_char_p]
        L.ref_fl_pcl_enqueue.argtypes = [f32p, u32p, C.c_int, C.c_uint64]
        L.ref_fl_odometry.argtypes = [f64p, f64p]
        L.ref_fl_state.argtypes = [f64p, f64p, f64p]
        L.ref_fl_fastlio_state.argtypes = [f64p]
        for n in ("ref_fl_undistorted", "ref_fl_down_body", "ref_fl_down_world", "ref_fl_last_preprocessed"):
            getattr(L, n).argtypes = [f32p, C.c_int]
        L.ref_fl_info.argtypes = [f64p]
        L.ref_fl_h_share.argtypes = [f64p, C.c_int, C.POINTER(C.c_uint8), f32p, f32p, C.POINTER(C.c_int), f64p, f64p, C.c_int]
        L.ref_fl_call.argtypes = [C.c_int, f64p, C.POINTER(C.c_int), f64p, f64p, f64p]
        _lib = L
    return _lib


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


class RefFastLio:
    """fastlio_init .. fastlio_odometry of laserMapping.cpp; stamps in integer microseconds as the reference's header"""

    def __init__(self, extT=(0, 0, 0), extR=np.eye(3), filter_num=1, max_point_num=-1, scan_period=0.1, undistort=True):
        t, r = np.ascontiguousarray(extT, np.float64), np.ascontiguousarray(extR, np.float64).reshape(9)
        lib().ref_fl_init(_p(t), _p(r), filter_num, max_point_num, float(scan_period), int(undistort))

    def imu_enqueue(self, stamp, gyr, acc_ms2):
        g, a = np.ascontiguousarray(gyr, np.float64), np.ascontiguousarray(acc_ms2, np.float64)
        lib().ref_fl_imu_enqueue(float(stamp), _p(a), _p(g))

    def set_wheelspeed(self, on):
        lib().ref_fl_set_wheelspeed(int(on))

    def ins_enqueue(self, rtk_valid, stamp_us, heading, pitch, roll, Ve, Vn, Vu, sensor="Wheel"):
        lib().ref_fl_ins_enqueue(int(rtk_valid), int(stamp_us), heading, pitch, roll, Ve, Vn, Vu, sensor.encode())

    def pcl_enqueue(self, xyzi, t_us, stamp_us):
        p, t = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4), np.ascontiguousarray(t_us, np.uint32)
        lib().ref_fl_pcl_enqueue(_p(p, C.c_float), _p(t, C.c_uint32), len(p), int(stamp_us))

    def main(self):
        return bool(lib().ref_fl_main())

    def is_init(self):
        return bool(lib().ref_fl_is_init())

    def state(self):
        s, s0, P = np.zeros(26), np.zeros(26), np.zeros(529)
        lib().ref_fl_state(_p(s), _p(s0), _p(P))
        return s, s0, P.reshape(23, 23)

    def get_state(self):
        return self.state()[0]

    def odometry(self):
        a, b = np.zeros(16), np.zeros(16)
        lib().ref_fl_odometry(_p(a), _p(b))
        return a.reshape(4, 4), b.reshape(4, 4)

    def fastlio_state(self):
        v = np.zeros(20)
        n = lib().ref_fl_fastlio_state(_p(v))
        return v[:n]

    def _cloud(self, fn, width, cap=400000):
        out = np.zeros((cap, width), np.float32)
        n = fn(_p(out, C.c_float), cap)
        return out[:n].copy()

    def undistorted(self):
        """x y z intensity curvature(ms) normal_xyz of feats_undistort"""
        return self._cloud(lib().ref_fl_undistorted, 8)

    def down_body(self):
        return self._cloud(lib().ref_fl_down_body, 4)

    def down_world(self):
        return self._cloud(lib().ref_fl_down_world, 4)

    def last_preprocessed(self):
        return self._cloud(lib().ref_fl_last_preprocessed, 8)

    def h_share(self, state26, converge=True):
        """one call of the reference's h_share_model on the current scan / map at a given state; converge False: reuse the neighbour
        lists, True: search, 2: search, then canonical neighbour order and a re-linearisation"""
        n = len(self.down_body())
        s = np.ascontiguousarray(state26, np.float64)
        sel, nv, nn, cnt = np.zeros(n, np.uint8), np.zeros((n, 4), np.float32), np.zeros((n, 5, 4), np.float32), np.zeros(n, np.int32)
        rows, h = np.zeros((max(n, 1), 12)), np.zeros(max(n, 1))
        m = lib().ref_fl_h_share(_p(s), int(converge), _p(sel, C.c_uint8), _p(nv, C.c_float), _p(nn, C.c_float), _p(cnt, C.c_int), _p(rows), _p(h), n)
        return dict(n_eff=m, selected=sel, normvec=nv, nn=nn, nn_cnt=cnt, rows=rows[:max(m, 0)], h=h[:max(m, 0)], degenerate=self.info()["degenerate"])

    def set_canonical(self, on=True):
        """canonical-order mode: every neighbour search of the filter's measurement model is followed by an ordering of the lists
        and a re-linearisation (see ref_fastlio.cpp h_share_logged)"""
        lib().ref_fl_set_canonical(int(on))

    def canonical_neighbours(self):
        lib().ref_fl_canonical_neighbours()

    def calls(self, clear=True):
        """what every h_share_model call of the filter saw (state, converge) and produced since the last clear"""
        out = []
        for i in range(lib().ref_fl_num_calls()):
            s, fl, tr, HtH, Hth = np.zeros(26), np.zeros(4, np.int32), np.zeros(1), np.zeros(36), np.zeros(6)
            lib().ref_fl_call(i, _p(s), _p(fl, C.c_int), _p(tr), _p(HtH), _p(Hth))
            out.append(dict(state=s, converge=bool(fl[0]), valid=bool(fl[1]), n_eff=int(fl[2]), degenerate=bool(fl[3]), total_residual=tr[0],
                            HtH=HtH.reshape(6, 6), Hth=Hth))
        if clear:
            lib().ref_fl_clear_calls()
        return out

    def set_logging(self, on):
        lib().ref_fl_set_logging(int(on))

    def map_add(self, xyzi):
        p = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        return lib().ref_fl_map_add(_p(p, C.c_float), len(p))

    def map_dump(self):
        """every map point, voxel by voxel, inside a voxel in push_back order (IVox::GetAllPoints)"""
        n = lib().ref_fl_map_dump(None, 0)
        out = np.zeros((max(n, 1), 4), np.float32)
        n = lib().ref_fl_map_dump(_p(out, C.c_float), len(out))
        return out[:n]

    def set_nearby(self, n):
        lib().ref_fl_set_nearby(int(n))

    def reset_cache(self):
        lib().ref_fl_reset_cache()

    def register(self, raw_xyzi, state26, P):
        """downsample + iterated filter update of one raw cloud against the current map from a given prior (no map insert)"""
        r = np.ascontiguousarray(raw_xyzi, np.float32).reshape(-1, 4)
        s, Pi = np.ascontiguousarray(state26, np.float64), np.ascontiguousarray(P, np.float64).reshape(-1)
        so, Po = np.zeros(26), np.zeros(529)
        rc = lib().ref_fl_register(_p(r, C.c_float), len(r), _p(s), _p(Pi), _p(so), _p(Po))
        return rc, so, Po.reshape(23, 23)

    def map_voxels(self):
        return lib().ref_fl_map_voxels()

    def info(self):
        v = np.zeros(8)
        lib().ref_fl_info(_p(v))
        return dict(effct_feat_num=int(v[0]), feats_down_size=int(v[1]), degenerate=bool(v[2]), travel_distance=v[3], ekf_inited=bool(v[4]),
                    nearby_type=int(v[5]), lidar_buffer=int(v[6]), imu_buffer=int(v[7]))
