"""ctypes binding of oracle/liblio_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (lidar-slam-detection_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblio_oracle.so")

STATE_DIM = 26  # pos3 rot4(xyzw) R_il4 t_il3 vel3 bg3 ba3 grav3
DOF = 23
G_LEN = 9.809


MEAS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "lio_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "liblio_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    f32p, f64p, i32p, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint8))
    L.orc_voxel_downsample.argtypes = [f32p, C.c_int, C.c_float, f32p, C.c_int]
    L.orc_voxel_downsample.restype = C.c_int
    L.orc_esti_plane.argtypes = [f32p, C.c_float, f32p]
    L.orc_esti_plane.restype = C.c_int
    L.orc_ivox_create.argtypes = [C.c_float, C.c_int, C.c_uint64, C.c_double]
    L.orc_ivox_create.restype = C.c_void_p
    L.orc_ivox_destroy.argtypes = [C.c_void_p]
    L.orc_ivox_set_stencil.argtypes = [C.c_void_p, C.c_int]
    L.orc_ivox_set_tie_mode.argtypes = [C.c_void_p, C.c_int]
    L.orc_lio_set_tie_mode.argtypes = [C.c_void_p, C.c_int]
    L.orc_ivox_knn_as_reference.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.orc_ivox_add.argtypes = [C.c_void_p, f32p, C.c_int, C.c_double]
    L.orc_ivox_num_voxels.argtypes = [C.c_void_p]
    L.orc_ivox_num_voxels.restype = C.c_uint64
    L.orc_ivox_num_points.argtypes = [C.c_void_p]
    L.orc_ivox_dump.argtypes = [C.c_void_p, f32p, C.c_uint64]
    L.orc_ivox_dump.restype = C.c_int64
    L.orc_ivox_num_points.restype = C.c_uint64
    L.orc_ivox_knn.argtypes = [C.c_void_p, f32p, C.c_int, f32p, i32p, C.c_int]
    L.orc_ivox_knn.restype = C.c_uint64
    L.orc_ivox_stencil_points.argtypes = [C.c_void_p, f32p, C.c_int]
    L.orc_ivox_stencil_points.restype = C.c_uint64
    L.orc_lio_create.argtypes = [C.c_float, C.c_int, C.c_uint64, C.c_double, C.c_int]
    L.orc_lio_create.restype = C.c_void_p
    L.orc_lio_destroy.argtypes = [C.c_void_p]
    L.orc_lio_set_state.argtypes = [C.c_void_p, f64p]
    L.orc_lio_get_state.argtypes = [C.c_void_p, f64p]
    L.orc_lio_set_cov.argtypes = [C.c_void_p, f64p]
    L.orc_lio_get_cov.argtypes = [C.c_void_p, f64p]
    L.orc_lio_set_flags.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
    L.orc_lio_set_stencil.argtypes = [C.c_void_p, C.c_int]
    L.orc_lio_map_add.argtypes = [C.c_void_p, f32p, C.c_int, C.c_double]
    L.orc_lio_map_num_points.argtypes = [C.c_void_p]
    L.orc_lio_map_num_points.restype = C.c_uint64
    L.orc_lio_map_num_voxels.argtypes = [C.c_void_p]
    L.orc_lio_map_num_voxels.restype = C.c_uint64
    L.orc_lio_map_dump.argtypes = [C.c_void_p, f32p, C.c_uint64]
    L.orc_lio_map_dump.restype = C.c_int64
    L.orc_lio_set_ds.argtypes = [C.c_void_p, f32p, C.c_int]
    L.orc_lio_reset_cache.argtypes = [C.c_void_p]
    L.orc_lio_imu_enqueue.argtypes = [C.c_void_p, C.c_double, f64p, f64p]
    L.orc_lio_pcl_enqueue.argtypes = [C.c_void_p, f32p, C.POINTER(C.c_uint32), C.c_int, C.c_double]
    L.orc_lio_frontend_config.argtypes = [C.c_void_p, f64p, f64p, C.c_int, C.c_double, C.c_int]
    L.orc_lio_ins_enqueue.argtypes = [C.c_void_p, C.c_double, f64p]
    L.orc_lio_set_wheelspeed.argtypes = [C.c_void_p, C.c_int]
    L.orc_lio_frontend_main.argtypes = [C.c_void_p]
    L.orc_lio_frontend_main.restype = C.c_int
    L.orc_lio_predict.argtypes = [C.c_void_p, C.c_double, f64p, f64p]
    L.orc_lio_get_undistorted.argtypes = [C.c_void_p, f32p, C.c_int]
    L.orc_lio_get_undistorted.restype = C.c_int
    L.orc_lio_get_odometry.argtypes = [C.c_void_p, f64p, f64p]
    L.orc_lio_is_init.argtypes = [C.c_void_p]
    L.orc_lio_set_max_point_num.argtypes = [C.c_void_p, C.c_int]
    L.orc_undistort_delta.argtypes = [f32p, f32p, C.POINTER(C.c_uint32), C.c_int, C.c_double]
    L.orc_undistort_poses.argtypes = [C.POINTER(C.c_uint64), f64p, C.c_int, f32p, C.POINTER(C.c_uint32), C.c_int, C.c_uint64]
    L.orc_lio_get_ds_world.argtypes = [C.c_void_p, f32p, C.c_int]
    L.orc_kf_update_cb.argtypes = [f64p, f64p, C.c_double, C.c_int, MEAS_FN, C.c_void_p, C.c_int, f64p, f64p]
    L.orc_kf_update_ws_cb.argtypes = [f64p, f64p, C.c_double, C.c_int, MEAS_FN, C.c_void_p, C.c_int, f64p, C.c_int, f64p, f64p]
    L.orc_so3_Exp.argtypes = [f64p, C.c_double, f64p]
    L.orc_undistort_point.argtypes = [f64p, f64p, f64p, f64p, f64p, C.c_double, f32p, f64p, f64p, f64p, f64p, f32p]
    L.orc_lio_get_ds.argtypes = [C.c_void_p, f32p, C.c_int]
    L.orc_lio_get_ds.restype = C.c_int
    L.orc_lio_linearize.argtypes = [C.c_void_p, C.c_int, u8p, f32p, i32p, f32p, f64p, f64p, f64p, i32p]
    L.orc_lio_linearize.restype = C.c_int
    L.orc_lio_update.argtypes = [C.c_void_p]
    L.orc_lio_update.restype = C.c_int
    L.orc_lio_pass_log.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p, i32p, f64p, f64p, f64p, f64p]
    L.orc_lio_pass_log.restype = C.c_int
    L.orc_lio_map_incremental.argtypes = [C.c_void_p]
    L.orc_lio_map_incremental.restype = C.c_int
    L.orc_lio_process_scan.argtypes = [C.c_void_p, f32p, C.c_int, C.c_double]
    L.orc_lio_process_scan.restype = C.c_int
    L.orc_lio_travel.argtypes = [C.c_void_p]
    L.orc_lio_travel.restype = C.c_double
    L.orc_lio_is_degenerate.argtypes = [C.c_void_p]
    L.orc_lio_is_degenerate.restype = C.c_int
    L.orc_lio_last_degeneracy.argtypes = [C.c_void_p, f32p, f32p, f64p]
    L.orc_state_boxplus.argtypes = [f64p, f64p, f64p]
    L.orc_state_boxminus.argtypes = [f64p, f64p, f64p]
    L.orc_A_matrix.argtypes = [f64p, f64p]
    L.orc_S2_Bx.argtypes = [f64p, f64p]
    L.orc_S2_Nx_yy.argtypes = [f64p, f64p]
    L.orc_S2_Mx.argtypes = [f64p, f64p, f64p]
    L.orc_eig3.argtypes = [f64p, f64p, f64p]
    L.orc_inverse.argtypes = [f64p, C.c_int, f64p]
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def voxel_downsample(pts_xyzi, leaf):
    pts = _f32(pts_xyzi).reshape(-1, 4)
    out = np.empty_like(pts)
    m = lib().orc_voxel_downsample(_p(pts, C.c_float), len(pts), float(leaf), _p(out, C.c_float), len(pts))
    assert m >= 0
    return out[:m].copy()


def esti_plane(five_xyzi, thr=0.1):
    p = _f32(five_xyzi).reshape(5, 4)
    out = np.zeros(4, np.float32)
    ok = lib().orc_esti_plane(_p(p, C.c_float), float(thr), _p(out, C.c_float))
    return bool(ok), out


class IVox:
    def __init__(self, res=0.5, stencil=19, capacity=1 << 40, max_distance=100.0):
        self.h = lib().orc_ivox_create(res, stencil, capacity, max_distance)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ivox_destroy(self.h)
            self.h = None

    def set_stencil(self, s):
        lib().orc_ivox_set_stencil(self.h, s)

    def add(self, pts, travel=0.0):
        p = _f32(pts).reshape(-1, 4)
        lib().orc_ivox_add(self.h, _p(p, C.c_float), len(p), float(travel))

    @property
    def num_voxels(self):
        return int(lib().orc_ivox_num_voxels(self.h))

    @property
    def num_points(self):
        return int(lib().orc_ivox_num_points(self.h))

    def dump(self):
        n = self.num_points
        out = np.zeros((max(n, 1), 4), np.float32)
        got = lib().orc_ivox_dump(self.h, _p(out, C.c_float), n)
        assert got == n
        return out[:n]

    def knn(self, q, threads=8):
        q = _f32(q).reshape(-1, 4)
        out = np.zeros((len(q), 5, 4), np.float32)
        cnt = np.zeros(len(q), np.int32)
        visited = lib().orc_ivox_knn(self.h, _p(q, C.c_float), len(q), _p(out, C.c_float), _p(cnt, C.c_int), threads)
        return out, cnt, int(visited)

    def set_tie_mode(self, mode):
        """1 (default): the reference's own choice among candidates exactly as far as the fifth nearest; 0: the five smallest in (d2, x, y, z)"""
        lib().orc_ivox_set_tie_mode(self.h, int(mode))

    def knn_as_reference(self, q):
        """the list exactly as GetClosestPoint returns it (nearest first, the rest as std::nth_element leaves them)"""
        q = _f32(q).reshape(-1, 4)
        out = np.zeros((len(q), 5, 4), np.float32)
        cnt = np.zeros(len(q), np.int32)
        lib().orc_ivox_knn_as_reference(self.h, _p(q, C.c_float), len(q), _p(out, C.c_float), _p(cnt, C.c_int))
        return out, cnt

    def stencil_points(self, q):
        q = _f32(q).reshape(-1, 4)
        return int(lib().orc_ivox_stencil_points(self.h, _p(q, C.c_float), len(q)))


def default_state():
    s = np.zeros(STATE_DIM)
    s[6] = 1.0
    s[10] = 1.0
    s[23] = G_LEN
    return s


def init_cov():
    """IMU_init covariance (IMU_Processing.hpp:224-231)."""
    P = np.eye(23)
    for i in (6, 7, 8, 9, 10, 11):
        P[i, i] = 0.00001
    for i in (15, 16, 17):
        P[i, i] = 0.0001
    for i in (18, 19, 20):
        P[i, i] = 0.001
    P[21, 21] = P[22, 22] = 0.00001
    return P


class Lio:
    """The restated FastLIO engine (globals of laserMapping.cpp as one object)."""

    def __init__(self, res=0.5, stencil=75, capacity=100000, max_distance=100.0, threads=8):
        self.h = lib().orc_lio_create(res, stencil, capacity, max_distance, threads)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_lio_destroy(self.h)
            self.h = None

    def set_state(self, s):
        s = _f64(s)
        assert s.size == STATE_DIM
        lib().orc_lio_set_state(self.h, _p(s, C.c_double))

    def get_state(self):
        s = np.zeros(STATE_DIM)
        lib().orc_lio_get_state(self.h, _p(s, C.c_double))
        return s

    def set_cov(self, P):
        P = _f64(P).reshape(23, 23)
        lib().orc_lio_set_cov(self.h, _p(P, C.c_double))

    def get_cov(self):
        P = np.zeros((23, 23))
        lib().orc_lio_get_cov(self.h, _p(P, C.c_double))
        return P

    def set_flags(self, ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=0.0):
        lib().orc_lio_set_flags(self.h, int(ekf_inited), int(first_scan), travel, first_lidar_time)

    def set_stencil(self, s):
        lib().orc_lio_set_stencil(self.h, s)

    def set_tie_mode(self, mode):
        lib().orc_lio_set_tie_mode(self.h, int(mode))

    def map_add(self, pts, travel=0.0):
        p = _f32(pts).reshape(-1, 4)
        lib().orc_lio_map_add(self.h, _p(p, C.c_float), len(p), float(travel))

    @property
    def map_num_points(self):
        return int(lib().orc_lio_map_num_points(self.h))

    @property
    def map_num_voxels(self):
        return int(lib().orc_lio_map_num_voxels(self.h))

    def map_dump(self):
        n = self.map_num_points
        out = np.zeros((max(n, 1), 4), np.float32)
        m = lib().orc_lio_map_dump(self.h, _p(out, C.c_float), max(n, 1))
        assert m == n
        return out[:n]

    def set_ds(self, ds):
        d = _f32(ds).reshape(-1, 4)
        self._n = len(d)
        lib().orc_lio_set_ds(self.h, _p(d, C.c_float), len(d))

    def imu_enqueue(self, stamp, gyr, acc_ms2):
        g, a = _f64(gyr), _f64(acc_ms2)
        lib().orc_lio_imu_enqueue(self.h, float(stamp), _p(g, C.c_double), _p(a, C.c_double))

    def pcl_enqueue(self, xyzi, t_us, stamp):
        p, t = _f32(xyzi).reshape(-1, 4), np.ascontiguousarray(t_us, np.uint32)
        lib().orc_lio_pcl_enqueue(self.h, _p(p, C.c_float), _p(t, C.c_uint32), len(p), float(stamp))

    def frontend_config(self, extT=(0, 0, 0), extR_xyzw=(0, 0, 0, 1), filter_num=1, scan_period=0.1, undistort=True, max_point_num=-1):
        t, r = _f64(extT), _f64(extR_xyzw)
        lib().orc_lio_frontend_config(self.h, _p(t, C.c_double), _p(r, C.c_double), filter_num, float(scan_period), int(undistort))
        lib().orc_lio_set_max_point_num(self.h, int(max_point_num))

    def set_wheelspeed(self, on):
        """wheelspeed_en of laserMapping.cpp:83 (a constant false in the reference: its wheel-speed rows, :794-811, are dead code there)"""
        lib().orc_lio_set_wheelspeed(self.h, int(on))

    def ins_enqueue(self, stamp, vel_imu):
        v = _f64(vel_imu)
        lib().orc_lio_ins_enqueue(self.h, float(stamp), _p(v, C.c_double))

    def frontend_main(self):
        return lib().orc_lio_frontend_main(self.h)

    def predict(self, dt, acc, gyro):
        a, g = _f64(acc), _f64(gyro)
        lib().orc_lio_predict(self.h, float(dt), _p(a, C.c_double), _p(g, C.c_double))

    def get_undistorted(self, cap=300000):
        out = np.zeros((cap, 4), np.float32)
        n = lib().orc_lio_get_undistorted(self.h, _p(out, C.c_float), cap)
        assert n >= 0
        return out[:n].copy()

    def is_init(self):
        return bool(lib().orc_lio_is_init(self.h))

    def get_odometry(self):
        a, b = np.zeros(STATE_DIM), np.zeros(STATE_DIM)
        lib().orc_lio_get_odometry(self.h, _p(a, C.c_double), _p(b, C.c_double))
        return a, b

    def reset_cache(self):
        lib().orc_lio_reset_cache(self.h)

    def get_ds(self, cap=100000):
        out = np.zeros((cap, 4), np.float32)
        n = lib().orc_lio_get_ds(self.h, _p(out, C.c_float), cap)
        assert n >= 0
        self._n = n
        return out[:n].copy()

    def get_ds_world(self, cap=100000):
        out = np.zeros((cap, 4), np.float32)
        n = lib().orc_lio_get_ds_world(self.h, _p(out, C.c_float), cap)
        assert n >= 0
        return out[:n].copy()

    def linearize(self, converge=True):
        n = self._n
        sel = np.zeros(n, np.uint8)
        nv = np.zeros((n, 4), np.float32)
        cnt = np.zeros(n, np.int32)
        nn = np.zeros((n, 5, 4), np.float32)
        JtJ = np.zeros(36)
        Jtr = np.zeros(6)
        sres = C.c_double(0)
        deg = C.c_int(0)
        n_eff = lib().orc_lio_linearize(self.h, int(converge), _p(sel, C.c_uint8), _p(nv, C.c_float), _p(cnt, C.c_int),
                                        _p(nn, C.c_float), _p(JtJ, C.c_double), _p(Jtr, C.c_double), C.byref(sres), C.byref(deg))
        return dict(n_eff=n_eff, selected=sel, normvec=nv, nn_cnt=cnt, nn=nn, JtJ=JtJ.reshape(6, 6), Jtr=Jtr,
                    sum_abs_res=sres.value, degenerate=deg.value)

    def update(self):
        n = lib().orc_lio_update(self.h)
        logs = []
        for i in range(n):
            knn, ne, va, dg = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            sr = C.c_double()
            JtJ, Jtr, dx = np.zeros(36), np.zeros(6), np.zeros(23)
            lib().orc_lio_pass_log(self.h, i, C.byref(knn), C.byref(ne), C.byref(va), C.byref(dg), C.byref(sr),
                                   _p(JtJ, C.c_double), _p(Jtr, C.c_double), _p(dx, C.c_double))
            logs.append(dict(knn=knn.value, n_eff=ne.value, valid=va.value, degenerate=dg.value, sum_abs_res=sr.value,
                             JtJ=JtJ.reshape(6, 6), Jtr=Jtr, dx=dx))
        return logs

    def pass_logs(self):
        """the passes of the last update (also after frontend_main)"""
        logs, i = [], 0
        while True:
            knn, ne, va, dg = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            sr = C.c_double()
            JtJ, Jtr, dx = np.zeros(36), np.zeros(6), np.zeros(23)
            if lib().orc_lio_pass_log(self.h, i, C.byref(knn), C.byref(ne), C.byref(va), C.byref(dg), C.byref(sr), _p(JtJ, C.c_double), _p(Jtr, C.c_double), _p(dx, C.c_double)) < 0:
                return logs
            logs.append(dict(knn=knn.value, n_eff=ne.value, valid=va.value, degenerate=dg.value, sum_abs_res=sr.value, JtJ=JtJ.reshape(6, 6), Jtr=Jtr, dx=dx))
            i += 1

    def map_incremental(self):
        return lib().orc_lio_map_incremental(self.h)

    def process_scan(self, raw, lidar_beg_time):
        r = _f32(raw).reshape(-1, 4)
        return lib().orc_lio_process_scan(self.h, _p(r, C.c_float), len(r), float(lidar_beg_time))

    @property
    def travel(self):
        return lib().orc_lio_travel(self.h)

    @property
    def is_degenerate(self):
        return bool(lib().orc_lio_is_degenerate(self.h))

    def last_degeneracy(self):
        """contri / strong (f32 sums of laserMapping.cpp:946-964) and the eigenvalues of sum n n^T of the last measurement pass"""
        c, s, w = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3)
        lib().orc_lio_last_degeneracy(self.h, _p(c, C.c_float), _p(s, C.c_float), _p(w, C.c_double))
        return dict(contri=c, strong=s, eigval=w)


def undistort_delta(xyzi, stamp_us, delta_pose, scan_period=0.1):
    """undistortPoints(delta_pose, points, scan_period), slam_utils.cpp:163-191"""
    p = np.array(xyzi, np.float32).reshape(-1, 4).copy()
    st, d = np.ascontiguousarray(stamp_us, np.uint32), np.ascontiguousarray(delta_pose, np.float32).reshape(16)
    lib().orc_undistort_delta(_p(d, C.c_float), _p(p, C.c_float), _p(st, C.c_uint32), len(p), float(scan_period))
    return p


def undistort_poses(xyzi, stamp_us, header_us, pose_stamps_us, pose_T):
    """undistortPoints(poses, points), slam_utils.cpp:193-228"""
    p = np.array(xyzi, np.float32).reshape(-1, 4).copy()
    st = np.ascontiguousarray(stamp_us, np.uint32)
    ps, pt = np.ascontiguousarray(pose_stamps_us, np.uint64), np.ascontiguousarray(pose_T, np.float64).reshape(-1, 16)
    lib().orc_undistort_poses(_p(ps, C.c_uint64), _p(pt, C.c_double), len(ps), _p(p, C.c_float), _p(st, C.c_uint32), len(p), int(header_us))
    return p


def state_boxplus(s, d):
    s, d = _f64(s), _f64(d)
    o = np.zeros(STATE_DIM)
    lib().orc_state_boxplus(_p(s, C.c_double), _p(d, C.c_double), _p(o, C.c_double))
    return o


def state_boxminus(a, b):
    a, b = _f64(a), _f64(b)
    o = np.zeros(DOF)
    lib().orc_state_boxminus(_p(a, C.c_double), _p(b, C.c_double), _p(o, C.c_double))
    return o


def A_matrix(v):
    v = _f64(v)
    o = np.zeros(9)
    lib().orc_A_matrix(_p(v, C.c_double), _p(o, C.c_double))
    return o.reshape(3, 3)


def S2_Bx(v):
    v = _f64(v)
    o = np.zeros(6)
    lib().orc_S2_Bx(_p(v, C.c_double), _p(o, C.c_double))
    return o.reshape(3, 2)


def S2_Nx_yy(v):
    v = _f64(v)
    o = np.zeros(6)
    lib().orc_S2_Nx_yy(_p(v, C.c_double), _p(o, C.c_double))
    return o.reshape(2, 3)


def S2_Mx(v, d):
    v, d = _f64(v), _f64(d)
    o = np.zeros(6)
    lib().orc_S2_Mx(_p(v, C.c_double), _p(d, C.c_double), _p(o, C.c_double))
    return o.reshape(3, 2)


def eig3(A):
    A = _f64(A).reshape(3, 3)
    w, V = np.zeros(3), np.zeros(9)
    lib().orc_eig3(_p(A, C.c_double), _p(w, C.c_double), _p(V, C.c_double))
    return w, V.reshape(3, 3)


def inverse(A):
    A = _f64(A)
    n = A.shape[0]
    o = np.zeros_like(A)
    lib().orc_inverse(_p(A, C.c_double), n, _p(o, C.c_double))
    return o


def so3_Exp(w, dt):
    w = _f64(w)
    R = np.zeros(9)
    lib().orc_so3_Exp(_p(w, C.c_double), float(dt), _p(R, C.c_double))
    return R.reshape(3, 3)


def undistort_point(R_imu, vel, pos, acc, gyr, dt, p, end_pos, end_rot, ril, til):
    a = [_f64(v).ravel() for v in (R_imu, vel, pos, acc, gyr)]
    b = [_f64(v).ravel() for v in (end_pos, end_rot, ril, til)]
    pp = _f32(p)
    out = np.zeros(3, np.float32)
    lib().orc_undistort_point(*[_p(v, C.c_double) for v in a], float(dt), _p(pp, C.c_float), *[_p(v, C.c_double) for v in b], _p(out, C.c_float))
    return out


def kf_update_ws(s, P, R, meas_fn, ins_vel, degenerate=False, max_iter=4, cap=4096):
    """kf_update with the wheel-speed rows (laserMapping.cpp:794-811) appended to the model's rows in every pass"""
    s, P, v = _f64(s), _f64(P).reshape(-1), _f64(ins_vel)
    so, Po = np.zeros(STATE_DIM), np.zeros(529)
    lib().orc_kf_update_ws_cb(_p(s, C.c_double), _p(P, C.c_double), float(R), max_iter, meas_fn, None, cap, _p(v, C.c_double), int(degenerate),
                              _p(so, C.c_double), _p(Po, C.c_double))
    return so, Po.reshape(23, 23)


def kf_update(s, P, R, meas_fn, max_iter=4, cap=4096):
    """esekf::update_iterated_dyn_share_modified restated, with a caller-supplied measurement model (a MEAS_FN)"""
    s, P = _f64(s), _f64(P).reshape(-1)
    so, Po = np.zeros(STATE_DIM), np.zeros(529)
    lib().orc_kf_update_cb(_p(s, C.c_double), _p(P, C.c_double), float(R), max_iter, meas_fn, None, cap, _p(so, C.c_double), _p(Po, C.c_double))
    return so, Po.reshape(23, 23)
