"""Host-side objects over the C ABI: Map (iVox), Scan (per-scan buffers) and Engine (fastlio_main body).

Names and argument meaning follow the reference's FastLIO frontend
(/root/reference/slam/mapping/fastlio/src/laserMapping.cpp): `Engine.process_scan` is the part of
`fastlio_main()` that follows IMU processing, `Engine.update` is
`kf.update_iterated_dyn_share_modified(LASER_POINT_COV)`, `Map.add` is `ivox->AddPoints`,
`Map.knn` is `ivox->GetClosestPoint(p, out, 5, 5.0)`, `Scan.voxel_downsample` is `downSizeFilterSurf.filter`.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, f32, f64, lib, ptr

STATE_DIM = 26  # pos3 rot4(xyzw) R_il4 t_il3 vel3 bg3 ba3 grav3  (use-ikfom.hpp:12-21)
DOF = 23
G_LEN = 9.809


def default_state():
    s = np.zeros(STATE_DIM)
    s[6] = 1.0
    s[10] = 1.0
    s[23] = G_LEN
    return s


def init_cov():
    """covariance after IMU initialisation (IMU_Processing.hpp:224-231)"""
    P = np.eye(23)
    for i in (6, 7, 8, 9, 10, 11):
        P[i, i] = 0.00001
    for i in (15, 16, 17):
        P[i, i] = 0.0001
    for i in (18, 19, 20):
        P[i, i] = 0.001
    P[21, 21] = P[22, 22] = 0.00001
    return P


def _pose_ext(state):
    s = f64(state)
    pose = np.concatenate([s[0:3], s[3:7]])
    ext = np.concatenate([s[11:14], s[7:11]])
    return f64(pose), f64(ext)


class Map:
    def __init__(self, resolution=0.5, stencil=19, max_points=2_000_000, max_voxels=1_000_000, device=0, _borrow=None):
        self._own = _borrow is None
        self.h = _borrow if _borrow is not None else lib().lio_map_create(device, resolution, max_points, max_voxels, stencil)
        if not self.h:
            raise capi.LioError("lio_map_create failed: " + lib().lio_last_error().decode())

    def set_lru(self, capacity_voxels, max_distance=100.0):
        """IVox capacity_ / max_distance_: least-recently-touched voxels are dropped above `capacity_voxels` (call before the first add)"""
        check(lib().lio_map_set_lru(self.h, int(capacity_voxels), float(max_distance)), "set_lru")

    def lru_stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().lio_map_lru_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def lru_exact_stats(self):
        """(voxels dropped and re-created inside a batch as the reference's point-by-point order does, batches in which that order was not followed)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().lio_map_lru_exact_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def clear(self):
        """back to an empty map (memory kept)"""
        check(lib().lio_map_clear(self.h), "map clear")

    def set_tie_mode(self, mode):
        """1 (default): candidates exactly as far as the fifth nearest are kept as the reference's std::nth_element keeps them; 0: smallest (d2, x, y, z);
        2: the lists exactly as the reference returns them, order included (every query redone by the reference's selection: slow, a parity mode)"""
        check(lib().lio_map_set_tie_mode(self.h, int(mode)), "set_tie_mode")

    def tie_stats(self):
        """(queries whose neighbour set the reference's selection decided, those of them left unresolved)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().lio_map_tie_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pool_stats(self):
        top, cap = C.c_uint64(), C.c_uint64()
        check(lib().lio_map_pool_stats(self.h, C.byref(top), C.byref(cap)), "pool stats")
        return top.value, cap.value

    def close(self):
        if getattr(self, "h", None) and self._own and lib is not None:  # `lib` is gone at interpreter shutdown
            lib().lio_map_destroy(self.h)
        self.h = None

    __del__ = close

    def set_stencil(self, s):
        check(lib().lio_map_set_stencil(self.h, s), "set_stencil")

    def add(self, pts, travel=0.0):
        p = f32(pts).reshape(-1, 4)
        check(lib().lio_map_insert(self.h, ptr(p, C.c_float), len(p), float(travel)), "map insert")

    def add_device(self, dptr, n, travel=0.0):
        check(lib().lio_map_insert_device(self.h, C.c_void_p(dptr), n, float(travel)), "map insert (device)")

    def stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().lio_map_stats(self.h, C.byref(a), C.byref(b)), "map stats")
        return int(a.value), int(b.value)

    @property
    def num_points(self):
        return self.stats()[0]

    @property
    def num_voxels(self):
        return self.stats()[1]

    @property
    def nbytes(self):
        return int(lib().lio_map_bytes(self.h))

    @property
    def knn_candidates(self):
        return int(lib().lio_map_knn_candidates(self.h))

    @property
    def knn_unique(self):
        """distinct points per launch the kNN sweep loaded, summed over the counted launches (diagnostic kernel variant only)"""
        return int(lib().lio_map_knn_unique(self.h))

    @property
    def knn_touched(self):
        """points the kNN sweep loaded so far (counted by the diagnostic kernel variant only: Batch.enable_kernel_timing(2))"""
        return int(lib().lio_map_knn_touched(self.h))

    def dump(self):
        n = self.num_points
        out = np.zeros((max(n, 1), 4), np.float32)
        m = check(lib().lio_map_dump(self.h, ptr(out, C.c_float), max(n, 1)), "map dump")
        return out[:m]

    def knn(self, q):
        q = f32(q).reshape(-1, 4)
        out = np.zeros((len(q), 5, 4), np.float32)
        cnt = np.zeros(len(q), np.int32)
        check(lib().lio_map_knn(self.h, ptr(q, C.c_float), len(q), ptr(out, C.c_float), ptr(cnt, C.c_int32)), "map knn")
        return out, cnt


class Scan:
    def __init__(self, max_raw=262144, max_ds=100000, device=0, _borrow=None):
        self._own = _borrow is None
        self.max_ds = max_ds
        self.h = _borrow if _borrow is not None else lib().lio_scan_create(device, max_raw, max_ds)
        if not self.h:
            raise capi.LioError("lio_scan_create failed: " + lib().lio_last_error().decode())

    def close(self):
        if getattr(self, "h", None) and self._own and lib is not None:
            lib().lio_scan_destroy(self.h)
        self.h = None

    __del__ = close

    def reset(self):
        check(lib().lio_scan_reset(self.h), "scan reset")

    def upload(self, body_xyzi):
        p = f32(body_xyzi).reshape(-1, 4)
        check(lib().lio_scan_upload(self.h, ptr(p, C.c_float), len(p)), "scan upload")
        self._keep = p

    def set_device(self, dptr, n):
        check(lib().lio_scan_set_device(self.h, C.c_void_p(dptr), n), "scan set_device")

    def undistort_delta(self, stamp_us, delta_pose, scan_period=0.1, on_device=False):
        """undistortPoints(delta_pose, points, scan_period) of the localisation mode (slam_utils.cpp:163-191) on the uploaded cloud;
        stamp_us: host uint32 array, or a device pointer with on_device=True"""
        d = np.ascontiguousarray(delta_pose, np.float32).reshape(16)
        if on_device:
            sp = C.c_void_p(int(stamp_us))
        else:
            self._stamps = np.ascontiguousarray(stamp_us, np.uint32)
            sp = C.c_void_p(self._stamps.ctypes.data)
        check(lib().lio_scan_undistort_delta(self.h, sp, int(on_device), d.ctypes.data_as(C.POINTER(C.c_float)), float(scan_period)), "undistort_delta")

    def undistort_poses(self, stamp_us, header_stamp_us, pose_stamps_us, pose_T):
        """undistortPoints(poses, points) of slam_utils.cpp:193-228 on the uploaded cloud (host stamps)"""
        self._stamps = np.ascontiguousarray(stamp_us, np.uint32)
        ps, pt = np.ascontiguousarray(pose_stamps_us, np.uint64), np.ascontiguousarray(pose_T, np.float64).reshape(-1, 16)
        check(lib().lio_scan_undistort_poses(self.h, C.c_void_p(self._stamps.ctypes.data), 0, int(header_stamp_us), ps.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             pt.ctypes.data_as(C.POINTER(C.c_double)), len(ps)), "undistort_poses")

    def download_raw(self, cap=1 << 18):
        out = np.zeros((cap, 4), np.float32)
        n = lib().lio_scan_download_raw(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), cap)
        if n < 0:
            check(n, "download_raw")
        return out[:n].copy()

    def voxel_downsample(self, leaf=0.5, sync=True):
        n = C.c_uint32(0)
        check(lib().lio_scan_voxel_downsample(self.h, float(leaf), int(sync), C.byref(n)), "voxel downsample")
        return int(n.value)

    @staticmethod
    def voxel_downsample_batch(scans, leaf=0.5):
        """lio_scan_voxel_downsample_batch: the VoxelGrid of many scans with one set of launches; returns the point counts"""
        hs = (C.c_void_p * len(scans))(*[s.h for s in scans])
        n = (C.c_uint32 * len(scans))()
        check(lib().lio_scan_voxel_downsample_batch(hs, len(scans), float(leaf), n), "voxel downsample batch")
        return [int(v) for v in n]

    def set_ds(self, ds):
        d = f32(ds).reshape(-1, 4)
        check(lib().lio_scan_set_ds(self.h, ptr(d, C.c_float), len(d)), "set_ds")

    @property
    def num_ds(self):
        return check(lib().lio_scan_num_ds(self.h), "num_ds")

    def get_ds(self):
        out = np.zeros((self.max_ds, 4), np.float32)
        n = check(lib().lio_scan_download_ds(self.h, ptr(out, C.c_float), self.max_ds), "download ds")
        return out[:n].copy()

    def get_world(self):
        out = np.zeros((self.max_ds, 4), np.float32)
        n = check(lib().lio_scan_download_world(self.h, ptr(out, C.c_float), self.max_ds), "download world")
        return out[:n].copy()

    def set_degeneracy_mode(self, mode):
        """0 auto (bound-gated), 1 always evaluate, 2 never"""
        check(lib().lio_scan_set_degeneracy_mode(self.h, int(mode)))

    def degeneracy(self, V):
        V = f64(V).reshape(3, 3)
        c, s_ = np.zeros(3), np.zeros(3)
        check(lib().lio_p2plane_degeneracy(self.h, ptr(V, C.c_double), ptr(c, C.c_double), ptr(s_, C.c_double)), "degeneracy")
        return c, s_

    def enable_kernel_timing(self, mask=3):
        """bit mask: 1 = kNN kernel, 2 = linearize kernel, 0 = off"""
        check(lib().lio_scan_enable_kernel_timing(self.h, int(mask)))

    def kernel_times(self, reset=True):
        t = capi.KernelTimes()
        check(lib().lio_scan_kernel_times(self.h, C.byref(t), int(reset)))
        return {k: getattr(t, k) for k, _ in t._fields_ if k != "pad"}

    def get_match(self):
        n = self.num_ds
        sel = np.zeros(n, np.uint8)
        nv = np.zeros((n, 4), np.float32)
        cnt = np.zeros(n, np.int32)
        nn = np.zeros((n, 5, 4), np.float32)
        check(lib().lio_scan_download_match(self.h, ptr(sel, C.c_uint8), ptr(nv, C.c_float), ptr(cnt, C.c_int32), ptr(nn, C.c_float)),
              "download match")
        return dict(selected=sel, normvec=nv, nn_cnt=cnt, nn=nn)


def linearize(map_, scan, state, redo_knn=True):
    """one evaluation of h_share_model_geometric at `state` (26 doubles); returns the normal equations"""
    pose, ext = _pose_ext(state)
    ne = capi.NormalEq()
    check(lib().lio_p2plane_linearize(map_.h, scan.h, ptr(pose, C.c_double), ptr(ext, C.c_double), int(redo_knn), C.byref(ne)), "linearize")
    return dict(n_eff=int(ne.n_eff), n_ds=int(ne.n_ds), JtJ=np.array(ne.JtJ).reshape(6, 6), Jtr=np.array(ne.Jtr),
                nnT=np.array(ne.nnT).reshape(3, 3), eigvec=np.array(ne.eigvec).reshape(3, 3), eigval=np.array(ne.eigval),
                contri=np.array(ne.contri), strong=np.array(ne.strong), sum_abs_res=float(ne.sum_abs_res), n_tie=int(ne.n_tie),
                knn_candidates=(int(ne.n_knn_candidates_hi) << 32) | int(ne.n_knn_candidates_lo))


def map_incremental(map_, scan, state, map_leaf=0.5, ekf_inited=True, travel=0.0):
    pose, ext = _pose_ext(state)
    return check(lib().lio_map_incremental(map_.h, scan.h, ptr(pose, C.c_double), ptr(ext, C.c_double), float(map_leaf), int(ekf_inited),
                                           float(travel)), "map_incremental")


class Engine:
    """The FastLIO per-scan engine (state + covariance + map + scan buffers) on one GPU."""

    def __init__(self, resolution=0.5, stencil=75, max_points=2_000_000, max_voxels=1_000_000, max_raw=262144, max_ds=100000, device=0,
                 shared_map=None):
        if shared_map is not None:  # read-only engine on somebody else's map (several may run concurrently)
            self._shared = shared_map
            self.h = lib().lio_engine_create_shared(shared_map.h, max_raw, max_ds)
        else:
            self.h = lib().lio_engine_create(device, resolution, stencil, max_points, max_voxels, max_raw, max_ds)
        if not self.h:
            raise capi.LioError("lio_engine_create failed: " + lib().lio_last_error().decode())
        self.map = Map(_borrow=lib().lio_engine_map(self.h))
        self.scan = Scan(max_ds=max_ds, _borrow=lib().lio_engine_scan(self.h))

    def close(self):
        if getattr(self, "h", None) and getattr(self, "_own", True) and lib is not None:
            lib().lio_engine_destroy(self.h)
        self.h = None

    __del__ = close

    def set_device_loop(self, on=True):
        """run the iterate loop of this engine's updates on the device (one submission, one wait) instead of from the host"""
        check(lib().lio_engine_set_device_loop(self.h, int(on)), "set_device_loop")

    def set_state(self, s):
        s = f64(s)
        assert s.size == STATE_DIM
        check(lib().lio_engine_set_state(self.h, ptr(s, C.c_double)))

    def get_state(self):
        s = np.zeros(STATE_DIM)
        check(lib().lio_engine_get_state(self.h, ptr(s, C.c_double)))
        return s

    def set_cov(self, P):
        P = f64(P).reshape(23, 23)
        check(lib().lio_engine_set_cov(self.h, ptr(P, C.c_double)))

    def get_cov(self):
        P = np.zeros((23, 23))
        check(lib().lio_engine_get_cov(self.h, ptr(P, C.c_double)))
        return P

    def set_flags(self, ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=0.0):
        check(lib().lio_engine_set_flags(self.h, int(ekf_inited), int(first_scan), float(travel), float(first_lidar_time)))

    def set_stencil(self, s):
        self.map.set_stencil(s)

    def map_add(self, pts, travel=0.0):
        self.map.add(pts, travel)

    def set_ds(self, ds):
        self.scan.set_ds(ds)

    def get_ds(self):
        return self.scan.get_ds()

    def update(self):
        n = check(lib().lio_engine_update(self.h), "engine update")
        logs = []
        for i in range(n):
            pl = capi.PassLog()
            check(lib().lio_engine_pass_log(self.h, i, C.byref(pl)))
            logs.append(dict(knn=pl.knn, n_eff=pl.n_eff, valid=pl.valid, degenerate=pl.degenerate, sum_abs_res=pl.sum_abs_res,
                             JtJ=np.array(pl.JtJ).reshape(6, 6), Jtr=np.array(pl.Jtr), dx=np.array(pl.dx)))
        return logs

    def map_incremental(self, map_leaf=0.5, ekf_inited=True):
        return map_incremental(self.map, self.scan, self.get_state(), map_leaf, ekf_inited, self.travel)

    def process_scan(self, raw, lidar_beg_time):
        r = f32(raw).reshape(-1, 4)
        return check(lib().lio_engine_process_scan(self.h, ptr(r, C.c_float), len(r), float(lidar_beg_time)), "process_scan")

    def process_scan_device(self, dptr, n, lidar_beg_time):
        return check(lib().lio_engine_process_scan_device(self.h, C.c_void_p(dptr), n, float(lidar_beg_time)), "process_scan")

    def enable_timing(self, on=True):
        check(lib().lio_engine_enable_timing(self.h, int(on)))

    # ---- the IMU front half: fastlio_init / _imu_enqueue / _pcl_enqueue / _main / _odometry / _state
    # (reference: slam/mapping/fastlio/src/laserMapping.cpp:1025,397,311,1126,692,714) ----
    def fastlio_init(self, extT=(0.0, 0.0, 0.0), extR=np.eye(3), filter_num=1, max_point_num=-1, scan_period=0.1, undistort=True):
        t, r = f64(extT), f64(extR).reshape(-1)
        assert t.size == 3 and r.size == 9
        check(lib().lio_fastlio_init(self.h, ptr(t, C.c_double), ptr(r, C.c_double), filter_num, max_point_num, float(scan_period), int(undistort)))

    def fastlio_is_init(self):
        return bool(lib().lio_fastlio_is_init(self.h))

    def fastlio_imu_enqueue(self, stamp, gyr, acc_ms2):
        g, a = f64(gyr), f64(acc_ms2)
        check(lib().lio_fastlio_imu_enqueue(self.h, float(stamp), ptr(g, C.c_double), ptr(a, C.c_double)))

    def fastlio_set_wheelspeed(self, on=True):
        """wheelspeed_en of laserMapping.cpp:83: append the wheel-speed rows (:794-811) when the scan's last INS sample is within 10 ms of its end"""
        check(lib().lio_fastlio_set_wheelspeed(self.h, int(on)), "set_wheelspeed")

    def fastlio_ins_enqueue(self, stamp, vel_imu):
        v = f64(vel_imu)
        check(lib().lio_fastlio_ins_enqueue(self.h, float(stamp), ptr(v, C.c_double)))

    def fastlio_pcl_enqueue(self, xyzi, stamp_us, header_stamp):
        p = f32(xyzi).reshape(-1, 4)
        t = np.ascontiguousarray(stamp_us, np.uint32)
        assert len(t) == len(p)
        check(lib().lio_fastlio_pcl_enqueue(self.h, ptr(p, C.c_float), ptr(t, C.c_uint32), len(p), float(header_stamp)))

    def fastlio_pcl_enqueue_device(self, d_xyzi, d_stamp_us, n, header_stamp):
        check(lib().lio_fastlio_pcl_enqueue_device(self.h, C.c_void_p(d_xyzi), C.c_void_p(d_stamp_us), n, float(header_stamp)))

    def fastlio_main(self):
        """one fastlio_main pass; returns capi MAIN_* (IDLE when there was nothing to do)"""
        return check(lib().lio_fastlio_main(self.h), "fastlio_main")

    def fastlio_odometry(self):
        a, b = np.zeros(16), np.zeros(16)
        check(lib().lio_fastlio_odometry(self.h, ptr(a, C.c_double), ptr(b, C.c_double)))
        return a.reshape(4, 4), b.reshape(4, 4)

    def fastlio_state(self):
        s = np.zeros(20)
        check(lib().lio_fastlio_state(self.h, ptr(s, C.c_double)))
        return s

    def fastlio_start_state(self):
        s = np.zeros(STATE_DIM)
        check(lib().lio_fastlio_start_state(self.h, ptr(s, C.c_double)))
        return s

    def undistorted(self, cap=300000):
        out = np.zeros((cap, 4), np.float32)
        n = C.c_uint32(0)
        check(lib().lio_fastlio_download_undistorted(self.h, ptr(out, C.c_float), cap, C.byref(n)))
        return out[:n.value].copy()


    def set_reduce_hook(self, fn):
        """fn(buf: np.ndarray) replaces the engine's local normal-equation sums by the global ones IN PLACE
        (see lsd_amd.dist.NormalEqAllGather); None removes the hook"""
        if fn is None:
            self._hook = None
            check(lib().lio_engine_set_reduce_hook(self.h, capi.REDUCE_FN(0), None))
            return

        def _cb(_ctx, p, n):
            fn(np.ctypeslib.as_array(p, shape=(n,)))

        self._hook = capi.REDUCE_FN(_cb)  # keep the trampoline alive
        check(lib().lio_engine_set_reduce_hook(self.h, self._hook, None))

    def set_joint(self, others=(), comm=None):
        """joint registration natively (lio_engine_set_joint): this engine drives the filter, `others` are engines holding further sub-maps on
        this GPU, `comm` (lio.Comm) joins the ranks that hold the rest"""
        self._joint_keep = (list(others), comm)
        hs = (C.c_void_p * max(len(others), 1))(*[o.h for o in others])
        check(lib().lio_engine_set_joint(self.h, hs, len(others), comm.h if comm is not None else None), "set_joint")

    def joint_register(self, raw, lidar_beg_time, state, cov):
        r = f32(raw).reshape(-1, 4)
        s, P = f64(state).copy(), f64(cov).reshape(-1).copy()
        rc = check(lib().lio_engine_joint_register(self.h, ptr(r, C.c_float), len(r), float(lidar_beg_time), ptr(s, C.c_double), ptr(P, C.c_double)),
                   "joint_register")
        return rc, s, P.reshape(23, 23)

    def joint_register_device(self, dptr, n, lidar_beg_time, state, cov):
        """lio_engine_joint_register_device: the cloud is already resident on the GPU"""
        s, P = f64(state).copy(), f64(cov).reshape(-1).copy()
        rc = check(lib().lio_engine_joint_register_device(self.h, C.c_void_p(dptr), n, float(lidar_beg_time), ptr(s, C.c_double), ptr(P, C.c_double)),
                   "joint_register_device")
        return rc, s, P.reshape(23, 23)

    def set_static_map(self, on=True):
        check(lib().lio_engine_set_static_map(self.h, int(on)))

    def flush(self):
        """lio_engine_flush: wait for the map_incremental the last scan enqueued; raises if the map overflowed"""
        check(lib().lio_engine_flush(self.h), "flush")

    def timings(self):
        t = capi.Timings()
        check(lib().lio_engine_timings(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    @property
    def travel(self):
        return lib().lio_engine_travel(self.h)

    @property
    def is_degenerate(self):
        return bool(lib().lio_engine_is_degenerate(self.h))


def overlap_filter(cloud, xy_range=100.0, min_z=0.5):
    """`filter` of overlap_merge.hpp:213-223: keep sqrt(x^2 + y^2) < range && z > floor (f32)"""
    c = np.ascontiguousarray(cloud, np.float32).reshape(-1, 4)
    d = np.sqrt(c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1])
    return c[(d < np.float32(xy_range)) & (c[:, 2] > np.float32(min_z))]


class Ndt:
    """The localization matcher (fast_gicp::NDTCuda, P2D): set_target = setInputTarget, the source is a Scan
    (upload + voxel_downsample = setInputSource), align = pcl::Registration::align(guess)."""

    def __init__(self, resolution=1.0, search_method=7, max_points=1_000_000, max_voxels=500_000, max_source_points=100_000, device=0):
        self.h = lib().lio_ndt_create(device, resolution, search_method, max_points, max_voxels, max_source_points)
        if not self.h:
            raise capi.LioError("lio_ndt_create failed: " + lib().lio_last_error().decode())

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib().lio_ndt_destroy(self.h)
        self.h = None

    __del__ = close

    def enable_kernel_timing(self, on=True):
        check(lib().lio_ndt_enable_kernel_timing(self.h, int(on)))

    def kernel_times(self, reset=True):
        t = capi.NdtTimes()
        check(lib().lio_ndt_kernel_times(self.h, C.byref(t), int(reset)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def set_target(self, pts):
        p = f32(pts).reshape(-1, 4)
        check(lib().lio_ndt_set_target(self.h, ptr(p, C.c_float), len(p)), "ndt set_target")

    def set_target_device(self, dptr, n):
        check(lib().lio_ndt_set_target_device(self.h, C.c_void_p(dptr), n), "ndt set_target (device)")

    @property
    def num_voxels(self):
        return check(lib().lio_ndt_num_voxels(self.h), "ndt num_voxels")

    def voxel_at(self, p):
        p = f32(p)
        mean, cinv = np.zeros(3, np.float32), np.zeros(9, np.float32)
        n = check(lib().lio_ndt_voxel_at(self.h, ptr(p, C.c_float), ptr(mean, C.c_float), ptr(cinv, C.c_float)), "ndt voxel_at")
        return n, mean, cinv.reshape(3, 3)

    def linearize(self, scan, T, update_corr=True, with_derivatives=True):
        T = f64(T).reshape(4, 4)
        H, b, e, nc = np.zeros(36), np.zeros(6), C.c_double(0), C.c_uint32(0)
        check(lib().lio_ndt_linearize(self.h, scan.h, ptr(T, C.c_double), int(update_corr), int(with_derivatives), ptr(H, C.c_double),
                                      ptr(b, C.c_double), C.byref(e), C.byref(nc)), "ndt linearize")
        return dict(n_corr=int(nc.value), H=H.reshape(6, 6), b=b, err=e.value)

    def fitness_score(self, scan, T, max_range=25.0):
        """pcl getFitnessScore(max_range): (mean squared nearest-neighbour distance, points within range)"""
        t = f64(T).reshape(4, 4)
        sc, ni = C.c_double(0.0), C.c_uint32(0)
        check(lib().lio_ndt_fitness_score(self.h, scan.h, ptr(t, C.c_double), float(max_range), C.byref(sc), C.byref(ni)), "fitness score")
        return sc.value, ni.value

    def overlap_score(self, scan, relpose, max_range=1.0, xy_range=100.0, min_z=0.5):
        """calc_fitness_score of overlap_merge.hpp:225-263 against this target (filter the target cloud with `overlap_filter` first):
        (mean squared nearest-neighbour distance of the inliers, inlier share of the filtered source)"""
        t = f64(relpose).reshape(4, 4)
        sc, ir = C.c_double(0.0), C.c_double(0.0)
        check(lib().lio_ndt_overlap_score(self.h, scan.h, ptr(t, C.c_double), float(max_range), float(xy_range), float(min_z), C.byref(sc), C.byref(ir)),
              "overlap score")
        return sc.value, ir.value

    def align(self, scan, guess, **params):
        g = f64(guess).reshape(4, 4)
        prm = capi.NdtParams()
        lib().lio_ndt_default_params(C.byref(prm))
        for k, v in params.items():
            setattr(prm, k, v)
        out = np.zeros((4, 4))
        it, conv = C.c_int(0), C.c_int(0)
        check(lib().lio_ndt_align(self.h, scan.h, ptr(g, C.c_double), C.byref(prm), ptr(out, C.c_double), C.byref(it), C.byref(conv)), "ndt align")
        return out, bool(conv.value), int(it.value)

    def prepare_batch(self, scans, guesses, targets=None):
        """the job array of lio_ndt_align_batch, marshalled once (so that a timed region can be the one C call); targets: one lio.Ndt per job
        (None = this one)"""
        n = len(scans)
        arr = (capi.AlignJob * n)()
        keep = [f64(g).reshape(16).copy() for g in guesses]
        for i in range(n):
            arr[i].target = targets[i].h if targets is not None and targets[i] is not None else None
            arr[i].source = scans[i].h
            arr[i].guess = ptr(keep[i], C.c_double)
        return arr, keep

    def run_batch(self, prepared, **params):
        arr, _ = prepared
        prm = capi.NdtParams()
        lib().lio_ndt_default_params(C.byref(prm))
        for k, v in params.items():
            setattr(prm, k, v)
        return lib().lio_ndt_align_batch(self.h, arr, len(arr), C.byref(prm))

    def align_batch(self, scans, guesses, targets=None, **params):
        """B alignments per launch against this target (lio_ndt_align_batch): scans = lio.Scan objects holding their downsampled clouds;
        returns a list of (T 4 x 4, converged, iterations, evaluations, rc)"""
        prep = self.prepare_batch(scans, guesses, targets)
        check(self.run_batch(prep, **params), "ndt align_batch")
        return [(np.array(a.out).reshape(4, 4), bool(a.converged), int(a.iterations), int(a.evaluations), int(a.rc)) for a in prep[0]]


class Gicp:
    """fast_gicp::FastGICP on the device (lio_gicp_*): 20-NN covariances with PLANE regularisation, nearest-neighbour correspondences,
    (C_B + R C_A R^T)^-1 cost, LM on SE(3)"""

    def __init__(self, grid_resolution=1.0, max_points=200_000, k=20, device=0):
        self.max_points = max_points
        self.h = lib().lio_gicp_create(device, float(grid_resolution), int(max_points), int(k))
        if not self.h:
            raise capi.LioError("lio_gicp_create failed: " + lib().lio_last_error().decode())

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib().lio_gicp_destroy(self.h)
        self.h = None

    __del__ = close

    def set_target(self, pts):
        p = f32(pts).reshape(-1, 4)
        check(lib().lio_gicp_set_target(self.h, ptr(p, C.c_float), len(p)), "gicp set_target")

    def set_source(self, pts):
        p = f32(pts).reshape(-1, 4)
        check(lib().lio_gicp_set_source(self.h, ptr(p, C.c_float), len(p)), "gicp set_source")

    def set_voxel_mode(self, voxel_resolution=1.0, search_method=1):
        """fast_gicp::FastVGICP: Gaussian-voxel target of `voxel_resolution` (0 = back to the kd-tree form), DIRECT1 / 7 / 27 lookup"""
        check(lib().lio_gicp_set_voxel_mode(self.h, float(voxel_resolution), int(search_method)), "gicp set_voxel_mode")

    def voxel_at(self, p):
        """(points in the target voxel holding p, mean (3,), covariance (3, 3)); 0 points = no such voxel"""
        p = f32(p)
        m, c = np.zeros(3), np.zeros(6)
        n = check(lib().lio_gicp_voxel_at(self.h, ptr(p, C.c_float), ptr(m, C.c_double), ptr(c, C.c_double)), "gicp voxel_at")
        return n, m, np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]])

    def download(self, which):
        """(points (n, 4) in internal order, regularised covariances (n, 3, 3))"""
        pts, cov = np.zeros((self.max_points, 4), np.float32), np.zeros((self.max_points, 6))
        n = check(lib().lio_gicp_download(self.h, int(which), ptr(pts, C.c_float), ptr(cov, C.c_double), self.max_points), "gicp download")
        c = cov[:n]
        full = np.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(n, 3, 3)
        return pts[:n].copy(), full

    def correspondences(self):
        out = np.zeros(self.max_points, np.int32)
        n = check(lib().lio_gicp_correspondences(self.h, ptr(out, C.c_int32), self.max_points), "gicp correspondences")
        return out[:n].copy()

    def linearize(self, T, max_corr_dist=2.0, update_corr=True, with_derivatives=True):
        T = f64(T).reshape(4, 4)
        H, b, err, nc = np.zeros((6, 6)), np.zeros(6), C.c_double(0), C.c_uint32(0)
        check(lib().lio_gicp_linearize(self.h, ptr(T, C.c_double), float(max_corr_dist), int(update_corr), int(with_derivatives), ptr(H, C.c_double),
                                       ptr(b, C.c_double), C.byref(err), C.byref(nc)), "gicp linearize")
        return dict(H=H, b=b, err=err.value, n_corr=nc.value)

    def align(self, guess, max_corr_dist=2.0, **params):
        g = f64(guess).reshape(4, 4)
        prm = capi.NdtParams()
        lib().lio_ndt_default_params(C.byref(prm))
        prm.rotation_epsilon_deg, prm.transformation_epsilon, prm.max_iterations, prm.max_process_time_ms = 1e-2, 0.01, 64, -1.0
        for k, v in params.items():
            setattr(prm, k, v)
        out = np.zeros((4, 4))
        it, conv = C.c_int(0), C.c_int(0)
        check(lib().lio_gicp_align(self.h, ptr(g, C.c_double), C.byref(prm), float(max_corr_dist), ptr(out, C.c_double), C.byref(it), C.byref(conv)), "gicp align")
        return out, bool(conv.value), int(it.value)


def transform_cloud_f32(cloud, M):
    """pcl::transformPointCloud(in, out, M) for XYZ(I) points as PCL 1.9.1 evaluates it in f32: xyz = M(0..2, 0..3) * [x y z 1], left to right"""
    c = np.ascontiguousarray(cloud, np.float32).reshape(-1, 4).copy()
    Mf = np.asarray(M, np.float64).astype(np.float32)
    x, y, z = c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy()
    for r in range(3):
        c[:, r] = ((Mf[r, 0] * x + Mf[r, 1] * y) + Mf[r, 2] * z) + Mf[r, 3]
    return c


def transform_cloud_f64(cloud, M):
    """pcl::transformPointCloud(in, out, M) with an Eigen::Matrix4d (overlap_merge.hpp:190-194: the connected frames are moved by the f64 `relative`):
    PCL 1.9.1 evaluates that in double, terms left to right, and casts the result to float (csrc/slam_wrapper.cpp does the same for the static transform)"""
    c = np.ascontiguousarray(cloud, np.float32).reshape(-1, 4).copy()
    Md = np.asarray(M, np.float64)
    x, y, z = c[:, 0].astype(np.float64), c[:, 1].astype(np.float64), c[:, 2].astype(np.float64)
    for r in range(3):
        c[:, r] = (((Md[r, 0] * x + Md[r, 1] * y) + Md[r, 2] * z) + Md[r, 3]).astype(np.float32)
    return c


class OverlapMatcher:
    """OverlapDetector::matching of the map-merge / relocalisation tools (slam/localization/include/overlap_merge.hpp:151-211) on the
    device: per candidate the overlap pre-check and the coarse NDT alignment with its fitness score, then the fine GICP alignment of the
    new key frame against the best candidate accumulated with its connected frames, then the final fitness test.  Same constants
    (constructor :45-59): coarse = select_registration_method("NDT_CUDA"), fine = FAST_GICP with max correspondence distance 0.5 and
    transformation epsilon 0.001."""

    def __init__(self, max_points=400_000, device=0, ndt_resolution=1.0, gicp_grid=1.0):
        self.fitness_score_max_range = 25.0
        self.fitness_score_thresh = 1.5
        self.fitness_inlier_thresh = 0.2
        self.max_points = max_points
        self.ndt = Ndt(resolution=ndt_resolution, search_method=7, max_points=max_points, max_voxels=max_points, max_source_points=max_points, device=device)
        self.gicp = Gicp(grid_resolution=gicp_grid, max_points=max_points, k=20, device=device)
        self.scan = Scan(max_raw=max_points, max_ds=max_points, device=device)

    def close(self):
        for o in (self.ndt, self.gicp, self.scan):
            o.close()

    def calc_fitness_score(self, cloud1, cloud2, relpose, max_range):
        """calc_fitness_score(cloud1, cloud2, relpose, max_range) (:214-263) -> (score, inlier ratio)"""
        self.ndt.set_target(overlap_filter(cloud1))
        self.scan.set_ds(cloud2)
        return self.ndt.overlap_score(self.scan, np.asarray(relpose, np.float32).astype(np.float64), max_range)

    def matching(self, new_points, new_odom, candidates, connected=()):
        """new_points / new_odom: the new key frame (cloud, 4 x 4 odometry); candidates: [(points, odom), ...]; connected: frames linked
        to the candidates, [(candidate index, points, odom), ...].  Returns None (no overlap) or dict(best, relative_pose (4 x 4 f32, new
        frame -> accumulated candidate frame ... as the reference returns it), score, coarse=[...])"""
        new_points = f32(new_points).reshape(-1, 4)
        new_odom = f64(new_odom).reshape(4, 4)
        best_score, best, rel, coarse = np.inf, None, None, []
        for ci, (pts, odom) in enumerate(candidates):
            guess = (np.linalg.inv(new_odom) @ f64(odom).reshape(4, 4)).astype(np.float32)
            fit = self.calc_fitness_score(new_points, pts, guess, 1.0)
            if fit[1] < self.fitness_inlier_thresh:
                coarse.append(dict(candidate=ci, skipped="inlier ratio", ratio=fit[1]))
                continue
            self.ndt.set_target(new_points)
            self.scan.set_ds(pts)
            T, conv, it = self.ndt.align(self.scan, guess.astype(np.float64))
            if not conv:
                coarse.append(dict(candidate=ci, skipped="not converged"))
                continue
            Tf = T.astype(np.float32)
            score, _ = self.ndt.fitness_score(self.scan, Tf.astype(np.float64), self.fitness_score_max_range)
            coarse.append(dict(candidate=ci, T=Tf, score=score, iterations=it))
            if score > best_score:
                continue
            best_score, best, rel = score, ci, Tf
        if best is None:
            return None
        # finetune: the best candidate plus its connected frames in the candidate's frame
        bo = f64(candidates[best][1]).reshape(4, 4)
        accum = [f32(candidates[best][0]).reshape(-1, 4)]
        for ci, pts, odom in connected:
            if ci == best:
                accum.append(transform_cloud_f64(pts, np.linalg.inv(bo) @ f64(odom).reshape(4, 4)))
        accum = np.concatenate(accum)
        self.gicp.set_target(accum)
        self.gicp.set_source(new_points)
        g = np.linalg.inv(rel.astype(np.float64)).astype(np.float32)  # Eigen::Isometry3f(relative_pose).inverse()
        T, conv, it = self.gicp.align(g.astype(np.float64), max_corr_dist=0.5, transformation_epsilon=0.001)
        if not conv:
            return None
        relative_pose = T.astype(np.float32)
        score, _ = self.calc_fitness_score(accum, new_points, relative_pose, self.fitness_score_max_range)
        if score > self.fitness_score_thresh:
            return None
        return dict(best=best, relative_pose=relative_pose, score=score, coarse=coarse, fine_iterations=it)


class Comm:
    """one rank of an RCCL communicator (lio_comm_*): the all-gather of per-rank normal equations for joint registration across GPUs"""

    def __init__(self, rank=0, world=1, device=0, uid=None):
        u = (C.c_uint8 * 128)(*uid) if uid is not None else None
        self.h = lib().lio_comm_init(device, rank, world, u)
        if not self.h:
            raise capi.LioError("lio_comm_init failed: " + lib().lio_last_error().decode())
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id():
        u = (C.c_uint8 * 128)()
        check(lib().lio_comm_unique_id(u), "comm unique id")
        return bytes(u)

    def allgather(self, d_local, d_gathered, d_sum=None, stream=None):
        check(lib().lio_allgather_normal_eq(self.h, d_local, d_gathered, d_sum, stream), "allgather")

    def stats(self):
        n, t = C.c_uint64(), C.c_double()
        check(lib().lio_comm_stats(self.h, C.byref(n), C.byref(t)))
        return n.value, t.value

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib().lio_comm_destroy(self.h)
        self.h = None

    __del__ = close


class Batch:
    """throughput mode, batched (lio_batch_*): B scans per launch against one resident static map, filter loop on the device"""

    def __init__(self, shared_map, n_slots=8, n_groups=3, max_raw=262144, max_ds=100000, sub_maps=None, comm=None):
        """sub_maps / comm: the joint mode (lio_batch_create_joint) -- every job is registered against `shared_map` AND the further `sub_maps`
        resident on this GPU and, through `comm` (lio.Comm), against the sub-maps the other ranks hold; every rank submits the same job list"""
        self.map = shared_map
        self.n_slots, self.n_groups = n_slots, n_groups
        self._keep = (list(sub_maps or []), comm)
        if sub_maps or comm is not None:
            ms = [shared_map] + list(sub_maps or [])
            hs = (C.c_void_p * len(ms))(*[m.h for m in ms])
            self.h = lib().lio_batch_create_joint(hs, len(ms), comm.h if comm is not None else None, n_slots, n_groups, max_raw, max_ds)
        else:
            self.h = lib().lio_batch_create(shared_map.h, n_slots, n_groups, max_raw, max_ds)
        if not self.h:
            raise capi.LioError("lio_batch_create failed: " + lib().lio_last_error().decode())

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib().lio_batch_destroy(self.h)
        self.h = None

    __del__ = close

    def set_gather_hook(self, fn, rank, world):
        """lio_batch_set_gather_hook: fn(d_local: int, d_gathered: int, n_records: int, stream: int) -> int (0 = ok) moves the round's records
        between the ranks instead of RCCL (lsd_amd.dist.RecordsAllGatherHost: gloo, two ranks on one GPU in the tests)"""

        def _cb(_ctx, d_local, d_gathered, n, stream):
            try:
                return int(fn(int(d_local or 0), int(d_gathered or 0), int(n), int(stream or 0)) or 0)
            except Exception:  # nothing may propagate through the C frames
                import traceback

                traceback.print_exc()
                return -1

        self._gather = capi.GATHER_FN(_cb)  # keep the trampoline alive
        check(lib().lio_batch_set_gather_hook(self.h, self._gather, None, int(rank), int(world)), "set_gather_hook")

    def exchange_stats(self):
        """lio_batch_exchange_stats: the all-gather of a joint round's downsampled clouds -- points per slot chunk in force, jobs that ran again because
        their cloud was cut, bytes one rank contributes per round (zeros for one rank / LIO_JOINT_SPLIT_DS=0)"""
        cap, again, nbytes = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        check(lib().lio_batch_exchange_stats(self.h, C.byref(cap), C.byref(again), C.byref(nbytes)))
        return {"chunk_points": cap.value, "jobs_rerun": again.value, "bytes_per_rank_and_round": nbytes.value}

    def enable_kernel_timing(self, on=True):
        check(lib().lio_batch_enable_kernel_timing(self.h, int(on)))

    def kernel_times(self, reset=True):
        t = capi.BatchTimes()
        check(lib().lio_batch_kernel_times(self.h, C.byref(t), int(reset)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def engine(self, group, slot):
        """the engine behind a slot (borrowed: its scan buffers / neighbour cache, the pass log of a host continuation)"""
        h = lib().lio_batch_engine(self.h, group, slot)
        if not h:
            raise capi.LioError("no such slot")
        e = Engine.__new__(Engine)
        e.h = h
        e._own = False
        e.map = self.map
        e.scan = Scan(max_ds=100000, _borrow=lib().lio_engine_scan(h))
        return e

    def process(self, jobs):
        """jobs: list of dicts {dptr, n, t, state (26,), cov (23,23)}; returns (rc, list of result dicts) like process_batch"""
        return process_batch(None, jobs, batch=self)


JOB_KEEP_CACHE = 1  # LIO_JOB_KEEP_CACHE of include/lio_hip.h
JOB_IDLE = 2        # LIO_JOB_IDLE: the session has no scan this round (lio_batch_sequences_step)
JOB_HOST_RAW = 4    # LIO_JOB_HOST_RAW: "dptr" is a HOST address (pinned: PinnedCloud); the library copies it to HBM on the round's stream


class PinnedCloud:
    """a cloud in page-locked host memory (lio_pinned_alloc): .array is an (n, 4) float32 view to fill, .ptr the address for a JOB_HOST_RAW job"""

    def __init__(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 4)
        self.n = len(pts)
        self.nbytes = max(pts.nbytes, 16)
        self.ptr = lib().lio_pinned_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError((lib().lio_last_error() or b'').decode())
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_float)), shape=(max(self.n, 1), 4))[: self.n]
        self.array[...] = pts

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self.array = None
                lib().lio_pinned_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class SequenceBatch:
    """throughput mode WITH map_incremental (lio_batch_create_sequences): n_groups x n_slots independent SLAM sessions, one per slot, each with its
    own map; step() registers the next scan of every session and inserts it into the session's map in one submission per group"""

    def __init__(self, n_slots=8, n_groups=1, resolution=0.5, stencil=19, max_points=2_000_000, max_voxels=1_000_000, max_raw=262144, max_ds=100000, device=0):
        self.n_slots, self.n_groups = n_slots, n_groups
        self.n = n_slots * n_groups
        self.h = lib().lio_batch_create_sequences(device, resolution, stencil, max_points, max_voxels, n_slots, n_groups, max_raw, max_ds)
        if not self.h:
            raise capi.LioError("lio_batch_create_sequences failed: " + lib().lio_last_error().decode())
        self.arr = (capi.ScanJob * self.n)()
        self.states_in = np.zeros((self.n, STATE_DIM))
        self.covs_in = np.zeros((self.n, 23 * 23))
        self.states_out = np.zeros((self.n, STATE_DIM))
        self.covs_out = np.zeros((self.n, 23 * 23))
        for i in range(self.n):
            a = self.arr[i]
            a.state_in = self.states_in[i].ctypes.data_as(C.POINTER(C.c_double))
            a.cov_in = self.covs_in[i].ctypes.data_as(C.POINTER(C.c_double))
            a.state_out = self.states_out[i].ctypes.data_as(C.POINTER(C.c_double))

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib().lio_batch_destroy(self.h)
        self.h = None

    __del__ = close

    def engine(self, session):
        """the engine that owns session `session`'s map and file-scope state (borrowed)"""
        h = lib().lio_batch_engine(self.h, session // self.n_slots, session % self.n_slots)
        if not h:
            raise capi.LioError("no such session")
        e = Engine.__new__(Engine)
        e.h = h
        e._own = False
        e.map = Map(_borrow=lib().lio_engine_map(h))
        e.scan = Scan(max_ds=100000, _borrow=lib().lio_engine_scan(h))
        return e

    def enable_kernel_timing(self, on=True):
        check(lib().lio_batch_enable_kernel_timing(self.h, int(on)))

    def kernel_times(self, reset=True):
        t = capi.BatchTimes()
        check(lib().lio_batch_kernel_times(self.h, C.byref(t), int(reset)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def fastlio_main(self):
        """lio_batch_fastlio_main: one fastlio_main per session (their engines carry the front half: engine(s).fastlio_init, fastlio_imu_enqueue,
        fastlio_pcl_enqueue...), the registrations of all sessions as one round; returns (rc, per-session return codes)"""
        if not hasattr(self, "_fl_rc"):
            self._fl_rc = (C.c_int * self.n)()
        rc = lib().lio_batch_fastlio_main(self.h, self._fl_rc)
        return rc, list(self._fl_rc)

    def load(self, jobs):
        """jobs: one entry per session -- None (idle this round) or a dict {dptr, n, t, state (26,), cov (23,23)}"""
        assert len(jobs) == self.n
        for i, j in enumerate(jobs):
            a = self.arr[i]
            if j is None:
                a.flags = JOB_IDLE
                a.d_raw, a.n_raw = None, 0
                continue
            a.flags = 0
            a.d_raw = j["dptr"]
            a.n_raw = j["n"]
            a.lidar_beg_time = float(j["t"])
            self.states_in[i] = f64(j["state"])
            self.covs_in[i] = f64(j["cov"]).reshape(-1)

    def run(self):
        """the C call alone (on what load() marshalled)"""
        return lib().lio_batch_sequences_step(self.h, self.arr, self.n, self.covs_out.ctypes.data_as(C.POINTER(C.c_double)))

    def step(self, jobs):
        """returns (rc, list of result dicts -- None for an idle session)"""
        self.load(jobs)
        rc = self.run()
        a = self.arr
        res = [None if jobs[i] is None else dict(rc=a[i].rc, n_ds=a[i].n_ds, n_pass=a[i].n_pass, n_knn_pass=a[i].n_knn_pass, state=self.states_out[i].copy(),
                                                  cov=self.covs_out[i].reshape(23, 23).copy()) for i in range(self.n)]
        return rc, res



class PreparedJobs:
    """a job list marshalled once into the C ABI's lio_scan_job array (so that a timed region can be the one C call and nothing else)"""

    def __init__(self, jobs):
        n = len(jobs)
        self.n = n
        self.arr = (capi.ScanJob * n)()
        self.outs = np.zeros((n, STATE_DIM))
        self._keep = []
        cache = {}
        for i, j in enumerate(jobs):
            key = (id(j["state"]), id(j["cov"]))
            if key not in cache:  # the same arrays handed in many times (a repeated list) are converted once
                cache[key] = (f64(j["state"]), f64(j["cov"]).reshape(-1))
                self._keep.append(cache[key])
            st, cv = cache[key]
            a = self.arr[i]
            a.d_raw = j["dptr"]
            a.n_raw = j["n"]
            a.flags = int(j.get("flags", 0))  # 0: an independent scan (the slot forgets its neighbour cache first); JOB_KEEP_CACHE: one of a sequence
            a.lidar_beg_time = float(j["t"])
            a.state_in = ptr(st, C.c_double)
            a.cov_in = ptr(cv, C.c_double)
            a.state_out = self.outs[i].ctypes.data_as(C.POINTER(C.c_double))

    def results(self):
        a = self.arr
        return [dict(rc=a[i].rc, n_ds=a[i].n_ds, n_pass=a[i].n_pass, n_knn_pass=a[i].n_knn_pass, state=self.outs[i]) for i in range(self.n)]


def run_prepared(prep, engines=None, batch=None):
    """the C call alone: lio_batch_process (batch=) or lio_engines_process_batch (engines=) on a PreparedJobs; returns its return code"""
    if batch is not None:
        return lib().lio_batch_process(batch.h, prep.arr, prep.n)
    hs = (C.c_void_p * len(engines))(*[e.h for e in engines])
    return lib().lio_engines_process_batch(hs, len(engines), prep.arr, prep.n)


def process_batch(engines, jobs, batch=None):
    """register independent scans concurrently (C++ worker threads, one per engine; see lio_engines_process_batch) or, with
    batch=, through the batched device-resident engine (lio_batch_process).
    jobs: list of dicts {dptr, n, t, state (26,), cov (23,23)}; returns (rc, list of result dicts)"""
    prep = PreparedJobs(jobs)
    rc = run_prepared(prep, engines=engines, batch=batch)
    return rc, prep.results()


def state_boxplus(s, d):
    s, d = f64(s), f64(d)
    o = np.zeros(STATE_DIM)
    lib().lio_state_boxplus(ptr(s, C.c_double), ptr(d, C.c_double), ptr(o, C.c_double))
    return o


def state_boxminus(a, b):
    a, b = f64(a), f64(b)
    o = np.zeros(DOF)
    lib().lio_state_boxminus(ptr(a, C.c_double), ptr(b, C.c_double), ptr(o, C.c_double))
    return o


def state_predict(s, P, dt, Q12, acc, gyro):
    """one esekf::predict step on the host (no GPU needed): returns (state26, P 23x23)"""
    s, P, q, a, g = f64(s), f64(P).reshape(-1), f64(Q12), f64(acc), f64(gyro)
    assert s.size == STATE_DIM and P.size == 529 and q.size == 12
    so, Po = np.zeros(STATE_DIM), np.zeros(529)
    check(lib().lio_state_predict(ptr(s, C.c_double), ptr(P, C.c_double), float(dt), ptr(q, C.c_double), ptr(a, C.c_double), ptr(g, C.c_double),
                                  ptr(so, C.c_double), ptr(Po, C.c_double)))
    return so, Po.reshape(23, 23)


NO_EFFECTIVE_POINTS = "no effective points"  # a model's answer for a pass in which h_share_model_geometric returns early (laserMapping.cpp:888-893)


def make_meas_fn(model):
    """wrap model(state26, converge) -> (rows (n, 6), h (n,)), None (the pass is invalid) or NO_EFFECTIVE_POINTS (the point-to-plane part found
    nothing: the rows of the previous pass survive in the copied struct, laserMapping.cpp:991) as the C measurement callback of lio_eskf_update_cb"""
    def _cb(ctx, s26, converge, n_out, rows, h, cap):
        r = model(np.ctypeslib.as_array(s26, shape=(STATE_DIM,)).copy(), bool(converge))
        if r is None:
            return 0
        if isinstance(r, str) and r == NO_EFFECTIVE_POINTS:
            n_out[0] = 0
            return 2
        R, H = np.asarray(r[0], np.float64).reshape(-1, 6), np.asarray(r[1], np.float64).ravel()
        n = len(H)
        assert n <= cap and len(R) == n
        np.ctypeslib.as_array(rows, shape=(cap * 6,))[:n * 6] = R.ravel()
        np.ctypeslib.as_array(h, shape=(cap,))[:n] = H
        n_out[0] = n
        return 1
    return capi.MEAS_FN(_cb)


def eskf_update(s, P, R, model, max_iter=4, cap=4096):
    """one iterated ESKF update on the host filter with a Python measurement model (see make_meas_fn)"""
    s, P = f64(s), f64(P).reshape(-1)
    so, Po = np.zeros(STATE_DIM), np.zeros(529)
    fn = make_meas_fn(model)
    check(lib().lio_eskf_update_cb(ptr(s, C.c_double), ptr(P, C.c_double), float(R), max_iter, fn, None, cap, ptr(so, C.c_double), ptr(Po, C.c_double)))
    return so, Po.reshape(23, 23)


def eskf_update_ws(s, P, R, model, ins_vel, degenerate=False, max_iter=4, cap=4096):
    """eskf_update with the wheel-speed rows (laserMapping.cpp:794-811) appended to the model's rows in every pass"""
    s, P, v = f64(s), f64(P).reshape(-1), f64(ins_vel)
    so, Po = np.zeros(STATE_DIM), np.zeros(529)
    fn = make_meas_fn(model)
    check(lib().lio_eskf_update_ws_cb(ptr(s, C.c_double), ptr(P, C.c_double), float(R), max_iter, fn, None, cap, ptr(v, C.c_double), int(degenerate),
                                      ptr(so, C.c_double), ptr(Po, C.c_double)))
    return so, Po.reshape(23, 23)


def sums_of_rows(rows, h):
    """the 29 numbers a device linearisation hands the filter: J^T J upper triangle row by row, J^T h, sum |r|, N_eff"""
    rows, h = np.asarray(rows, np.float64).reshape(-1, 6), np.asarray(h, np.float64).ravel()
    JtJ, Jth = rows.T @ rows, rows.T @ h
    return np.concatenate([[JtJ[a, b] for a in range(6) for b in range(a, 6)], Jth, [np.abs(h).sum(), float(len(h))]])


def eskf_update_sums(s, P, R, model, max_iter=4, degenerate_detect=False):
    """the DEVICE-resident form of the iterated update (csrc/eskf_dev.h compiled for the host) with a Python model:
    model(state26, converge) -> (rows (n, 6), h (n,)) or None.  Returns (state26, P, logs, status)."""
    s, P = f64(s), f64(P).reshape(-1)
    so, Po = np.zeros(STATE_DIM), np.zeros(529)
    last = {}

    def _sums(ctx, s26, converge, acc):
        r = model(np.ctypeslib.as_array(s26, shape=(STATE_DIM,)).copy(), bool(converge))
        if r is None:
            return 0
        last["rows"] = np.asarray(r[0], np.float64).reshape(-1, 6)
        np.ctypeslib.as_array(acc, shape=(29,))[:] = sums_of_rows(r[0], r[1])
        return 1

    def _deg(ctx, V, cs):  # laserMapping.cpp:946-964 on the rows of the last evaluation
        Vm = np.ctypeslib.as_array(V, shape=(9,)).reshape(3, 3)
        n = last["rows"][:, :3]
        n = n / np.linalg.norm(n, axis=1, keepdims=True)
        d = np.abs(n @ Vm).astype(np.float32).astype(np.float64)
        out = np.ctypeslib.as_array(cs, shape=(6,))
        out[:3] = np.where(d > np.float32(0.1736), d, 0).sum(0)
        out[3:] = np.where(d > np.float32(0.7070), d, 0).sum(0)

    f1, f2 = capi.SUMS_FN(_sums), capi.DEG_FN(_deg)
    logs = (capi.PassLog * 8)()
    status = C.c_int(0)
    n = check(lib().lio_eskf_update_sums_cb(ptr(s, C.c_double), ptr(P, C.c_double), float(R), max_iter, int(degenerate_detect), f1, f2, None,
                                            ptr(so, C.c_double), ptr(Po, C.c_double), logs, 8, C.byref(status)), "eskf_update_sums")
    out = [dict(knn=l.knn, n_eff=l.n_eff, valid=l.valid, degenerate=l.degenerate, sum_abs_res=l.sum_abs_res, JtJ=np.array(l.JtJ).reshape(6, 6),
                Jtr=np.array(l.Jtr), dx=np.array(l.dx)) for l in logs[:n]]
    return so, Po.reshape(23, 23), out, status.value


class PoseEstimator:
    """hdl_localization::PoseEstimator over the device matcher: 23-state UKF (host, f32) + lio_ndt_align.
    Stamps in microseconds, quaternions (w, x, y, z), as in the reference."""

    def __init__(self, pos, quat_wxyz, stamp_us=0, imu_ext=np.eye(4), cool_time=1.0):
        a, b, c = f32(imu_ext).reshape(-1), f32(pos), f32(quat_wxyz)
        self.h = lib().lio_pose_estimator_create(ptr(a, C.c_float), int(stamp_us), ptr(b, C.c_float), ptr(c, C.c_float), float(cool_time))
        if not self.h:
            raise capi.LioError("lio_pose_estimator_create failed")

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib().lio_pose_estimator_destroy(self.h)
        self.h = None

    __del__ = close

    def predict(self, stamp_us, acc=None, gyro=None):
        if acc is None:
            return check(lib().lio_pose_estimator_predict(self.h, int(stamp_us), None, None), "predict")
        a, g = f32(acc), f32(gyro)
        return check(lib().lio_pose_estimator_predict(self.h, int(stamp_us), ptr(a, C.c_float), ptr(g, C.c_float)), "predict")

    def match(self, ndt, scan, params=None):
        """returns (ok, observation (7,), LM iterations)"""
        obs = np.zeros(7, np.float32)
        it = C.c_int(0)
        p = params
        if p is None:
            p = capi.NdtParams()
            lib().lio_ndt_default_params(C.byref(p))
        ok = check(lib().lio_pose_estimator_match(self.h, ndt.h, scan.h, C.byref(p), ptr(obs, C.c_float), C.byref(it)), "match")
        return bool(ok), obs, it.value

    @staticmethod
    def _gps(gps):
        """gps: None or (T 4x4, precision, dimension) -> pointer to a lio_gps_observation (kept alive by the caller's frame)"""
        if gps is None:
            return None, None
        g = capi.GpsObservation()
        T = f64(gps[0]).reshape(16)
        for i in range(16):
            g.T[i] = T[i]
        g.precision, g.dimension = float(gps[1]), int(gps[2])
        return g, C.byref(g)

    def guess(self, gps=None):
        """the pose the matcher starts from: the filter's, fused with the GNSS observation when there is one"""
        keep, gp = self._gps(gps)
        T = np.zeros(16, np.float32)
        check(lib().lio_pose_estimator_guess(self.h, gp, ptr(T, C.c_float)), "guess")
        return T.reshape(4, 4)

    def observe(self, init_guess, aligned, converged=True, gps=None):
        """what match() makes of the matcher's answer: (ok, observation (7,), observation covariance (7, 7))"""
        keep, gp = self._gps(gps)
        a, b = f32(init_guess).reshape(16), f32(aligned).reshape(16)
        obs, cov = np.zeros(7, np.float32), np.zeros(49, np.float32)
        ok = check(lib().lio_pose_estimator_observe(self.h, ptr(a, C.c_float), ptr(b, C.c_float), int(converged), gp, ptr(obs, C.c_float), ptr(cov, C.c_float)), "observe")
        return bool(ok), obs, cov.reshape(7, 7)

    def match_gps(self, ndt, scan, gps, params=None):
        keep, gp = self._gps(gps)
        obs, cov, it = np.zeros(7, np.float32), np.zeros(49, np.float32), C.c_int(0)
        p = params
        if p is None:
            p = capi.NdtParams()
            lib().lio_ndt_default_params(C.byref(p))
        ok = check(lib().lio_pose_estimator_match_gps(self.h, ndt.h, scan.h, C.byref(p), gp, ptr(obs, C.c_float), ptr(cov, C.c_float), C.byref(it)), "match_gps")
        return bool(ok), obs, cov.reshape(7, 7), it.value

    def match_gps_only(self, gps):
        keep, gp = self._gps(gps)
        obs, cov = np.zeros(7, np.float32), np.zeros(49, np.float32)
        ok = check(lib().lio_pose_estimator_match_gps_only(self.h, gp, ptr(obs, C.c_float), ptr(cov, C.c_float)), "match_gps_only")
        return bool(ok), obs, cov.reshape(7, 7)

    def get_timed_pose(self, stamp_us, acc_g, gyro_dps):
        """one INS sample (g, deg/s) -> (accepted, pose 4x4 at that sample); extends the state queue that correct() re-predicts"""
        a, g, T = f64(acc_g), f64(gyro_dps), np.zeros(16)
        ok = check(lib().lio_pose_estimator_get_timed_pose(self.h, int(stamp_us), ptr(a, C.c_double), ptr(g, C.c_double), ptr(T, C.c_double)), "get_timed_pose")
        return bool(ok), T.reshape(4, 4)

    def predict_nostate(self, stamp_us):
        T = np.zeros(16)
        check(lib().lio_pose_estimator_predict_nostate(self.h, int(stamp_us), ptr(T, C.c_double)), "predict_nostate")
        return T.reshape(4, 4)

    def correct(self, stamp_us, observation):
        z = f32(observation)
        check(lib().lio_pose_estimator_correct(self.h, int(stamp_us), ptr(z, C.c_float)), "correct")

    def get(self):
        m, c = np.zeros(23, np.float32), np.zeros(529, np.float32)
        check(lib().lio_pose_estimator_get(self.h, ptr(m, C.c_float), ptr(c, C.c_float)))
        return m, c.reshape(23, 23)

    def set(self, mean=None, cov=None):
        m = f32(mean) if mean is not None else None
        c = f32(cov).reshape(-1) if cov is not None else None
        check(lib().lio_pose_estimator_set(self.h, ptr(m, C.c_float) if m is not None else None, ptr(c, C.c_float) if c is not None else None))

    def matrix(self):
        T = np.zeros(16, np.float32)
        check(lib().lio_pose_estimator_matrix(self.h, ptr(T, C.c_float)))
        return T.reshape(4, 4)


class LocalMap:
    """Localization::runUpdateLocalMap on the device: key-frame clouds resident in HBM, local map = gather + VoxelGrid + NDT target"""

    def __init__(self, max_total_points=20_000_000, max_local_points=200_000, max_keyframe_points=200_000, device=0):
        self.h = lib().lio_localmap_create(device, max_total_points, max_local_points, max_keyframe_points)
        if not self.h:
            raise capi.LioError("lio_localmap_create failed: " + lib().lio_last_error().decode())
        self._cap = max_local_points + max_keyframe_points

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib().lio_localmap_destroy(self.h)
        self.h = None

    __del__ = close

    def add_keyframe(self, world_xyzi, position):
        p, c = f32(world_xyzi).reshape(-1, 4), f32(position)
        return check(lib().lio_localmap_add_keyframe(self.h, ptr(p, C.c_float), len(p), ptr(c, C.c_float)), "add_keyframe")

    def update(self, ndt, pose_xyz, update_distance=10.0, radius=30.0, key_frame_distance=1.0, leaf=0.2):
        """returns (code, key frames used, points in the new target); code as lio_localmap_update"""
        x = f64(pose_xyz)
        nk, npts = C.c_int(0), C.c_uint32(0)
        rc = check(lib().lio_localmap_update(self.h, ndt.h, ptr(x, C.c_double), float(update_distance), float(radius), float(key_frame_distance), float(leaf),
                                             C.byref(nk), C.byref(npts)), "localmap update")
        return rc, nk.value, npts.value

    def download(self):
        out = np.zeros((self._cap, 4), np.float32)
        n = check(lib().lio_localmap_download(self.h, ptr(out, C.c_float), self._cap))
        return out[:n].copy()
