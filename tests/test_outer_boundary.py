"""The reference's OUTER boundary: the compiled pybind11 module `slam_wrapper` (lidar-slam-detection_amd/csrc/slam_wrapper.cpp) that
slam/slam.py and slam/map_manager.py import.
  CPU (needs /root/reference): every function and argument name of the reference's PYBIND11_MODULE block exists with the same names in the
      same order; the reference's slam/slam.py, imported BYTE FOR BYTE (three stand-in modules for things this container lacks, SURVEY.md
      Appendix B), constructs SLAM and runs start() / stop() through the compiled module -- without a GPU setup_slam() returns False, as the
      reference's does when its back end cannot start;
  GPU: >= 13 scans through process() with the reference's units and dictionaries; the odometry follows the analytic trajectory and equals the
      ctypes path over the same C ABI."""
import os
import re
import sys
import types

import numpy as np
import pytest

REF = "/root/reference"
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lidar-slam-detection_amd", "python")
if PKG not in sys.path:
    sys.path.insert(0, PKG)


def _module():
    import slam_wrapper  # the compiled module (lidar-slam-detection_amd/python/slam_wrapper.*.so)

    assert slam_wrapper.__file__.endswith(".so"), slam_wrapper.__file__
    return slam_wrapper


def test_module_loads_and_is_compiled():
    m = _module()
    assert m.__doc__ and "mapping python interface" in m.__doc__


@pytest.mark.skipif(not os.path.exists(REF + "/slam/src/slam_wrapper.cpp"), reason="needs /root/reference")
def test_surface_matches_the_reference_module():
    m = _module()
    src = open(REF + "/slam/src/slam_wrapper.cpp").read()
    block = src[src.index("PYBIND11_MODULE(slam_wrapper, m)"):]
    defs = re.findall(r'm\.def\("(\w+)",\s*&\w+(.*?)\);', block, re.S)
    assert len(defs) >= 45
    for name, rest in defs:
        assert hasattr(m, name), name
        args = re.findall(r'py::arg\("(\w+)"\)', rest)
        doc = getattr(m, name).__doc__
        sig = doc.split("\n")[0]
        got = re.findall(r"(\w+): ", sig[sig.index("(") + 1:sig.rindex(")")])
        assert got == args, (name, got, args)


def _stand_ins():
    """cpp_utils_ext (only set_thread_priority is used), shapely.geometry (Point.within(Polygon)), proto.internal_pb2"""
    m = types.ModuleType("cpp_utils_ext")
    m.set_thread_priority = lambda name, prio: None
    sys.modules["cpp_utils_ext"] = m
    sh, geo = types.ModuleType("shapely"), types.ModuleType("shapely.geometry")

    class Point:
        def __init__(self, *xy):
            self.xy = xy

        def within(self, poly):
            return False

    class Polygon:
        def __init__(self, pts):
            self.pts = pts

    geo.Point, geo.Polygon = Point, Polygon
    sh.geometry = geo
    sys.modules["shapely"], sys.modules["shapely.geometry"] = sh, geo
    pr, pb = types.ModuleType("proto"), types.ModuleType("proto.internal_pb2")
    pr.internal_pb2 = pb
    sys.modules["proto"], sys.modules["proto.internal_pb2"] = pr, pb


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _cfg(d):
    return _Cfg({k: _cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


@pytest.mark.skipif(not os.path.exists(REF + "/slam/slam.py"), reason="needs /root/reference")
def test_reference_slam_py_runs_unchanged_on_the_compiled_module():
    _module()
    _stand_ins()
    if REF not in sys.path:
        sys.path.insert(1, REF)
    import importlib

    ref_slam = importlib.import_module("slam.slam")  # /root/reference/slam/slam.py, byte for byte
    assert ref_slam.__file__ == REF + "/slam/slam.py"
    assert ref_slam.slam.__file__.endswith(".so") and "lidar-slam-detection_amd" in ref_slam.slam.__file__   # its `import slam_wrapper as slam`

    class Log:
        def info(self, *a):
            pass

        warn = error = debug = info

    config = _cfg(dict(
        input=dict(mode="offline"), camera=[],
        ins=dict(extrinsic_parameters=[0.3, 0.1, -0.2, 2.0, 1.0, -4.0], imu_extrinsic_parameters=[0.05, -0.02, 0.1, -1.0, 0.5, 3.0], ins_type="6D"),
        output=dict(localization=dict(UDP=dict(use=False, destination="127.0.0.1", port=9000))),
        slam=dict(origin=dict(use=False, latitude=31.0, longitude=121.0, altitude=4.0),
                  mapping=dict(key_frames_range=50.0, ground_constraint=True, loop_closure=True, gravity_constraint=False),
                  localization=dict(colouration=False))))
    s = ref_slam.SLAM("mapping", "FastLIO", "/tmp/lsd_map", ["0-Ouster", "IMU"], 0.2, [1.0, 10.0], config, Log())
    s.start()  # slam.py:50-85: init_slam, set_camera_param, set_*_external_param, set_ins_config, set_destination, setup_slam, set_map_origin, ...
    assert s.isInited()
    import slam_wrapper as sw

    assert sw.get_mapping_ground_constraint() is True
    assert np.allclose(sw.get_map_origin(), [[31.0, 121.0, 4.0, 0, 0, 0, 0]])
    # a few of the calls slam.py / map_manager.py make off the hot path: type-correct values, nothing raises
    assert sw.update_odom() == {"odoms": {}, "keyframes": []} and sw.get_graph_status() == {"loop_detected": False}
    assert sw.get_graph_map() == {"points": {}, "images": {}, "poses": {}, "stamps": {}} and sw.get_graph_edges() == {} and sw.run_robust_graph_optimization("mapping") == {}
    G = np.eye(4, dtype=np.float32)
    G[0, 3] = 60.0
    A = sw.pointcloud_align(np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32), G)
    assert A.shape == (4, 4) and A[0, 3] == 0.0  # graph_utils.cpp:26-30: a guess farther than 50 m loses its translation
    s.stop()


def _rpyt(x, y, z, yaw, pitch, roll):
    """getTransformFromRPYT (slam/common/slam_utils.cpp:89-96): translation * Rz(yaw) * Rx(pitch) * Ry(roll), degrees"""
    k = 0.01745329251994
    cy, sy, cp, sp, cr, sr = np.cos(yaw * k), np.sin(yaw * k), np.cos(pitch * k), np.sin(pitch * k), np.cos(roll * k), np.sin(roll * k)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Ry = np.array([[cr, 0, sr], [0, 1, 0], [-sr, 0, cr]])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = Rz @ Rx @ Ry, [x, y, z]
    return T


def _drive(process, n=16, imu_ext=(0.05, -0.02, 0.10, 3.0, 0.5, -1.0), ins_ext=(0.30, 0.10, -0.20, -4.0, 1.0, 2.0)):
    from lsd_amd import synth

    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
    tr = synth.Trajectory()
    T_li = _rpyt(*imu_ext)
    imu = synth.imu_stream(tr, 0.0, 0.1 * n + 0.3, rate=200.0)
    ii, outs = 0, []
    for k in range(n):
        tb = k * 0.1
        pts, st = synth.make_sweep(scene, tr, tb, ext_R=T_li[:3, :3], ext_t=T_li[:3, 3], seed=k, n_az=900, fov_deg=(-24.8, 2.0))
        rows = []
        while ii < len(imu) and imu[ii][0] <= tb + 0.12:
            t, g, a = imu[ii]
            rows.append([t * 1e6, *(g * 180.0 / np.pi), *(a / 9.81)])
            ii += 1
        attr = dict(timestamp=int(round(tb * 1e6)), points_attr=np.stack([st.astype(np.float32), np.zeros(len(st), np.float32)], 1))
        out = process({"0-lidar": pts}, {"0-lidar": attr}, {}, {}, {}, {}, np.array(rows, np.float64).reshape(-1, 7), int(round(tb * 1e6)))
        outs.append((tb, out))
    return tr, outs


@pytest.mark.gpu
def test_process_through_the_compiled_module():
    from lsd_amd import capi

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device")
    sw = _module()
    imu_ext = (0.05, -0.02, 0.10, 3.0, 0.5, -1.0)
    ins_ext = (0.30, 0.10, -0.20, -4.0, 1.0, 2.0)
    assert sw.init_slam("mapping", "", "FastLIO", ["0-lidar", "IMU"], 0.5, 1.0, 10.0, 100) == ["IMU", "0-lidar"]
    sw._set_capacity(4_000_000, 1 << 20)
    sw.set_ins_external_param(*ins_ext)
    sw.set_imu_external_param(*imu_ext)
    assert sw.setup_slam() is True
    try:
        tr, outs = _drive(sw.process)
    finally:
        sw.deinit_slam()
    T_li, T_ln = _rpyt(*imu_ext), _rpyt(*ins_ext)
    T_ni = T_li @ np.linalg.inv(T_ln)
    W0 = np.eye(4)
    W0[:3, :3], W0[:3, 3] = tr.R(0.0), tr.pos(0.0)
    worst = 0.0
    for tb, a in outs:
        assert a["slam_valid"] is True and a["frame_start_timestamp"] == int(round(tb * 1e6))
        p = a["pose"]
        assert set(p) == {"latitude", "longitude", "altitude", "heading", "pitch", "roll", "Ve", "Vn", "Vu", "Status", "state", "timestamp", "odom_matrix"}
        M = p["odom_matrix"]
        assert M.dtype == np.float32 and M.shape == (4, 4) and p["state"] == "Mapping" and 0.0 <= p["heading"] < 360.0
        if tb >= 0.85:
            Wt = np.eye(4)
            Wt[:3, :3], Wt[:3, 3] = tr.R(tb), tr.pos(tb)
            truth = np.linalg.inv(T_ni) @ (np.linalg.inv(W0) @ Wt) @ T_ni
            dp = float(np.linalg.norm(M[:3, 3] - truth[:3, 3]))
            worst = max(worst, dp)
            assert dp < 0.05, (tb, dp)
    print("compiled slam_wrapper.process: worst position error %.4f m over %d scans" % (worst, len(outs)))


@pytest.mark.gpu
def test_reference_hdl_fastlio_class_linked_against_the_library():
    """INTEGRATION.md's Option 0, linked and RUN: the reference's own Mapping::HDL_FastLIO (fastlio.cpp compiled whole, its runLio thread, its
    feedImuData / feedPointData / getPose) over liblio_hip.so through the binding the document shows.  Its poses equal the compiled
    slam_wrapper's (same engine, same inputs) and follow the trajectory; the per-scan delta odometry and the IMU prediction of getPose run too."""
    from lsd_amd import capi

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device")
    import ref_hdl_fastlio

    if not ref_hdl_fastlio.available():
        pytest.skip("oracle/_ref/libref_hdl_fastlio.so not built (needs /root/reference at build time)")
    imu_ext = (0.05, -0.02, 0.10, 3.0, 0.5, -1.0)
    ins_ext = (0.30, 0.10, -0.20, -4.0, 1.0, 2.0)
    T_li, T_ln = _rpyt(*imu_ext), _rpyt(*ins_ext)
    H = ref_hdl_fastlio.HdlFastLio("0-lidar", T_static=T_ln, T_imu=T_li, scan_period=0.1)
    poses = []

    def process(points, points_attr, _a, _b, _c, _d, imu_rows, timestamp):
        for row in imu_rows:  # numpy_to_imu's units
            H.feed_imu(row[0] / 1e6, row[1:4] / 180.0 * np.pi, row[4:7] * 9.81)
        attr = points_attr["0-lidar"]
        T, D, n_imu = H.frame(points["0-lidar"], attr["points_attr"][:, 0].astype(np.uint32), attr["timestamp"])
        poses.append((T, D, n_imu))
        return None

    try:
        tr, _ = _drive(process)
    finally:
        H.close()
    sw = _module()
    assert sw.init_slam("mapping", "", "FastLIO", ["0-lidar", "IMU"], 0.5, 1.0, 10.0, 100) == ["IMU", "0-lidar"]
    sw._set_capacity(8_000_000, 400_000)  # the binding's lio_engine_create arguments
    sw.set_ins_external_param(*ins_ext)
    sw.set_imu_external_param(*imu_ext)
    assert sw.setup_slam() is True
    try:
        _, outs = _drive(sw.process)
    finally:
        sw.deinit_slam()
    assert len(poses) == len(outs)
    for k, ((T, D, n_imu), (tb, o)) in enumerate(zip(poses, outs)):
        assert np.abs(T.astype(np.float32) - o["pose"]["odom_matrix"]).max() < 1e-6, k   # the same engine behind both: same numbers
        if tb >= 0.85:
            assert n_imu >= 10 and np.isfinite(D).all()  # getPose's IMU prediction (fastlio.cpp:18-101) ran on the 200 Hz samples of the scan


@pytest.mark.gpu
def test_pointcloud_align_runs_gicp_on_the_device():
    """slam_wrapper.pointcloud_align (graph_utils.cpp:20-46): Generalized-ICP from a guess -- here lio_gicp_* with the reference's settings"""
    import gicp_cases
    import slam_wrapper as sw

    c = gicp_cases.make("room_small")
    T = sw.pointcloud_align(c["source"], c["target"], c["guess"].astype(np.float32))
    assert T.dtype == np.float32 and T.shape == (4, 4)
    assert np.abs(T[:3, 3] - c["truth"][:3, 3]).max() < 0.01 and np.abs(T[:3, :3] - c["truth"][:3, :3]).max() < 2e-3
    assert np.abs(c["guess"][:3, 3] - c["truth"][:3, 3]).max() > 0.1


# ---- the localisation mode at the outer boundary ------------------------------------------------------------------------------------------
def _write_map(sw, root, scene, n_frames=30, n_az=600):
    """a map on disk in the reference's own layout (<map>/graph/<id>/{data, cloud.pcd}, KeyFrame::save) written by the module's dump_keyframe --
    what map_manager.py's saving thread calls: key frames every 2 m along x, clouds in the key frame's own frame, poses to 6 significant digits"""
    from lsd_amd import synth

    os.makedirs(os.path.join(root, "graph"), exist_ok=True)
    frames = []
    for k in range(n_frames):
        pos = np.array([-20.0 + 2.0 * k, 0.3 * np.sin(0.5 * k), 1.8])
        q = synth.quat_from_rotvec([0, 0, 0.05 * np.cos(0.3 * k)])
        raw, _ = synth.make_scan(scene, pos, q, seed=400 + k, n_az=n_az, fov_deg=(-24.8, 2.0))
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3] = synth.quat_to_R(q), pos
        d = os.path.join(root, "graph", "%06d" % k)
        os.makedirs(d, exist_ok=True)
        pts = raw.copy()
        pts[:, 3] /= 255.0  # dump_keyframe scales the intensity by 255 (numpy_to_pointcloud(points, 255.0), graph_utils.cpp:124)
        sw.dump_keyframe(d, 1_000_000 + 100_000 * k, k, pts, T)
        frames.append((raw, T))
    return frames


def test_keyframe_files_have_the_reference_layout(tmp_path):
    """dump_keyframe -> `data` (stamp sec nsec / estimate 4x4 / odom 4x4 / id, keyframe.cpp:122-132) + `cloud.pcd` (PCL binary, x y z intensity)"""
    sw = _module()
    pts = np.array([[1.0, 2.0, 3.0, 0.5], [-4.0, 5.5, 0.25, 1.0]], np.float32)
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [10.5, -2.25, 1.0]
    sw.dump_keyframe(str(tmp_path), 12_345_678, 7, pts, T)
    toks = open(tmp_path / "data").read().split()
    assert toks[:3] == ["stamp", "12", "345678000"] and toks[3] == "estimate" and toks[20] == "odom" and toks[-2:] == ["id", "7"]
    assert [float(v) for v in toks[4:20]] == T.reshape(-1).tolist()
    raw = open(tmp_path / "cloud.pcd", "rb").read()
    head, body = raw[:raw.index(b"DATA binary\n") + 12], raw[raw.index(b"DATA binary\n") + 12:]
    assert b"FIELDS x y z intensity" in head and b"POINTS 2" in head and len(body) == 32
    got = np.frombuffer(body, np.float32).reshape(2, 4)
    assert np.array_equal(got[:, :3], pts[:, :3]) and np.array_equal(got[:, 3], pts[:, 3] * 255.0)


@pytest.mark.skipif(not os.path.exists(REF + "/slam/slam.py"), reason="needs /root/reference")
def test_reference_slam_py_starts_in_localization_mode(tmp_path):
    """the reference's slam/slam.py, byte for byte, constructed with mode = "localization" on the compiled module (CPU): start() goes through
    init_slam / setup_slam (which reads the map's key frames from disk; without a GPU it then returns False as the reference's does when its back
    end cannot start) and hands the key frames to map_manager.py (slam.py:79-81: get_graph_map -> MapManager.update)"""
    from lsd_amd import synth

    sw = _module()
    _stand_ins()
    if REF not in sys.path:
        sys.path.insert(1, REF)
    import importlib

    ref_slam = importlib.import_module("slam.slam")
    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
    root = str(tmp_path / "map")
    frames = _write_map(sw, root, scene, n_frames=6, n_az=150)

    class Log:
        def info(self, *a):
            pass

        warn = error = debug = info

    s = ref_slam.SLAM("localization", "FastLIO", root, ["0-lidar", "IMU"], 0.2, [1.0, 10.0], _LOC_CONFIG(), Log())
    assert s.method == "Localization"
    s.start()
    try:
        assert s.isInited()
        mm = s.map_manager
        assert sorted(mm.data["stamps"], key=int) == [str(k) for k in range(len(frames))] and mm.vertex_id == len(frames)
        assert mm.data["stamps"]["2"] == 1_000_000 + 200_000
        assert np.array_equal(mm.data["points"]["3"][:, :3], frames[3][0][:, :3])
        assert np.allclose(mm.data["points"]["3"][:, 3], frames[3][0][:, 3], rtol=1e-6)  # / 255 on the way in, * 255 inside dump_keyframe
        assert np.abs(mm.data["poses"]["3"] - frames[3][1]).max() < 2e-3  # `data` keeps 6 significant digits
    finally:
        s.stop()


def _LOC_CONFIG():
    return _cfg(dict(
        input=dict(mode="offline"), camera=[],
        ins=dict(extrinsic_parameters=[0.0, 0.0, 0.0, 0.0, 0.0, 0.0], imu_extrinsic_parameters=[0.0, 0.0, 0.0, 0.0, 0.0, 0.0], ins_type="6D"),
        output=dict(localization=dict(UDP=dict(use=False, destination="127.0.0.1", port=9000))),
        slam=dict(origin=dict(use=True, latitude=0.0, longitude=0.0, altitude=0.0),
                  mapping=dict(key_frames_range=50.0, ground_constraint=False, loop_closure=False, gravity_constraint=False),
                  localization=dict(colouration=False))))


@pytest.mark.gpu
def test_localization_mode_through_the_compiled_module(tmp_path):
    """the calls slam.py makes in localisation mode (slam.py:50-85, :140-160), on the GPU: init_slam("localization", map_path, "Localization", ...)
    -> setup_slam loads the map from disk (key frames -> HBM) -> get_graph_map -> set_init_pose starts the filter -> process() returns localised
    poses along a drive through the mapped area: VoxelGrid + NDT-P2D + UKF + local-map updates on the device (docs/slam.md: decimetre level)"""
    from lsd_amd import capi, synth

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device")
    sw = _module()
    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
    root = str(tmp_path / "map")
    frames = _write_map(sw, root, scene)
    assert sw.init_slam("offline", root, "Localization", ["0-lidar", "IMU"], 0.2, 1.0, 10.0, 50.0) == ["0-lidar", "IMU"]
    sw.set_ins_external_param(0, 0, 0, 0, 0, 0)
    sw.set_imu_external_param(0, 0, 0, 0, 0, 0)
    assert sw.setup_slam() is True
    try:
        gm = sw.get_graph_map()
        assert sorted(gm["stamps"], key=int) == [str(k) for k in range(len(frames))] and np.array_equal(gm["points"]["3"][:, :3], frames[3][0][:, :3])
        tr = synth.Trajectory(p0=(-10.0, 0.2, 1.8), t_static=0.2, speed=4.0, heading=0.0, sway=0.3, yaw_amp=0.1)
        imu = synth.imu_stream(tr, 0.0, 4.3, rate=100.0, seed=5, gyr_sigma=1e-3, acc_sigma=1e-2)
        # before an initial pose: "Initializing", identity pose (the global locator is out of scope; slam.set_init_pose supplies the pose)
        pts, st = synth.make_sweep(scene, tr, 0.0, seed=500, n_az=900, fov_deg=(-24.8, 2.0))
        attr = dict(timestamp=0, points_attr=np.stack([st.astype(np.float32), np.zeros(len(st), np.float32)], 1))
        out = sw.process({"0-lidar": pts}, {"0-lidar": attr}, {}, {}, {}, {}, np.zeros((0, 7)), 0)
        assert out["pose"]["state"] == "Initializing" and np.array_equal(out["pose"]["odom_matrix"], np.eye(4, dtype=np.float32))
        p0, R0 = tr.pos(0.1), tr.R(0.1)
        yaw0 = np.degrees(np.arctan2(R0[1, 0], R0[0, 0]))
        sw.set_init_pose(p0[0] + 0.2, p0[1] - 0.15, p0[2], yaw0 + 1.0, 0.0, 0.0)  # slam.py's set_init_pose interface: x, y, z, yaw, pitch, roll
        ii, errs, states = 0, [], []
        for k in range(1, 41):
            tb = k * 0.1
            rows = []
            while ii < len(imu) and imu[ii][0] <= tb:
                t, g, a = imu[ii]
                rows.append([t * 1e6, *(g * 180.0 / np.pi), *(a / 9.81)])
                ii += 1
            pts, st = synth.make_sweep(scene, tr, tb, seed=500 + k, n_az=900, fov_deg=(-24.8, 2.0))
            attr = dict(timestamp=int(round(tb * 1e6)), points_attr=np.stack([st.astype(np.float32), np.zeros(len(st), np.float32)], 1))
            out = sw.process({"0-lidar": pts}, {"0-lidar": attr}, {}, {}, {}, {}, np.array(rows, np.float64).reshape(-1, 7), int(round(tb * 1e6)))
            M = out["pose"]["odom_matrix"]
            assert M.dtype == np.float32 and out["slam_valid"] is True and 0.0 <= out["pose"]["heading"] < 360.0
            errs.append(float(np.linalg.norm(M[:3, 3] - tr.pos(tb))))
            states.append(out["pose"]["state"])
        print("localisation mode: position error median %.3f m, max after the first second %.3f m" % (np.median(errs), max(errs[10:])))
        assert max(errs[10:]) < 0.15 and np.median(errs) < 0.08
        assert states[0] == "Initializing" and states[-1] == "Localizing(L)"  # Localization::isStable: ten good frames
    finally:
        sw.deinit_slam()


@pytest.mark.skipif(not os.path.exists(REF + "/slam/map_manager.py"), reason="needs /root/reference")
def test_reference_map_manager_py_runs_unchanged_on_the_compiled_module(tmp_path):
    """the reference's slam/map_manager.py, byte for byte: MapManager's calls into the module -- add / delete edges and vertices, the export
    sequence (set_export_map_config -> export_points -> dump_map_points -> get_map_origin), the saving thread (dump_odometry, dump_graph,
    dump_keyframe) -- run with the reference's argument types; key frames it is given come back from disk as the localisation mode loads them"""
    sw = _module()
    _stand_ins()
    if REF not in sys.path:
        sys.path.insert(1, REF)
    import importlib

    mmod = importlib.import_module("slam.map_manager")
    assert mmod.__file__ == REF + "/slam/map_manager.py" and mmod.slam.__file__.endswith(".so")

    class Log:
        def info(self, *a):
            pass

        warn = error = debug = info

    sw.init_slam("mapping", "", "FastLIO", ["0-lidar", "IMU"], 0.5, 1.0, 10.0, 100)
    mm = mmod.MapManager(_cfg(dict(dummy=1)), Log())
    rng = np.random.default_rng(0)
    for k in range(3):
        T = np.eye(4, dtype=np.float32)
        T[0, 3] = 2.0 * k
        mm.add_key_frame(rng.normal(size=(50, 4)).astype(np.float32), {}, T, 1_000_000 + k)
    assert mm.vertex_id == 3 and mm.get_pose()["1"][3] == 2.0
    mm.add_edge(0, 1, np.eye(4).tolist())
    mm.del_edge(0)
    mm.set_vertex_fix(1, True)
    A = mm.keyframe_align("0", "1", np.eye(4).reshape(-1).tolist())
    assert len(A) == 16
    mm.set_export_map_config(-1.0, 3.0, "height")
    mm.del_vertex(2)
    assert sorted(mm.data["points"]) == ["0", "1"]
    # the saving thread's module calls, directly (its config dump goes through the web stack's RPC, out of scope)
    graph = tmp_path / "graph"
    graph.mkdir()
    sw.dump_odometry(str(graph))
    assert sw.dump_graph(str(graph)) == []
    for idx in ("0", "1"):
        d = graph / ("%06d" % int(idx))
        d.mkdir()
        sw.dump_keyframe(str(d), mm.data["stamps"][idx], int(idx), mm.data["points"][idx], mm.data["poses"][idx])
        assert (d / "data").exists() and (d / "cloud.pcd").exists()
    assert np.asarray(sw.get_map_origin()).shape == (1, 7)
    sw.deinit_slam()


@pytest.mark.gpu
def test_process_forwards_the_ins_velocity_to_the_front_half():
    """SLAM::run -> preprocessInsData -> HDL_FastLIO::feedInsData -> fastlio_ins_enqueue (slam.cpp:283-296, fastlio.cpp:185-187, laserMapping.cpp:
    417-441): with "RTK" among the sensors and a GNSS status the ins config accepts, the INS velocity (ENU -> ego -> IMU, up component zeroed)
    reaches IMU initialisation, which seeds the filter's velocity from it (IMU_Processing.hpp:201-204); an unaccepted status does not"""
    import ctypes as C

    from lsd_amd import capi, synth

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device")
    sw = _module()
    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
    tr = synth.Trajectory()  # at rest for 1.5 s: the IMU initialisation window
    imu = synth.imu_stream(tr, 0.0, 1.5, rate=200.0)
    seen = {}
    for label, gnss_status in (("accepted", 4), ("rejected", 1)):
        assert sw.init_slam("mapping", "", "FastLIO", ["0-lidar", "IMU", "RTK"], 0.5, 1.0, 10.0, 100) == ["IMU", "RTK", "0-lidar"]
        sw._set_capacity(2_000_000, 1 << 19)
        sw.set_ins_external_param(0, 0, 0, 0, 0, 0)
        sw.set_imu_external_param(0, 0, 0, 0, 0, 0)
        sw.set_ins_config(dict(ins_normal=dict(use=False, status=1, stable_time=0.0, precision=1.0), ins_float=dict(use=False, status=5, stable_time=0.0, precision=0.5),
                               ins_fix=dict(use=True, status=4, stable_time=0.0, precision=0.05)))
        assert sw.setup_slam() is True
        try:
            ii, vel = 0, None
            for k in range(12):
                tb = k * 0.1
                pts, st = synth.make_sweep(scene, tr, tb, seed=k, n_az=600, fov_deg=(-24.8, 2.0))
                rows = []
                while ii < len(imu) and imu[ii][0] <= tb + 0.12:
                    t, g, a = imu[ii]
                    rows.append([t * 1e6, *(g * 180.0 / np.pi), *(a / 9.81)])
                    ii += 1
                attr = dict(timestamp=int(round(tb * 1e6)), points_attr=np.stack([st.astype(np.float32), np.zeros(len(st), np.float32)], 1))
                rtk = dict(timestamp=int(round((tb + 0.09) * 1e6)), longitude=121.0, latitude=31.0, altitude=4.0, heading=90.0, pitch=0.0, roll=0.0,
                           gyro_x=0.0, gyro_y=0.0, gyro_z=0.0, acc_x=0.0, acc_y=0.0, acc_z=1.0, Ve=1.5, Vn=0.0, Vu=0.3, Status=gnss_status, Sensor="GNSS")
                sw.process({"0-lidar": pts}, {"0-lidar": attr}, {}, {}, {}, rtk, np.array(rows, np.float64).reshape(-1, 7), int(round(tb * 1e6)))
                if vel is None and capi.lib().lio_fastlio_is_init(C.c_void_p(sw._engine_handle())):
                    s26 = np.zeros(26)
                    capi.lib().lio_fastlio_start_state(C.c_void_p(sw._engine_handle()), s26.ctypes.data_as(C.POINTER(C.c_double)))
                    vel = s26[14:17].copy()  # pos3 rot4 R_il4 t_il3 vel3
            seen[label] = vel
        finally:
            sw.deinit_slam()
    assert seen["accepted"] is not None and seen["rejected"] is not None
    # heading 90 deg: getTransformFromRPYT(0, 0, 0, -90, 0, 0)^-1 turns east into the ego frame's axis; the speed arrives whole, the up part never
    assert abs(np.linalg.norm(seen["accepted"][:2]) - 1.5) < 0.15 and abs(seen["accepted"][2]) < 0.05, seen
    assert np.linalg.norm(seen["rejected"]) < 0.1, seen
