"""The localisation mode's constant-velocity motion compensation, undistortPoints(delta_pose, points, scan_period)
(slam/common/slam_utils.cpp:163-191): the oracle against the reference's OWN slam_utils.cpp compiled whole
(oracle/_ref/libref_slam_utils.so) and against the fixture recorded from it (tests/golden/undistort_delta.npz, travels to
the GPU box); on the GPU the HIP kernel (lio_scan_undistort_delta) against both.  f32 throughout: bit-exact on the CPU; the
device evaluates sin / cos of the half angle in f64 and rounds, so a point may differ by an ulp where that rounding and a
correctly rounded sinf / cosf part ways."""
import os

import numpy as np
import pytest

import ref_slam_utils as rs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "undistort_delta.npz")


def _same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_oracle_matches_golden(oracle_mod):
    g = np.load(GOLD)
    for D, ref in zip(g["deltas"], g["out"]):
        assert _same(oracle_mod.undistort_delta(g["points"], g["stamp_us"], D, float(g["scan_period"])), ref)
    # no rotation and no translation: the cloud is returned bit for bit
    assert _same(oracle_mod.undistort_delta(g["points"], g["stamp_us"], np.eye(4), 0.1), g["points"])
    assert len(oracle_mod.undistort_delta(np.zeros((0, 4)), np.zeros(0), np.eye(4), 0.1)) == 0


@pytest.mark.skipif(not rs.available(), reason="oracle/_ref/libref_slam_utils.so not built (needs /root/reference)")
def test_oracle_matches_the_references_slam_utils(oracle_mod):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from make_golden import undistort_delta_cases
    import slam_wrapper  # the compiled module (lidar-slam-detection_amd/python/slam_wrapper.*.so)

    rng = np.random.default_rng(11)
    for rep in range(4):
        pts, st, deltas = undistort_delta_cases(rng, n=4000)
        for period in (0.1, 0.05):
            for D in deltas:
                assert _same(oracle_mod.undistort_delta(pts, st, D, period), rs.undistort_delta(pts, st, D, period))
    # a rotation by pi about each axis: trace <= 0, the other branch of Eigen's matrix -> quaternion conversion
    for ax in range(3):
        D = -np.eye(4, dtype=np.float32)
        D[ax, ax] = D[3, 3] = 1
        D[:3, 3] = [0.3, -0.2, 0.1]
        assert _same(oracle_mod.undistort_delta(pts, st, D, 0.1), rs.undistort_delta(pts, st, D, 0.1))
    # getTransformFromRPYT (slam_utils.cpp:89-96) as restated for the slam_wrapper boundary
    for _ in range(50):
        a = rng.uniform(-180, 180, 6)
        assert np.abs(slam_wrapper._transform_from_rpyt(*a) - rs.transform_from_rpyt(*a)).max() < 1e-14 * max(1.0, np.abs(a[:3]).max())


GOLD_POSES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "undistort_poses.npz")


def _pose_cases(g):
    for k in range(int(g["n_cases"])):
        yield k, int(g[f"c{k}_header"]), g[f"c{k}_pose_stamps"], g[f"c{k}_pose_T"], g[f"c{k}_stamp_us"], g[f"c{k}_out"]


def test_oracle_pose_list_matches_golden(oracle_mod):
    """undistortPoints(poses, points) (slam_utils.cpp:193-228): the sequential walk over the cloud, against the fixture recorded from the
    reference's own code -- ordered and unordered clouds, 2 .. 12 poses, a pose before the header stamp, stamps past the last pose"""
    g = np.load(GOLD_POSES)
    moved = 0
    for k, header, ps, Ts, st, ref in _pose_cases(g):
        out = oracle_mod.undistort_poses(g["points"], st, header, ps, Ts)
        assert _same(out, ref), k
        moved += int((np.abs(out[:, :3] - g["points"][:, :3]).max(1) > 0).sum())
    assert moved > 2500


@pytest.mark.skipif(not rs.available(), reason="oracle/_ref/libref_slam_utils.so not built (needs /root/reference)")
def test_oracle_pose_list_matches_the_references_slam_utils(oracle_mod):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from make_golden import undistort_poses_cases

    rng = np.random.default_rng(23)
    for rep in range(3):
        pts, cases = undistort_poses_cases(rng, n=5000)
        for c in cases:
            a = rs.undistort_poses(pts, c["stamp_us"], c["header"], c["pose_stamps"], c["pose_T"])
            assert _same(oracle_mod.undistort_poses(pts, c["stamp_us"], c["header"], c["pose_stamps"], c["pose_T"]), a)


@pytest.mark.gpu
def test_hip_undistort_poses(oracle_mod):
    from lsd_amd import capi, lio

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box")
    g = np.load(GOLD_POSES)
    pts = g["points"]
    sc = lio.Scan(max_raw=1 << 16, max_ds=1 << 14)
    refused = 0
    for k, header, ps, Ts, st, ref in _pose_cases(g):
        sc.upload(pts)
        limits = (ps[1:] - np.uint64(header)).astype(np.uint64)  # unsigned wrap-around, as the reference computes them
        if np.any(np.diff(limits.astype(np.float64)) < 0):
            with pytest.raises(capi.LioError):  # interval ends that decrease: refused, not guessed
                sc.undistort_poses(st, header, ps, Ts)
            refused += 1
            continue
        sc.undistort_poses(st, header, ps, Ts)
        out = sc.download_raw()
        same_untouched = (ref[:, :3] == pts[:, :3]).all(1)
        assert np.array_equal(out[same_untouched].view(np.uint32), ref[same_untouched].view(np.uint32)), k  # points the walk never reached
        d = np.abs(out[:, :3] - ref[:, :3])
        scale = np.maximum(np.abs(ref[:, :3]).max(axis=1, keepdims=True), 1e-3)
        assert (d / (scale * 2.0 ** -23)).max() <= 8.0 and (d > 0).mean() < 1e-2, (k, (d / (scale * 2.0 ** -23)).max(), (d > 0).mean())
    assert refused <= 1
    sc.close()


@pytest.mark.gpu
def test_hip_undistort_delta(oracle_mod):
    import ctypes as C

    from lsd_amd import capi, lio

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box")
    g = np.load(GOLD)
    pts, st = g["points"], g["stamp_us"]
    sc = lio.Scan(max_raw=1 << 16, max_ds=1 << 14)
    worst, differing = 0.0, 0
    for D, ref in zip(g["deltas"], g["out"]):
        sc.upload(pts)
        sc.undistort_delta(st, D, float(g["scan_period"]))
        out = sc.download_raw()
        assert out.shape == ref.shape and np.array_equal(out[:, 3], ref[:, 3])
        assert np.array_equal(np.isnan(out[:, :3]), np.isnan(ref[:, :3]))
        ok = ~np.isnan(ref[:, 0])
        d = np.abs(out[ok, :3] - ref[ok, :3])
        scale = np.abs(ref[ok, :3]).max(axis=1, keepdims=True)  # a coordinate near zero is a difference of terms of the point's size
        worst, differing = max(worst, float((d / (scale * 2.0 ** -23)).max())), differing + int((d > 0).sum())
        assert d.max() < 2e-5
    # the rotation's sin / cos half angle is a rounded f64 value here and glibc's sinf / cosf there: about one coordinate in 200
    # moves, by a few ulp of the point's largest coordinate (2.9 measured)
    assert worst <= 8.0 and differing <= 1e-2 * pts.shape[0] * 3 * len(g["deltas"]), (worst, differing)
    # stamps already on the device, cloud owned by the caller: same result, the caller's cloud is not written
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes, hip.hipMemcpy.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    d_pts, d_st = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_pts), pts.nbytes) == 0 and hip.hipMalloc(C.byref(d_st), st.nbytes) == 0
    assert hip.hipMemcpy(d_pts, pts.ctypes.data, pts.nbytes, 1) == 0 and hip.hipMemcpy(d_st, st.ctypes.data, st.nbytes, 1) == 0
    D = g["deltas"][1]
    sc.upload(pts)
    sc.undistort_delta(st, D, 0.1)
    a = sc.download_raw()
    sc.set_device(d_pts.value, len(pts))
    sc.undistort_delta(d_st.value, D, 0.1, on_device=True)
    b = sc.download_raw()
    assert _same(a, b)
    back = np.zeros_like(pts)
    assert hip.hipMemcpy(back.ctypes.data, d_pts, pts.nbytes, 2) == 0 and _same(back, pts)
    # the compensated cloud is what the downsample then reads
    n_ds = sc.voxel_downsample(0.5)
    assert n_ds == len(oracle_mod.voxel_downsample(b, 0.5)) > 100
    hip.hipFree(d_pts), hip.hipFree(d_st)
    sc.close()
