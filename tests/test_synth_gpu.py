"""The torch generators bench.py uses for BASELINE configs 3 and 4 (lsd_amd/synth_gpu.py) against the numpy generators the tests use
(lsd_amd/synth.py): same scene, same rays, same trajectory model.  Runs on torch's CPU device here."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_sweeper_ranges_equal_the_numpy_ray_caster():
    from lsd_amd import synth, synth_gpu

    scene = synth.Scene(half=120.0, n_boxes=60, seed=3, box_size=(4.0, 30.0), keep_clear=8.0)
    traj = synth_gpu.Lawnmower(p0=(-80.0, -60.0, 1.8), speed=12.0, half_len=80.0, spacing=40.0, rows=3)
    sw = synth_gpu.Sweeper(scene, traj, torch.device("cpu"), n_beams=16, n_az=90, fov_deg=(-24.8, 2.0), max_range=100.0)
    for t_beg in (0.0, 3.0, 9.7, 20.05):
        r = sw.ranges(t_beg).numpy()
        d_l, frac = synth.lidar_dirs(16, 90, (-24.8, 2.0))
        stamp = np.round(frac * 0.1 * 1e6).astype(np.uint32)
        t_az = t_beg + stamp[::16].astype(np.float64) * 1e-6
        Rw, ow = traj.R(t_az), traj.pos(t_az)
        d_w = np.einsum("aij,abj->abi", Rw, d_l.reshape(90, 16, 3)).reshape(-1, 3)
        want = scene.raycast(np.repeat(ow, 16, axis=0), d_w, 100.0)
        assert np.array_equal(np.isfinite(r), np.isfinite(want))
        ok = np.isfinite(want)
        assert ok.sum() > 300 and np.abs(r[ok] - want[ok]).max() < 1e-9
    pts, st = sw.sweep(30)
    assert pts.dtype == np.float32 and pts.shape[1] == 4 and st.dtype == np.uint32 and len(st) == len(pts) and np.all(np.diff(st.astype(np.int64)) >= 0)


def test_lawnmower_is_a_smooth_drive_and_its_imu_matches_the_scalar_generator():
    from lsd_amd import synth, synth_gpu

    traj = synth_gpu.Lawnmower(p0=(-80.0, -60.0, 1.8), speed=12.0, half_len=80.0, spacing=40.0, rows=3)
    t = np.linspace(0.0, traj.duration(), 4000)
    p = traj.pos(t)
    v = np.linalg.norm(np.diff(p, axis=0), axis=1) / np.diff(t)
    assert v.max() < 12.0 * 1.01 + 0.2 and np.abs(np.diff(v)).max() < 0.5      # no jumps in position, speed ramps up smoothly
    assert abs(p[-1, 1] - (-60.0 + 2 * 40.0)) < 1e-3                            # ends on the third row
    ts, gyr, acc = synth_gpu.imu_stream(traj, 1.0, 9.0, rate=50.0)
    ref = synth.imu_stream(traj, 1.0, 9.0, rate=50.0)
    assert len(ref) == len(ts)
    for k in range(0, len(ts), 37):
        assert abs(ref[k][0] - ts[k]) < 1e-12 and np.abs(ref[k][1] - gyr[k]).max() < 1e-9 and np.abs(ref[k][2] - acc[k]).max() < 1e-6


def test_surface_sampler_covers_the_scene_like_the_numpy_one():
    from lsd_amd import synth, synth_gpu

    scene = synth.Scene(half=50.0, n_boxes=10, seed=1)
    a = synth_gpu.sample_surface(scene, 200_000, torch.device("cpu"), seed=2, sigma=0.01, chunk=70_000).numpy()
    b = scene.sample_surface(200_000, seed=2, sigma=0.01)
    assert a.shape == b.shape and a.dtype == np.float32
    # the same surfaces: share of ground points, of points above 1 m, bounding box
    for f in (lambda p: np.mean(np.abs(p[:, 2]) < 0.05), lambda p: np.mean(p[:, 2] > 1.0)):
        assert abs(f(a) - f(b)) < 0.01
    assert np.abs(a[:, :3].min(0) - b[:, :3].min(0)).max() < 0.2 and np.abs(a[:, :3].max(0) - b[:, :3].max(0)).max() < 0.2
