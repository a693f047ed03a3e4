import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle  # oracle/oracle.py (test infrastructure)

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def scene():
    from lsd_amd import synth

    return synth.Scene(half=100.0, n_boxes=40, seed=1)


@pytest.fixture(scope="session")
def small_world(scene):
    """a 300k-point map of the scene and one 64x600 scan near the origin with a perturbed initial guess"""
    import numpy as np
    from lsd_amd import synth

    map_pts = scene.sample_surface(300_000, seed=2, sigma=0.01)
    true_pos = np.array([1.0, -2.0, 1.8])
    true_q = synth.quat_from_rotvec([0.0, 0.0, 0.3])
    raw, t = synth.make_scan(scene, true_pos, true_q, seed=7, n_az=600)
    g_pos, g_q = synth.perturb_pose(true_pos, true_q, seed=11, max_t=0.2, max_deg=1.5)
    return dict(map=map_pts, raw=raw, true_pos=true_pos, true_q=true_q, guess_pos=g_pos, guess_q=g_q)


@pytest.fixture(params=["host_loop", "device_loop"])
def loop_mode(request, monkeypatch):
    """where the iterate loop of a single engine's update runs: engines created inside the test take the mode from LIO_DEVICE_LOOP
    (lio_engine_set_device_loop changes it per engine); the results must not depend on it"""
    monkeypatch.setenv("LIO_DEVICE_LOOP", "1" if request.param == "device_loop" else "0")
    return request.param
