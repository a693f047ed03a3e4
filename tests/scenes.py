"""Scenes the box-room fixtures never produce: open ground (two unobservable directions), ground + one box (a direction just on
either side of the degeneracy decision of laserMapping.cpp:934-980), and scans so sparse that N_eff < 23 (esekfom.hpp:1715-1744)."""
import numpy as np

from lsd_amd import synth


def ground_scene(boxes=()):
    """ground plane z = 0 without perimeter walls, plus the given axis-aligned boxes [(lo3, hi3), ...]"""
    sc = synth.Scene(half=100.0, n_boxes=0, seed=1, wall_h=0.0)
    sc.lo = np.array([b[0] for b in boxes], float).reshape(-1, 3)
    sc.hi = np.array([b[1] for b in boxes], float).reshape(-1, 3)
    return sc


# name -> (boxes, what the third pass of the filter sees in the weak horizontal direction)
DEGENERATE_CASES = {
    "open_ground": ((), "two directions with contri < 250 and strong < 50: both projected out"),
    "box_6x3": ((([10, -3, 0], [16, 3, 3]),), "contri < 250 but strong >= 50: kept (the && of laserMapping.cpp:965)"),
    "box_12x4": ((([10, -6, 0], [22, 6, 4]),), "contri >= 250 while the eigenvalue bound (< 250) does not decide: evaluated, kept"),
    "box_20x6": ((([10, -10, 0], [30, 10, 6]),), "eigenvalue bound >= 250 for that direction; the other horizontal one is still degenerate"),
}


def degenerate_case(name, n_az=600, n_beams=64, pose_seed=3):
    boxes, _ = DEGENERATE_CASES[name]
    sc = ground_scene(boxes)
    mp = sc.sample_surface(300_000, seed=2, sigma=0.01)
    mp = np.ascontiguousarray(mp[(np.abs(mp[:, 0]) < 45) & (np.abs(mp[:, 1]) < 45)])
    pos = np.array([1.0, -2.0, 1.8])
    q = synth.quat_from_rotvec([0, 0, 0.3])
    raw, _ = synth.make_scan(sc, pos, q, seed=7, n_az=n_az, n_beams=n_beams, max_range=40.0)
    gp, gq = synth.perturb_pose(pos, q, seed=pose_seed, max_t=0.2, max_deg=1.5)
    return dict(map=mp, raw=raw, true_pos=pos, true_q=q, guess=synth.state_from_pose(gp, gq))


# ---- BASELINE.json configurations at full size ---------------------------------------------------
def config_scene():
    """200 m x 200 m ground + 40 boxes + 4 walls, seed 1"""
    return synth.Scene(half=100.0, n_boxes=40, seed=1)


def config_scan(scene, seed, fov_deg=(-25.0, 15.0), max_range=100.0):
    """one 64 x 1875 scan from a random pose near the origin and a prior within 0.3 m / 2 deg of it (config 2: seeds 1000..1099)"""
    rng = np.random.default_rng(seed)
    pos = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), 1.8])
    q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
    raw, _ = synth.make_scan(scene, pos, q, seed=seed, n_az=1875, fov_deg=fov_deg, max_range=max_range)
    gp, gq = synth.perturb_pose(pos, q, seed=seed + 5000, max_t=0.3, max_deg=2.0)
    return dict(raw=raw, pos=pos, q=q, guess=synth.state_from_pose(gp, gq))


# ---- device buffers from the HIP runtime the library itself is linked against (torch ships its own copy of the runtime; two in one
# process do not see each other's devices) --------------------------------------------------------------------------------------------
_hip_rt = None


def to_device(a):
    """copy a numpy array to the GPU; returns the device address (int).  The allocation lives until the process ends (tests only)."""
    import ctypes as C

    global _hip_rt
    if _hip_rt is None:
        _hip_rt = C.CDLL("libamdhip64.so")
        _hip_rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip_rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    a = np.ascontiguousarray(a)
    d = C.c_void_p()
    assert _hip_rt.hipMalloc(C.byref(d), max(a.nbytes, 16)) == 0
    assert _hip_rt.hipMemcpy(d, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0  # hipMemcpyHostToDevice
    return d.value
