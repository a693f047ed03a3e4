"""Inputs shared by tools/make_golden_localmap.py, tests/test_localmap_vs_ref.py and the GPU tests of rows N3 / N4: a drive's key frames and the
poses the local map is asked for, two overlapping clouds for the map-merge fitness score.  Seeded numpy only."""
import numpy as np

from lsd_amd import synth

LM_POSES = [([3.0, 0.2, 1.8], "first update"), ([7.0, 0.2, 1.8], "moved 4 m: nothing to do"), ([15.0, 0.2, 1.8], "moved 12 m"),
            ([500.0, 0.0, 0.0], "no key frame within 30 m: out of map"), ([62.0, 0.0, 1.8], "nearest key frame 24 m away")]
KEY_FRAME_DISTANCE, LEAF = 3.0, 0.2


def scene():
    return synth.Scene(half=60.0, n_boxes=20, seed=3)


def key_frames(sc=None, n=40, n_az=600):
    """key frames every 2 m along x, clouds already in the map frame (enough points for the 200 000-point cap to cut the concatenation)"""
    sc = sc or scene()
    rng = np.random.default_rng(3)
    frames, poses = [], []
    for k in range(n):
        pos = np.array([-40.0 + 2.0 * k, rng.uniform(-0.5, 0.5), 1.8])
        q = synth.quat_from_rotvec([0, 0, rng.uniform(-0.2, 0.2)])
        raw, _ = synth.make_scan(sc, pos, q, seed=100 + k, n_az=n_az, fov_deg=(-24.8, 2.0))
        w = raw.copy()
        w[:, :3] = (raw[:, :3].astype(np.float64) @ synth.quat_to_R(q).T + pos).astype(np.float32)
        frames.append(w)
        poses.append(pos.astype(np.float32))
    return frames, poses


def overlap_case(sc=None):
    """cloud1 (part of it below the 0.5 m floor), cloud2 = a noisy, displaced sample of it + far outliers, the relative pose"""
    sc = sc or scene()
    rng = np.random.default_rng(21)
    cloud1 = sc.sample_surface(300_000, seed=33, sigma=0.01)
    cloud1 = cloud1[np.linalg.norm(cloud1[:, :2], axis=1) < 120.0][:60_000]
    cloud1[:, 2] += 0.45  # the ground ends up just below the 0.5 m floor of the filter, boxes and walls above it
    T = np.eye(4)
    T[:3, :3] = synth.quat_to_R(synth.quat_from_rotvec([0.0, 0.01, 0.2]))
    T[:3, 3] = [2.0, -1.0, 0.05]
    Ti = np.linalg.inv(T)
    pick = cloud1[rng.choice(len(cloud1), 6000, replace=False)]
    cloud2 = pick.copy()
    cloud2[:, :3] = pick[:, :3] @ Ti[:3, :3].T + Ti[:3, 3] + rng.normal(0, 0.1, (6000, 3))
    cloud2 = np.concatenate([cloud2, np.concatenate([rng.uniform(-150, 150, (800, 2)), rng.uniform(-2, 40, (800, 1)), np.zeros((800, 1))], 1)]).astype(np.float32)
    return np.ascontiguousarray(cloud1), cloud2, T


def digest(cloud):
    """order-sensitive fingerprint of a cloud's bits (the golden file keeps this instead of 200 000 points)"""
    u = np.ascontiguousarray(cloud, np.float32).view(np.uint32).astype(np.uint64).reshape(-1)
    w = (np.arange(len(u), dtype=np.uint64) * np.uint64(2654435761) + np.uint64(1)) & np.uint64(0xFFFFFFFF)
    return np.array([len(cloud), int((u * w).sum() & np.uint64(0xFFFFFFFFFFFFFFFF)) >> 1, int(u.sum())], dtype=np.int64)
