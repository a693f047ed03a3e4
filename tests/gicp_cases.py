"""Seeded cloud pairs for the fine-matcher (GICP) parity tests: two scans of one scene from nearby poses, voxel-thinned (the reference
downsamples to 0.2 .. 0.5 m before it calls the matcher, overlap_merge.hpp / loop_detector.hpp), the source expressed in its own frame."""
import numpy as np

from lsd_amd import synth

CASES = {
    # name: (scene seed, n_az, n_beams, leaf, k, max_corr_dist, guess error (m, deg))
    "room_small": (3, 300, 16, 0.5, 20, 2.0, (0.3, 2.0)),
    "room_fine": (4, 400, 32, 0.4, 20, 0.5, (0.15, 1.0)),
    "room_k10": (5, 300, 16, 0.5, 10, 2.0, (0.3, 2.0)),
}


def _thin(pts, leaf):
    """one point per leaf-sized cell (the first in input order): a stand-in for the caller's voxel filter, keeps exact f32 coordinates"""
    key = np.floor(pts[:, :3] / leaf).astype(np.int64)
    _, first = np.unique(key, axis=0, return_index=True)
    return np.ascontiguousarray(pts[np.sort(first)])


def _pose(pos, q):
    T = np.eye(4)
    T[:3, :3] = synth.quat_to_R(q)
    T[:3, 3] = pos
    return T


def make(name):
    seed, n_az, n_beams, leaf, k, maxd, (et, edeg) = CASES[name]
    sc = synth.Scene(half=25.0, n_boxes=12, seed=seed)
    pa, qa = np.array([0.5, -1.0, 1.8]), synth.quat_from_rotvec([0, 0, 0.2])
    pb, qb = np.array([1.6, -0.4, 1.8]), synth.quat_from_rotvec([0.01, -0.02, 0.35])
    ra, _ = synth.make_scan(sc, pa, qa, seed=seed + 10, n_az=n_az, n_beams=n_beams, max_range=40.0)
    rb, _ = synth.make_scan(sc, pb, qb, seed=seed + 11, n_az=n_az, n_beams=n_beams, max_range=40.0)
    tgt = _thin(ra[:, :4].astype(np.float32), leaf)
    src = _thin(rb[:, :4].astype(np.float32), leaf)
    truth = np.linalg.inv(_pose(pa, qa)) @ _pose(pb, qb)  # source frame -> target frame
    rng = np.random.default_rng(seed + 20)
    ax, dt = rng.normal(size=3), rng.normal(size=3)
    err = _pose(dt / np.linalg.norm(dt) * et, synth.quat_from_rotvec(ax / np.linalg.norm(ax) * np.deg2rad(edeg)))
    return dict(target=tgt, source=src, truth=truth, guess=truth @ err, k=k, max_corr_dist=maxd)
