"""Inputs of the VoxelGrid pin (tests/test_voxelgrid_vs_ref.py, tools/make_golden_voxelgrid.py): seeded clouds that reach every statement of
pcl::VoxelGrid::applyFilter's first half -- the bounding box, the overflow guard, the voxel key, the skip of non-finite points, the per-voxel
f32 sums -- as the reference tree's own PCL-derived filter (pclomp::VoxelGridCovariance, oracle/ref_voxelgrid_cov.cpp) executes them."""
import numpy as np


def cases():
    """name -> (cloud f32 (n, 4), leaf, is_dense)"""
    from lsd_amd import synth

    out = {}
    sc = synth.Scene(half=40.0, n_boxes=12, seed=21)
    raw, _ = synth.make_scan(sc, np.array([0.3, 0.8, 1.7]), synth.quat_from_rotvec([0, 0, 0.2]), seed=23, n_az=300)
    raw = raw[:, :4].astype(np.float32)
    for leaf in (0.5, 0.2, 2.0):
        out[f"scan_leaf_{leaf}"] = (raw, leaf, True)
    rng = np.random.default_rng(5)
    p = np.concatenate([rng.normal(0, 30.0, (20000, 3)) * [1, 1, 0.05], rng.uniform(0, 255, (20000, 1))], 1).astype(np.float32)
    p[rng.integers(0, len(p), 20), rng.integers(0, 3, 20)] = np.nan  # skipped in a cloud that is not dense
    p[rng.integers(0, len(p), 5), 0] = np.inf
    p[1000:1200, :3] = p[1000, :3]                                    # 200 duplicates in one voxel
    p[3000:7000, :3] = p[3000, :3] + rng.uniform(0, 0.3, (4000, 3)).astype(np.float32)  # a voxel with thousands of points: long f32 sums
    out["random_with_nonfinite"] = (p, 0.5, False)
    q = p[np.isfinite(p[:, :3]).all(1)].copy()
    q[:, :3] += np.array([9000.0, -7000.0, 300.0], np.float32)       # large coordinates: f32 products next to integers
    out["far_from_the_origin"] = (q, 0.4, True)
    g = np.zeros((4096, 4), np.float32)                               # points exactly ON voxel faces, both signs
    ii = np.arange(4096)
    g[:, 0] = ((ii % 16) - 8) * 0.5
    g[:, 1] = (((ii // 16) % 16) - 8) * 0.25
    g[:, 2] = ((ii // 256) - 8) * 0.125
    g[:, 3] = ii % 7
    out["on_the_faces"] = (g, 0.5, True)
    out["one_point"] = (np.array([[1.25, -3.5, 0.75, 9.0]], np.float32), 0.5, True)
    h = rng.uniform(-1, 1, (1000, 4)).astype(np.float32)              # 4 km x 4 km x 400 m at 1 cm: the int32 guard fires
    h[:, :3] *= np.array([2000.0, 2000.0, 200.0], np.float32)
    out["guard_fires"] = (h, 0.01, True)
    k = h.copy()                                                      # ... and just does not: 2 000 x 2 000 x 200 cells at 2 m
    out["guard_just_passes"] = (k, 2.0, True)
    return out
