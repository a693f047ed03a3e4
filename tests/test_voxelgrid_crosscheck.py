"""A second, independent restatement of pcl::VoxelGrid::applyFilter (PCL 1.9.1 voxel_grid.hpp -- not in the reference tree; since round 3 the
oracle is pinned to the PCL-derived filter the tree does hold, tests/test_voxelgrid_vs_ref.py, and this file stays as a cross-check) written in numpy along another path than oracle/lio_oracle.cpp:
vectorised floor / lexicographic sort by (iz, iy, ix) instead of the linear index, f64 means instead of f32 running sums.  It cross-checks
what does not depend on summation order: which points share a voxel, the output ORDER (ascending linear index = z-major, then y, then x), the
point count, the int32 overflow guard, the handling of non-finite points -- and the centroids to 1e-5 m."""
import numpy as np
import pytest


def voxelgrid_numpy(pts, leaf):
    p = np.asarray(pts, np.float32).reshape(-1, 4)
    ok = np.isfinite(p[:, :3]).all(1)
    q = p[ok]
    if len(q) == 0:
        return q
    inv = np.float32(1.0) / np.float32(leaf)
    mn, mx = q[:, :3].min(0), q[:, :3].max(0)
    d = ((mx - mn) * inv).astype(np.int64) + 1          # getMinMax3D + the (dx * dy * dz) > INT32_MAX guard of applyFilter
    if int(d[0]) * int(d[1]) * int(d[2]) > np.iinfo(np.int32).max:
        return p                                        # "Leaf size is too small": output = input
    ijk = np.floor(q[:, :3] * inv).astype(np.int64)      # floor(p * inv_leaf) in f32, then integer
    order = np.lexsort((np.arange(len(q)), ijk[:, 0], ijk[:, 1], ijk[:, 2]))  # z-major, then y, then x; stable in the input index
    s = ijk[order]
    head = np.ones(len(q), bool)
    head[1:] = (s[1:] != s[:-1]).any(1)
    starts = np.flatnonzero(head)
    cnt = np.diff(np.append(starts, len(q)))
    sums = np.add.reduceat(q[order].astype(np.float64), starts, axis=0)
    return (sums / cnt[:, None]).astype(np.float32)


@pytest.mark.parametrize("seed,leaf,scale", [(0, 0.5, 30.0), (1, 0.2, 8.0), (2, 0.5, 120.0), (3, 1.0, 400.0)])
def test_second_restatement_agrees(oracle_mod, seed, leaf, scale):
    rng = np.random.default_rng(seed)
    p = np.concatenate([rng.normal(0, scale, (40000, 3)) * [1, 1, 0.05], rng.uniform(0, 255, (40000, 1))], 1).astype(np.float32)
    p[rng.integers(0, len(p), 20), rng.integers(0, 3, 20)] = np.nan     # non-finite points are skipped
    p[rng.integers(0, len(p), 5), 0] = np.inf
    p[1000:1200, :3] = p[1000, :3]                                        # 200 duplicates in one voxel
    a = oracle_mod.voxel_downsample(p, leaf)
    b = voxelgrid_numpy(p, leaf)
    assert len(a) == len(b)
    # (f32 running sums there, f64 means here: a voxel of 200 points at 100 m differs by a few f32 ulp of the coordinate times sqrt(count))
    assert np.abs(a[:, :3] - b[:, :3]).max() < 1e-4 * max(1.0, scale / 30.0) and np.abs(a[:, 3] - b[:, 3]).max() < 1e-2
    # same voxel per output slot (the order), checked on the integer cell of the centroids
    assert np.array_equal(np.floor(a[:, :3] / np.float32(leaf)), np.floor(b[:, :3] / np.float32(leaf)))


def test_second_restatement_overflow_guard_and_negative_cells(oracle_mod):
    p = np.array([[-0.1, -0.1, -0.1, 1], [-0.4, -0.3, -0.2, 3], [0.1, 0.1, 0.1, 5], [0.6, -0.6, 0.0, 7], [0.6, -0.6, 0.01, 9]], np.float32)
    a, b = oracle_mod.voxel_downsample(p, 0.5), voxelgrid_numpy(p, 0.5)
    assert np.allclose(a, b, atol=1e-6) and len(a) == 3
    far = np.array([[0, 0, 0, 1], [1e6, 1e6, 1e6, 2]], np.float32)  # 2e6^3 cells of 1 m > INT32_MAX: the filter returns its input
    a, b = oracle_mod.voxel_downsample(far, 1.0), voxelgrid_numpy(far, 1.0)
    assert np.array_equal(a, far) and np.array_equal(b, far)
