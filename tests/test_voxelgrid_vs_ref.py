"""SURVEY.md 8a rows 2 and 15 (pcl::VoxelGrid) pinned as far as the reference tree allows.  PCL itself is neither installed nor in the tree, but
the tree vendors ndt_omp, whose pclomp::VoxelGridCovariance::applyFilter (voxel_grid_covariance_omp_impl.hpp:49-330) is PCL's own
VoxelGridCovariance -- carrying, statement for statement, the half of pcl::VoxelGrid::applyFilter the restatement has to get right: getMinMax3D's
box, the int64 overflow guard, min_b / div_b / divb_mul, the key int(floor(x * inverse_leaf) - float(min_b)) . divb_mul, the skip of non-finite
points, per-leaf f32 sums of every field in input order divided by float(count), leaves in ascending key order.  oracle/ref_voxelgrid_cov.cpp
compiles it from where it lies; tests/golden/voxelgrid_vgc.npz holds what it returns on tests/voxelgrid_cases.py.
  * the harness reproduces the recorded vectors where /root/reference is mounted;
  * the oracle's voxel_downsample is BIT-EXACT against them: same leaves, same order, same four f32 centroids (CPU, no reference tree needed);
  * the HIP kernel chain against the same vectors (-m gpu), single scan and batched form.
What stays outside the pin: pcl::VoxelGrid proper sorts an index vector by key (std::sort: the order of the addends INSIDE a voxel is then
implementation-defined; oracle and product fix input order, which is what the pinned class executes), and on the guard it hands back the input
cloud where VoxelGridCovariance hands back an empty one (tested below as the stated difference)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import voxelgrid_cases as vc  # noqa: E402

GOLD = os.path.join(HERE, "golden", "voxelgrid_vgc.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def cases():
    return vc.cases()


def test_reference_filter_reproduces_the_recorded_vectors(gold, cases):
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    import ref_voxelgrid_cov as rv

    if not rv.available():
        pytest.skip("oracle/_ref/libref_voxelgrid_cov.so not built (no /root/reference here)")
    for name, (cloud, leaf, dense) in cases.items():
        r = rv.leaves(cloud, leaf, dense)
        assert (r is None) == bool(gold[name + "/fired"]), name
        if r is None:
            continue
        keys, cnt, cen, mb, db = r
        assert np.array_equal(keys, gold[name + "/keys"]) and np.array_equal(cnt, gold[name + "/counts"]), name
        assert np.array_equal(cen.view(np.uint32), gold[name + "/centroids"].view(np.uint32)), name
        assert np.array_equal(mb, gold[name + "/min_b"]) and np.array_equal(db, gold[name + "/div_b"]), name


def test_oracle_voxelgrid_is_bit_exact_against_the_reference_filter(oracle_mod, gold, cases):
    seen_long = 0
    for name, (cloud, leaf, dense) in cases.items():
        ds = oracle_mod.voxel_downsample(cloud, leaf)
        if bool(gold[name + "/fired"]):
            # the guard: pcl::VoxelGrid returns the input cloud (voxel_grid.hpp), VoxelGridCovariance an empty one -- the CONDITION is the pinned part
            assert len(ds) == len(cloud) and np.array_equal(ds.view(np.uint32), cloud.view(np.uint32)), name
            continue
        want = gold[name + "/centroids"]
        assert len(ds) == len(want), (name, len(ds), len(want))
        assert np.array_equal(ds.view(np.uint32), want.view(np.uint32)), name  # x, y, z AND intensity, leaf order included
        keys = gold[name + "/keys"]
        assert np.all(np.diff(keys) > 0)  # ascending linear index: the order both PCL classes emit
        seen_long = max(seen_long, int(gold[name + "/counts"].max()))
    assert seen_long > 2000  # f32 running sums over thousands of points were part of it


@pytest.mark.gpu
def test_hip_voxelgrid_is_bit_exact_against_the_reference_filter(gold, cases):
    from lsd_amd import lio

    names = [n for n in cases if not bool(gold[n + "/fired"])]
    single = lio.Scan(max_raw=1 << 17, max_ds=1 << 16)
    batch = [lio.Scan(max_raw=1 << 17, max_ds=1 << 16) for _ in names]
    by_leaf = {}
    for n, b in zip(names, batch):
        cloud, leaf, _ = cases[n]
        single.upload(cloud)
        assert single.voxel_downsample(leaf) == len(gold[n + "/centroids"]), n
        assert np.array_equal(single.get_ds().view(np.uint32), gold[n + "/centroids"].view(np.uint32)), n
        b.upload(cloud)
        by_leaf.setdefault(leaf, []).append((n, b))
    for leaf, group in by_leaf.items():  # the batched chain: one set of launches per leaf size
        lio.Scan.voxel_downsample_batch([b for _, b in group], leaf)
        for n, b in group:
            assert np.array_equal(b.get_ds().view(np.uint32), gold[n + "/centroids"].view(np.uint32)), n
    cloud, leaf, _ = cases["guard_fires"]  # the overflow guard on the device: output = input, as pcl::VoxelGrid does
    single.upload(cloud)
    assert single.voxel_downsample(leaf) == len(cloud) and np.array_equal(single.get_ds().view(np.uint32), cloud.view(np.uint32))
