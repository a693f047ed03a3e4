"""Voxels with thousands of points (a wall a metre from the sensor, the ground ring under it): their centroids are PCL's SEQUENTIAL f32 sums, which
csrc/voxelgrid.hip forms one coordinate per wave, 256 points per step, by integer arithmetic inside the running sum's binade
(monster_component_sum; the rule is held against the plain loop in tests/test_seqsum_math.py).  Here the kernels against the oracle's plain loops,
bit for bit, on clouds built to hit every branch: ties, binade crossings, the voxel at the origin, run lengths around the queue thresholds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(rng, kind, n_monster, n_back=20000):
    back = np.column_stack([rng.uniform(-40, 40, n_back), rng.uniform(-40, 40, n_back), rng.uniform(-2, 6, n_back), rng.uniform(0, 255, n_back)])
    lo = {"pos": (3.5, 0.0, -2.0), "neg": (-7.5, -3.0, -1.5), "origin": (0.0, 0.0, 0.0)}[kind.split("/")[0]]
    p = np.column_stack([rng.uniform(lo[0], lo[0] + 0.5, n_monster), rng.uniform(lo[1], lo[1] + 0.5, n_monster), rng.uniform(lo[2], lo[2] + 0.5, n_monster),
                         rng.uniform(0, 255, n_monster)])
    if kind.endswith("/grid"):      # coordinates on a coarse grid: x / q is an exact tie again and again
        p[:, :3] = np.floor(p[:, :3] * 2048) / 2048
        p[:, 3] = np.floor(p[:, 3])
    if kind.startswith("origin"):
        p[:5, :3] = [[0.0, 0.0, 0.0], [1e-30, 0.25, 0.25], [0.25, 1e-42, 0.25], [0.25, 0.25, 0.0], [0.49999997, 0.49999997, 0.49999997]]
    cloud = np.concatenate([back, p]).astype(np.float32)
    return cloud[rng.permutation(len(cloud))]


@pytest.mark.parametrize("kind,n_monster", [("pos", 60000), ("neg", 9000), ("origin", 5000), ("pos/grid", 30000), ("neg/grid", 2049), ("pos", 2048), ("pos", 2047),
                                            ("origin/grid", 2304)])
def test_monster_voxels_are_summed_exactly(oracle_mod, kind, n_monster):
    from lsd_amd import capi, lio

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device")
    rng = np.random.default_rng(len(kind) * 1000 + n_monster)
    cloud = _cloud(rng, kind, n_monster)
    want = oracle_mod.voxel_downsample(cloud, 0.5)
    s = lio.Scan(max_raw=1 << 17, max_ds=60000)
    s.upload(cloud)
    n = s.voxel_downsample(0.5)
    got = s.get_ds()
    assert n == len(want)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(1))[:5]
    # twice on the same buffers (the queues are re-armed), then a cloud without a monster
    assert s.voxel_downsample(0.5) == n and np.array_equal(s.get_ds().view(np.uint32), want.view(np.uint32))
    small = cloud[: 15000]
    s.upload(small)
    s.voxel_downsample(0.5)
    assert np.array_equal(s.get_ds().view(np.uint32), oracle_mod.voxel_downsample(small, 0.5).view(np.uint32))


def test_monster_voxels_in_the_batched_chain(oracle_mod):
    """several scans per launch, some with two monsters and hundreds of long runs, one without any"""
    from lsd_amd import lio

    rng = np.random.default_rng(77)
    clouds = []
    for k, (kind, n_m) in enumerate([("pos", 40000), ("neg/grid", 12000), ("origin", 3000), ("pos", 100)]):
        c = _cloud(rng, kind, n_m, n_back=30000)
        if k == 0:  # a second monster and a band of long runs
            extra = np.column_stack([rng.uniform(-6.0, -5.5, 7000), rng.uniform(2.0, 2.5, 7000), rng.uniform(-1.0, -0.5, 7000), rng.uniform(0, 255, 7000)])
            band = np.column_stack([rng.uniform(5, 15, 20000), rng.uniform(1.0, 1.5, 20000), rng.uniform(-2.0, -1.5, 20000), rng.uniform(0, 255, 20000)])
            c = np.concatenate([c, extra.astype(np.float32), band.astype(np.float32)])
            c = c[rng.permutation(len(c))]
        clouds.append(np.ascontiguousarray(c, np.float32))
    scans = [lio.Scan(max_raw=1 << 17, max_ds=60000) for _ in clouds]
    for s, c in zip(scans, clouds):
        s.upload(c)
    ns = lio.Scan.voxel_downsample_batch(scans, 0.5)
    for s, c, n in zip(scans, clouds, ns):
        want = oracle_mod.voxel_downsample(c, 0.5)
        assert n == len(want)
        assert np.array_equal(s.get_ds().view(np.uint32), want.view(np.uint32))
