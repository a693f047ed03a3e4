"""The batched, device-resident engine (lio_batch_*: B scans per launch, the batched 16-lanes-per-query kNN, the filter loop of
esekfom.hpp:1619-1931 on the device) against the oracle and against the per-scan engine it replaces in throughput mode."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenes  # noqa: E402

pytestmark = pytest.mark.gpu


def _dev():
    from lsd_amd import capi

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device")


# ---- whole registrations ------------------------------------------------------------------------------------------------------
def _jobs(scene, seeds, dev_tensors, fov=(-25.0, 15.0), max_range=100.0):
    jobs, meta = [], []
    from lsd_amd import lio

    P0 = lio.init_cov()
    for k, seed in enumerate(seeds):
        sc = scenes.config_scan(scene, seed, fov_deg=fov, max_range=max_range)
        jobs.append(dict(dptr=scenes.to_device(sc["raw"]), n=len(sc["raw"]), t=1.0 + 0.1 * k, state=sc["guess"], cov=P0))
        meta.append(sc)
    return jobs, meta


def test_batch_matches_oracle_and_per_scan_engine(oracle_mod):
    """20 scans (more than fit one round of 2 groups x 4 slots, one of them empty, one with three points) through lio_batch_process:
    same return codes, downsampled sizes, pass counts as the per-scan engine path; poses equal to the oracle's.  The jobs are independent
    scans (lio_scan_job.flags == 0): every slot forgets its neighbour cache before a job (laserMapping.cpp:1045-1047), so the results do not
    depend on which scan the slot registered before -- repeated calls and other (slots, groups) geometries give the same BITS."""
    _dev()
    from lsd_amd import lio, synth

    scene = scenes.config_scene()
    mp = scene.sample_surface(1_000_000, seed=2, sigma=0.01)
    the_map = lio.Map(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=1_000_000)
    the_map.add(mp)
    keep = []
    jobs, meta = _jobs(scene, range(3000, 3020), keep)
    jobs[7]["n"] = 0           # "FastLio undistort points is empty"
    jobs[11]["n"] = 3          # fewer than five downsampled points
    P0 = lio.init_cov()
    # the per-scan engine on the same shared map
    eng = lio.Engine(max_raw=1 << 18, max_ds=100000, shared_map=the_map)
    eng.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)
    rc1, res1 = lio.process_batch([eng], jobs)
    assert rc1 == 0
    b = lio.Batch(the_map, n_slots=4, n_groups=2)
    rc2, res2 = b.process(jobs)
    assert rc2 == 0
    o = oracle_mod.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=8)
    o.map_add(mp)
    o.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)

    def oracle_run(k):
        sc = meta[k]
        o.set_state(sc["guess"])
        o.set_cov(P0)
        o.set_ds(oracle_mod.voxel_downsample(sc["raw"][: jobs[k]["n"]], 0.5))
        return o.update(), o.get_state()

    worst = 0.0
    for k, (a, c) in enumerate(zip(res1, res2)):
        assert (a["rc"], a["n_ds"]) == (c["rc"], c["n_ds"]), (k, a, c)
        if a["rc"] != 3:
            assert k in (7, 11) and a["rc"] == 2
            continue
        sc = meta[k]
        o.reset_cache()
        lo, so = oracle_run(k)
        assert len(lo) == c["n_pass"] and sum(p["knn"] for p in lo) == c["n_knn_pass"], k
        d = float(np.abs(c["state"] - so).max())
        worst = max(worst, d)
        assert d < 1e-9, (k, d)
        # the per-scan engine (host-driven loop) forgets its cache per job too: the same registration
        assert (a["n_pass"], a["n_knn_pass"]) == (c["n_pass"], c["n_knn_pass"]) and np.abs(a["state"] - c["state"]).max() < 1e-9, k
        assert np.linalg.norm(c["state"][:3] - sc["pos"]) < 0.1
    print("batch vs oracle: worst |dstate|", worst)
    # a second call re-uses the slots (their neighbour caches hold another scan's neighbours by now): bit-identical answers
    rc3, res3 = b.process(jobs)
    assert rc3 == 0
    for a, c in zip(res2, res3):
        assert (a["rc"], a["n_ds"], a["n_pass"], a["n_knn_pass"]) == (c["rc"], c["n_ds"], c["n_pass"], c["n_knn_pass"])
        assert a["rc"] != 3 or np.array_equal(a["state"], c["state"])
    # ... and so does another geometry (jobs land in other slots, behind other scans)
    b2 = lio.Batch(the_map, n_slots=3, n_groups=3)
    rc4, res4 = b2.process(jobs[::-1])
    assert rc4 == 0
    for a, c in zip(res2, res4[::-1]):
        assert (a["rc"], a["n_ds"], a["n_pass"], a["n_knn_pass"]) == (c["rc"], c["n_ds"], c["n_pass"], c["n_knn_pass"])
        assert a["rc"] != 3 or np.array_equal(a["state"], c["state"])


def test_batch_sequence_jobs_keep_the_neighbour_cache(oracle_mod):
    """LIO_JOB_KEEP_CACHE: the jobs of a slot are consecutive scans of one sequence -- Nearest_Points persists across fastlio_main calls in the
    reference (stale where a search finds nothing in range, resized to the new scan, laserMapping.cpp:1274): the oracle replays the slot's
    history and must agree"""
    _dev()
    from lsd_amd import lio

    scene = scenes.config_scene()
    mp = scene.sample_surface(400_000, seed=2, sigma=0.01)  # sparse enough for searches that find fewer than five points in range
    the_map = lio.Map(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=1_000_000)
    the_map.add(mp)
    keep = []
    jobs, meta = _jobs(scene, range(3400, 3412), keep)
    for j in jobs:
        j["flags"] = lio.JOB_KEEP_CACHE
    jobs[5]["n"] = 3  # not registered: the cache of the slot's previous scan survives whole
    P0 = lio.init_cov()
    b = lio.Batch(the_map, n_slots=2, n_groups=2)
    rc, res = b.process(jobs)
    assert rc == 0
    o = oracle_mod.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=8)
    o.map_add(mp)
    o.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)
    n_slots_total = 4  # 2 groups x 2 slots handed out round robin: job k runs in the slot that ran jobs k - 4, k - 8 before it

    def oracle_run(k):
        sc = meta[k]
        o.set_state(sc["guess"])
        o.set_cov(P0)
        o.set_ds(oracle_mod.voxel_downsample(sc["raw"][: jobs[k]["n"]], 0.5))
        return o.update(), o.get_state()

    for k, c in enumerate(res):
        if k == 5:
            assert c["rc"] == 2
            continue
        assert c["rc"] == 3
        o.reset_cache()
        for h in range(k % n_slots_total, k, n_slots_total):
            if res[h]["rc"] == 3:
                oracle_run(h)
        lo, so = oracle_run(k)
        assert len(lo) == c["n_pass"] and sum(p["knn"] for p in lo) == c["n_knn_pass"], k
        assert np.abs(c["state"] - so).max() < 1e-9, k


@pytest.mark.parametrize("name", ["open_ground", "box_12x4"])
def test_batch_degenerate_scenes(oracle_mod, name):
    """the degeneracy sums and the projection inside the device-resident loop (step kernel)"""
    _dev()
    from lsd_amd import lio

    case = scenes.degenerate_case(name)
    the_map = lio.Map(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=500_000)
    the_map.add(case["map"])
    dptr = scenes.to_device(case["raw"])
    P0 = lio.init_cov()
    b = lio.Batch(the_map, n_slots=2, n_groups=1)
    rc, res = b.process([dict(dptr=dptr, n=len(case["raw"]), t=1.0, state=case["guess"], cov=P0)] * 3)
    assert rc == 0
    o = oracle_mod.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=8)
    o.map_add(case["map"])
    o.set_state(case["guess"])
    o.set_cov(P0)
    o.set_flags(ekf_inited=True, first_scan=False)
    o.set_ds(oracle_mod.voxel_downsample(case["raw"], 0.5))
    lo = o.update()
    so = o.get_state()
    assert o.is_degenerate
    for r in res:
        assert r["rc"] == 3 and r["n_pass"] == len(lo)
        assert np.abs(r["state"] - so).max() < 1e-8


def test_batch_sparse_scans_hand_over_to_the_host_filter(oracle_mod):
    """1 <= N_eff < 23: the device loop stops before that pass, the slot's engine continues with the dense gain branch"""
    _dev()
    from lsd_amd import lio

    P0 = lio.init_cov()
    for n_az, n_beams in ((6, 8), (8, 6), (10, 4)):
        case = scenes.degenerate_case("open_ground", n_az=n_az, n_beams=n_beams)
        the_map = lio.Map(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=500_000)
        the_map.add(case["map"])
        dptr = scenes.to_device(case["raw"])
        b = lio.Batch(the_map, n_slots=2, n_groups=1)
        rc, res = b.process([dict(dptr=dptr, n=len(case["raw"]), t=1.0, state=case["guess"], cov=P0)])
        assert rc == 0 and res[0]["rc"] == 3
        o = oracle_mod.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=8)
        o.map_add(case["map"])
        o.set_state(case["guess"])
        o.set_cov(P0)
        o.set_flags(ekf_inited=True, first_scan=False)
        o.set_ds(oracle_mod.voxel_downsample(case["raw"], 0.5))
        lo = o.update()
        assert res[0]["n_pass"] == len(lo) and res[0]["n_knn_pass"] == sum(p["knn"] for p in lo), (n_az, n_beams, res[0], len(lo))
        assert np.abs(res[0]["state"] - o.get_state()).max() < 1e-8
        del b


def test_host_raw_jobs_give_the_bits_of_resident_clouds():
    """LIO_JOB_HOST_RAW (round 5, the upload-included leg of bench.py): the same jobs with their clouds in pinned host memory -- the library copies a
    round's clouds to HBM on the round's stream -- return the states of the jobs whose clouds were resident, bit for bit, through the batched engine
    and through lio_engines_process_batch; a garbage flags word is rejected (never read as IDLE)"""
    _dev()
    from lsd_amd import capi, lio

    assert capi.lib().lio_abi_version() >= 5
    scene = scenes.config_scene()
    mp = scene.sample_surface(400_000, seed=2, sigma=0.01)
    the_map = lio.Map(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=1_000_000)
    the_map.add(mp)
    jobs, meta = _jobs(scene, range(3100, 3111), [])
    jobs[4]["n"] = 0
    pinned = [lio.PinnedCloud(sc["raw"]) for sc in meta]
    hjobs = [dict(j, dptr=pc.ptr, flags=lio.JOB_HOST_RAW) for j, pc in zip(jobs, pinned)]
    b = lio.Batch(the_map, n_slots=4, n_groups=2)
    rc0, res0 = b.process(jobs)
    rc1, res1 = b.process(hjobs)
    rc2, res2 = b.process(hjobs)  # (the raw rings exist now: the steady state)
    assert rc0 == rc1 == rc2 == 0
    for k, (a, c, d) in enumerate(zip(res0, res1, res2)):
        assert (a["rc"], a["n_ds"], a["n_pass"], a["n_knn_pass"]) == (c["rc"], c["n_ds"], c["n_pass"], c["n_knn_pass"]) == (d["rc"], d["n_ds"], d["n_pass"], d["n_knn_pass"]), k
        if a["rc"] == 3:
            assert np.array_equal(a["state"], c["state"]) and np.array_equal(a["state"], d["state"]), k
    assert sum(r["rc"] == 3 for r in res0) == 10 and res0[4]["rc"] == 2
    eng = lio.Engine(max_raw=1 << 18, max_ds=100000, shared_map=the_map)
    eng.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)
    rc3, res3 = lio.process_batch([eng], hjobs)
    assert rc3 == 0
    for k, (a, c) in enumerate(zip(res0, res3)):
        assert a["rc"] == c["rc"] and (a["rc"] != 3 or np.array_equal(a["state"], c["state"])), k
    # a flags word with an unknown bit AND the idle bit: rejected, not skipped
    bad = [dict(jobs[0], flags=0x42)]
    rc4, res4 = b.process(bad)
    assert res4[0]["rc"] == capi.LIO_E_INVALID


def test_large_clouds_take_the_strided_kernels_round_their_grids(oracle_mod):
    """Round 5: linearize_batch / vg_centroid_* / classify_seq stride over a slot's blocks with grids sized for the typical cloud (256 x 64 points,
    96 x 256 voxels) instead of the slot's capacity.  Clouds LARGER than one sweep of those grids must give the oracle's results too: a scan whose
    0.5 m grid holds more than 16 384 points through the batched engine (poses, pass structure), and a 0.1 m grid of more than 50 000 voxels through the
    single and the batched downsample (bit-exact centroids, in order)."""
    _dev()
    from lsd_amd import lio, synth

    scene = scenes.config_scene()
    mp = scene.sample_surface(1_000_000, seed=2, sigma=0.01)
    the_map = lio.Map(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=1_000_000)
    the_map.add(mp)
    P0 = lio.init_cov()
    # dense scans (every ray hits: the metric config's field of view, 2 040 azimuth steps): n_ds ~ 18 000 > 256 x 64
    rng = np.random.default_rng(77)
    jobs, meta = [], []
    for k in range(3):
        pos = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), 1.8])
        q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
        raw, _ = synth.make_scan(scene, pos, q, seed=4100 + k, n_az=2040, fov_deg=(-24.8, 2.0), max_range=150.0)  # 130 560 points, ~18 000 voxels of 0.5 m
        gp, gq = synth.perturb_pose(pos, q, seed=9100 + k, max_t=0.2, max_deg=1.0)
        meta.append(dict(raw=raw, guess=synth.state_from_pose(gp, gq)))
        jobs.append(dict(dptr=scenes.to_device(raw), n=len(raw), t=1.0 + 0.1 * k, state=meta[-1]["guess"], cov=P0))
    b = lio.Batch(the_map, n_slots=2, n_groups=2, max_raw=1 << 17, max_ds=100000)
    rc, res = b.process(jobs)
    assert rc == 0
    o = oracle_mod.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=8)
    o.map_add(mp)
    o.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)
    big = 0
    for k, r in enumerate(res):
        assert r["rc"] == 3, (k, r["rc"])
        big += r["n_ds"] > 16384
        o.reset_cache()
        o.set_state(meta[k]["guess"])
        o.set_cov(P0)
        ds = oracle_mod.voxel_downsample(meta[k]["raw"], 0.5)
        assert len(ds) == r["n_ds"], (k, len(ds), r["n_ds"])
        o.set_ds(ds)
        lo = o.update()
        assert (len(lo), sum(p["knn"] for p in lo)) == (r["n_pass"], r["n_knn_pass"]), k
        assert np.abs(r["state"] - o.get_state()).max() < 1e-9, k
    assert big >= 2, [r["n_ds"] for r in res]  # (the point of the test: more than one sweep of linearize_batch's grid)
    # the voxel grid at a 0.1 m leaf: more voxels than one sweep of the centroid kernels' grids, single and batched
    raw = meta[0]["raw"]
    ref = oracle_mod.voxel_downsample(raw, 0.1)
    assert len(ref) > 50_000
    s1 = lio.Scan(max_raw=1 << 17, max_ds=120000)
    s1.upload(raw)
    assert s1.voxel_downsample(0.1) == len(ref) and np.array_equal(s1.get_ds().view(np.uint32), ref.view(np.uint32))
    ss = [lio.Scan(max_raw=1 << 17, max_ds=120000) for _ in range(2)]
    for s_, m_ in zip(ss, meta[:2]):
        s_.upload(m_["raw"])
    ns = lio.Scan.voxel_downsample_batch(ss, 0.1)
    for s_, m_, n_ in zip(ss, meta[:2], ns):
        r_ = oracle_mod.voxel_downsample(m_["raw"], 0.1)
        assert n_ == len(r_) and np.array_equal(s_.get_ds().view(np.uint32), r_.view(np.uint32))


def test_xcd_aware_tile_mapping_with_a_ragged_slot_count(oracle_mod):
    """Round 5: the batched chain maps workgroup L to tile (L / 8) % T of slot ((L / 8) / T) * 8 + L % 8 for the slots of whole groups of eight and keeps
    blockIdx for the rest.  Twelve and nineteen scans per launch (one / two whole groups + a remainder), of different sizes incl. an empty one: every
    scan's downsampled cloud equals the oracle's bit for bit, in order."""
    _dev()
    from lsd_amd import lio, synth

    scene = scenes.config_scene()
    rng = np.random.default_rng(5)
    clouds = []
    for k in range(19):
        pos = np.array([rng.uniform(-20, 20), rng.uniform(-20, 20), 1.8])
        q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
        raw, _ = synth.make_scan(scene, pos, q, seed=8200 + k, n_az=120 + 40 * (k % 5), fov_deg=(-24.8, 2.0), max_range=120.0)
        clouds.append(raw if k != 7 else raw[:0])
    refs = [oracle_mod.voxel_downsample(c, 0.5) if len(c) else np.zeros((0, 4), np.float32) for c in clouds]
    scans = [lio.Scan(max_raw=1 << 15, max_ds=20000) for _ in range(19)]
    for s_, c in zip(scans, clouds):
        if len(c):
            s_.upload(c)
    for count in (12, 19):
        live = [(s_, r_) for s_, c, r_ in zip(scans[:count], clouds[:count], refs[:count]) if len(c)]
        ns = lio.Scan.voxel_downsample_batch([s_ for s_, _ in live], 0.5)
        assert len(live) in (11, 18)
        for (s_, r_), n_ in zip(live, ns):
            assert n_ == len(r_) and np.array_equal(s_.get_ds().view(np.uint32), r_.view(np.uint32))
