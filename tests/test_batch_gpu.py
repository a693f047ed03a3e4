"""The batched, device-resident engine (lio_batch_*: B scans per launch, the batched 16-lanes-per-query kNN (and the selectable one-lane-per-query kernel of knn_q.hip), the filter loop of
esekfom.hpp:1619-1931 on the device) against the oracle and against the per-scan engine it replaces in throughput mode."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenes  # noqa: E402
import test_gpu_parity as tgp  # noqa: E402

pytestmark = pytest.mark.gpu


def _dev():
    from lsd_amd import capi

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device")


# ---- the four-lanes-per-query kernel answers lio_map_knn: every kNN test of test_gpu_parity.py again ---------------------------
def test_knn_q_insert_and_knn_exact(oracle_mod, small_world, monkeypatch):
    monkeypatch.setenv("LIO_KNN_Q", "1")
    tgp.test_map_insert_and_knn_exact(oracle_mod, small_world)


def test_knn_q_ties_and_duplicates(oracle_mod, monkeypatch):
    monkeypatch.setenv("LIO_KNN_Q", "1")
    tgp.test_knn_exact_ties_and_duplicates(oracle_mod)


def test_knn_q_adversarial(oracle_mod, monkeypatch):
    monkeypatch.setenv("LIO_KNN_Q", "1")
    tgp.test_knn_pruned_sweep_adversarial(oracle_mod)


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
    same return codes, downsampled sizes, pass counts as the per-scan engine path; poses equal to the oracle's"""
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
    n_slots_total = 8  # 2 groups x 4 slots, handed out round robin: job k runs in the slot that ran jobs k - 8, k - 16 before it

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
        # The neighbour cache of a slot carries over from the scan it registered before (Nearest_Points persists across scans in the
        # reference, stale where a search finds nothing in range): give the oracle the same history -- the earlier jobs of that slot.
        # (The per-scan engine `eng` saw all 20 scans in a row: another history, so only sizes and return codes are compared with it.)
        sc = meta[k]
        o.reset_cache()
        for h in range(k % n_slots_total, k, n_slots_total):
            if res2[h]["rc"] == 3:
                oracle_run(h)
        lo, so = oracle_run(k)
        assert len(lo) == c["n_pass"] and sum(p["knn"] for p in lo) == c["n_knn_pass"], k
        d = float(np.abs(c["state"] - so).max())
        worst = max(worst, d)
        assert d < 1e-9, (k, d)
        assert np.linalg.norm(c["state"][:3] - sc["pos"]) < 0.1
    print("batch vs oracle: worst |dstate|", worst)
    # a second call re-uses the slots (their neighbour caches now hold another scan's neighbours: stale entries must not matter
    # because every first pass searches again) and must give the same answers
    rc3, res3 = b.process(jobs)
    assert rc3 == 0
    for a, c in zip(res2, res3):  # (other histories in the slots' caches: the registrations agree to what stale neighbours can move them)
        assert a["rc"] == c["rc"] and (a["rc"] != 3 or np.linalg.norm(a["state"][:3] - c["state"][:3]) < 5e-3)


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


def test_second_search_from_previous_neighbours_is_exact(oracle_mod):
    """From the second neighbour search of an update on, a query still in the voxel of the last full search is searched only in the voxels
    that can beat the neighbours found then (knn.hip).  Same scans through fresh Batch objects with the short cut on and off: states,
    pass / search counts BIT-identical, and the candidate statistic (the stencil's residents, counted or carried over)
    equal -- on scans whose prior is 0.3 m off (queries change voxel between searches: both paths run) and on sparse-map scans."""
    _dev()
    from lsd_amd import capi, lio

    from lsd_amd import synth

    scene = scenes.config_scene()
    # priors 0.3 m off (the iterate leaves the neighbourhood of the first search: the short cut is not even tried), a sparse map, and priors
    # 3 cm off -- a tracking front end's -- where the second search of most queries takes the short cut
    for n_map, seeds, near in ((1_000_000, range(3100, 3112), False), (60_000, range(3200, 3206), False), (1_000_000, range(3300, 3312), True)):
        mp = scene.sample_surface(n_map, seed=4, sigma=0.01)
        the_map = lio.Map(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=1_000_000)
        the_map.add(mp)
        keep = []
        jobs, meta = _jobs(scene, seeds, keep)
        if near:
            for k, (j, sc) in enumerate(zip(jobs, meta)):
                gp, gq = synth.perturb_pose(sc["pos"], sc["q"], seed=9000 + k, max_t=0.03, max_deg=0.2)
                j["state"] = synth.state_from_pose(gp, gq)
        out = {}
        for on in (1, 0):
            capi.lib().lio_debug_knn_reuse(on)
            try:
                b = lio.Batch(the_map, n_slots=4, n_groups=2)
                c0 = the_map.knn_candidates
                rc, res = b.process(jobs)
                assert rc == 0
                out[on] = (res, the_map.knn_candidates - c0)
                del b
            finally:
                capi.lib().lio_debug_knn_reuse(0)
        (ra, ca), (rb, cb) = out[1], out[0]
        assert ca == cb, (ca, cb)
        n_two = 0
        for a, c in zip(ra, rb):
            assert (a["rc"], a["n_ds"], a["n_pass"], a["n_knn_pass"]) == (c["rc"], c["n_ds"], c["n_pass"], c["n_knn_pass"])
            assert np.array_equal(a["state"], c["state"])
            n_two += a["n_knn_pass"] >= 2
        assert n_two >= len(ra) // 2  # the short cut had something to do
    # and through the single-scan engine (device loop)
    eng = lio.Engine(max_raw=1 << 18, max_ds=100000, shared_map=the_map)
    eng.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)
    states = {}
    for on in (1, 0):
        capi.lib().lio_debug_knn_reuse(on)
        try:
            e2 = lio.Engine(max_raw=1 << 18, max_ds=100000, shared_map=the_map)
            e2.set_device_loop(True)
            e2.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)
            st = []
            for k, j in enumerate(jobs):
                e2.set_state(j["state"])
                e2.set_cov(lio.init_cov())
                assert e2.process_scan_device(j["dptr"], j["n"], 1.0 + 0.1 * k) == 3
                st.append((e2.get_state(), e2.get_cov()))
            states[on] = st
        finally:
            capi.lib().lio_debug_knn_reuse(0)
    for (sa, pa), (sb, pb) in zip(states[1], states[0]):
        assert np.array_equal(sa, sb) and np.array_equal(pa, pb)
