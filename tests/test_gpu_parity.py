"""HIP path vs the CPU oracle, through the C ABI, on a real MI355X.  Bit-exact for every per-point quantity
(voxel centroids, neighbour sets, plane parameters, gates); f64 reductions to 1e-10 relative; poses to the
north-star tolerance 1e-4 m / 1e-5 rad (observed ~1e-10)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev():
    from lsd_amd import capi

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box")


def test_voxel_downsample_bitexact(oracle_mod, small_world):
    _dev()
    from lsd_amd import lio

    raw = small_world["raw"]
    ref = oracle_mod.voxel_downsample(raw, 0.5)
    s = lio.Scan(max_raw=1 << 18, max_ds=100000)
    s.upload(raw)
    n = s.voxel_downsample(0.5)
    got = s.get_ds()
    assert n == len(ref) == len(got)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))  # bit-exact, same order


def test_voxel_downsample_edge_cases(oracle_mod):
    _dev()
    from lsd_amd import lio

    s = lio.Scan(max_raw=1 << 16, max_ds=1 << 16)
    rng = np.random.default_rng(3)
    cases = {
        "single": np.array([[1.0, 2.0, 3.0, 4.0]], np.float32),
        "negative": np.concatenate([rng.uniform(-30, -10, (5000, 3)), rng.uniform(0, 255, (5000, 1))], 1).astype(np.float32),
        "nan_inf": np.concatenate([rng.uniform(-5, 5, (3000, 3)), rng.uniform(0, 255, (3000, 1))], 1).astype(np.float32),
        "dupes": np.repeat(np.array([[0.1, 0.2, 0.3, 1.0], [7.3, -2.2, 0.9, 2.0]], np.float32), 700, 0),
        "ragged_tile": np.concatenate([rng.uniform(-20, 20, (1025, 3)), rng.uniform(0, 255, (1025, 1))], 1).astype(np.float32),
        "overflow_guard": np.array([[0, 0, 0, 1], [1e6, 1e6, 1e6, 2], [5, 5, 5, 3]], np.float32),
    }
    cases["nan_inf"][::7, 0] = np.nan
    cases["nan_inf"][3::11, 2] = np.inf
    for name, pts in cases.items():
        leaf = 0.01 if name == "overflow_guard" else 0.5
        ref = oracle_mod.voxel_downsample(pts, leaf)
        s.upload(pts)
        n = s.voxel_downsample(leaf)
        got = s.get_ds()
        assert n == len(ref), name
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), name
    s.upload(np.zeros((0, 4), np.float32))
    assert s.voxel_downsample(0.5) == 0


def test_voxel_downsample_pass_prediction(oracle_mod):
    """the host launches as many radix passes as the previous scan needed and the device checks: clouds whose bounding boxes need
    1, 2, 3 and 4 passes in every order on one scan object -- each result bit-exact whether the prediction held or the chain ran twice"""
    _dev()
    from lsd_amd import lio

    rng = np.random.default_rng(5)
    clouds = {}
    for name, half in (("8 bits", (1.5, 1.5, 1.5)), ("15 bits", (8.0, 8.0, 8.0)), ("22 bits", (60.0, 60.0, 12.0)), ("26 bits", (200.0, 200.0, 25.0))):
        p = rng.uniform(-1, 1, (30_000, 3)) * np.array(half)
        clouds[name] = np.c_[p, rng.uniform(0, 255, len(p))].astype(np.float32)
    ref = {k: oracle_mod.voxel_downsample(v, 0.5) for k, v in clouds.items()}
    sc = lio.Scan(max_raw=1 << 16, max_ds=1 << 16)
    order = ["22 bits", "8 bits", "26 bits", "15 bits", "15 bits", "26 bits", "8 bits", "22 bits", "26 bits", "22 bits"]
    for name in order:
        sc.upload(clouds[name])
        n = sc.voxel_downsample(0.5)
        assert n == len(ref[name]), name
        assert np.array_equal(sc.get_ds().view(np.uint32), ref[name].view(np.uint32)), name
    sc.close()


def test_map_insert_and_knn_exact(oracle_mod, small_world):
    _dev()
    from lsd_amd import lio

    pts = small_world["map"]
    iv = oracle_mod.IVox(res=0.5, stencil=19)
    iv.add(pts)
    m = lio.Map(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=500_000)
    m.add(pts[:200_000])
    m.add(pts[200_000:])  # second batch grows existing voxels
    assert m.stats() == (iv.num_points, iv.num_voxels)
    d = m.dump()
    assert np.array_equal(np.sort(d.view(np.uint32).view([("a", np.uint32, 4)]).ravel(), order="a"),
                          np.sort(pts.view(np.uint32).view([("a", np.uint32, 4)]).ravel(), order="a"))
    rng = np.random.default_rng(5)
    q = pts[rng.choice(len(pts), 20000, replace=False)].copy()
    q[:, :3] += rng.normal(0, 0.2, (len(q), 3)).astype(np.float32)
    q[:100, 2] += 50.0  # far from everything: zero candidates
    for st in (19, 75, 7, 27, 1):
        iv.set_stencil(st)
        m.set_stencil(st)
        ref_pts, ref_cnt, _ = iv.knn(q)
        got_pts, got_cnt = m.knn(q)
        assert np.array_equal(ref_cnt, got_cnt), st
        assert np.array_equal(ref_pts[..., :3].view(np.uint32), got_pts[..., :3].view(np.uint32)), st


def test_knn_exact_ties_and_duplicates(oracle_mod):
    """dyadic lattice map + mid-cell queries: many exact d2 ties between different points (and exact duplicates);
    the (d2, x, y, z) canonical order must still be reproduced bit for bit (tie queue -> knn_exact_kernel)"""
    _dev()
    from lsd_amd import lio

    ax = np.arange(-16, 16) * 0.25
    X, Y, Z = np.meshgrid(ax, ax, np.arange(-4, 4) * 0.25, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel(), np.zeros(X.size)], 1).astype(np.float32)
    pts = np.concatenate([pts, pts[::5]])  # exact duplicates
    rng = np.random.default_rng(9)
    qi = rng.integers(-14, 14, (4000, 3))
    q = np.concatenate([qi * 0.25 + 0.125, np.zeros((4000, 1))], 1).astype(np.float32)
    q[:, 2] = np.clip(q[:, 2], -0.875, 0.625)
    q[::3, 0] += 0.0625  # some queries off-centre in x only: ties in (y, z) remain
    iv = oracle_mod.IVox(res=0.5, stencil=19)
    iv.add(pts)
    m = lio.Map(resolution=0.5, stencil=19, max_points=200_000, max_voxels=100_000)
    m.add(pts)
    for st in (19, 75):
        iv.set_stencil(st)
        m.set_stencil(st)
        ref_pts, ref_cnt, _ = iv.knn(q)
        got_pts, got_cnt = m.knn(q)
        assert np.array_equal(ref_cnt, got_cnt)
        assert np.array_equal(ref_pts[..., :3].view(np.uint32), got_pts[..., :3].view(np.uint32))


def _tie_lattice(rng, n, step=0.0625, half=1.5, first_id=1):
    """points on a dyadic lattice (exact f32 distances: ties everywhere), each with its own intensity so that WHICH of two equally distant
    points was kept can be read off the result"""
    ijk = rng.integers(-int(half / step), int(half / step) + 1, (n, 3))
    return np.concatenate([ijk * step, np.arange(n)[:, None] + float(first_id)], 1).astype(np.float32)


def test_knn_boundary_ties_are_the_references(oracle_mod):
    """Candidates EXACTLY as far as the fifth nearest: the kernels keep the ones the reference keeps -- std::nth_element over the stencil's
    candidates in nearby_grids_ / push_back order, every voxel cut to five first (ivox3d_node.hpp:107-127, ivox3d.h:156-164; csrc/refsel.h on
    the device, the push_back rank of every map point kept beside it).  Lattice map, inserted in several batches (voxels grow and move: the ranks
    move with them), intensities identify the points: all four components of all five neighbours must equal the oracle's, which is pinned
    element for element to the compiled ivox3d.h (tests/test_oracle_vs_ref.py) -- and, where that library is present, the reference's own."""
    _dev()
    from lsd_amd import lio

    import ref as refmod

    rng = np.random.default_rng(21)
    decided = 0
    for trial in range(3):
        pts = _tie_lattice(rng, 20000 + 10000 * trial)
        q = _tie_lattice(rng, 4000)
        q[:, :3] += np.float32(0.03125) * (trial % 2)
        cuts = sorted(rng.choice(len(pts), 4, replace=False).tolist())
        for st in (19, 7, 27, 75, 1):
            iv = oracle_mod.IVox(res=0.5, stencil=st)
            canon = oracle_mod.IVox(res=0.5, stencil=st)
            canon.set_tie_mode(0)
            m = lio.Map(resolution=0.5, stencil=st, max_points=200_000, max_voxels=100_000)
            r = refmod.IVox(stencil=st) if refmod.available() else None
            for lo, hi in zip([0] + cuts, cuts + [len(pts)]):
                iv.add(pts[lo:hi], float(lo))
                canon.add(pts[lo:hi], float(lo))
                m.add(pts[lo:hi], float(lo))
                if r is not None:
                    r.add(pts[lo:hi], float(lo))
            ref_pts, ref_cnt, _ = iv.knn(q)
            got_pts, got_cnt = m.knn(q)
            assert np.array_equal(ref_cnt, got_cnt), (trial, st)
            # positions in the canonical order; WHICH points: by their intensities (points at one and the same position have no order in (d2, x, y, z))
            assert np.array_equal(ref_pts[..., :3].view(np.uint32), got_pts[..., :3].view(np.uint32)), (trial, st)
            bad = np.flatnonzero(np.any(np.sort(ref_pts[..., 3], axis=1) != np.sort(got_pts[..., 3], axis=1), axis=1))
            assert len(bad) == 0, (trial, st, len(bad), bad[:5], ref_pts[bad[:2]], got_pts[bad[:2]])
            if r is not None:
                nn_r, cnt_r = r.knn(q)
                assert np.array_equal(cnt_r, got_cnt)
                assert np.array_equal(np.sort(nn_r[..., 3], axis=1), np.sort(got_pts[..., 3], axis=1)), (trial, st)
            n_b, n_un = m.tie_stats()
            can_pts, _, _ = canon.knn(q)
            differ = int(np.any(np.sort(can_pts[..., 3], axis=1) != np.sort(ref_pts[..., 3], axis=1), axis=1).sum())
            assert n_un == 0 and n_b >= differ
            decided += differ
            # the earlier definition is still there: the five smallest in (d2, x, y, z) -- duplicates of one position apart (their order is
            # not defined by that key), so positions only
            m.set_tie_mode(0)
            got0, cnt0 = m.knn(q)
            assert np.array_equal(cnt0, got_cnt)
            assert np.array_equal(can_pts[..., :3].view(np.uint32), got0[..., :3].view(np.uint32)), (trial, st)
            m.close()
    assert decided > 200, "the lattice did not produce boundary ties whose resolution differs between the two definitions"


def test_knn_boundary_ties_in_a_voxel_too_large_for_the_staging_area():
    """a stencil voxel with more in-range points than the redo's staging area holds: counted (lio_map_tie_stats), the canonical five kept"""
    _dev()
    from lsd_amd import lio

    rng = np.random.default_rng(3)
    big = np.concatenate([rng.integers(-3, 4, (3000, 3)) * 0.0625, np.arange(3000)[:, None] + 1.0], 1).astype(np.float32)  # one voxel, 3000 lattice points
    m = lio.Map(resolution=0.5, stencil=19, max_points=50_000, max_voxels=10_000)
    m.add(big)
    q = np.array([[0.03125, 0.03125, 0.03125, 0.0]], np.float32)
    got, cnt = m.knn(q)
    assert cnt[0] == 5
    n_b, n_un = m.tie_stats()
    assert n_b == 1 and n_un == 1
    d2 = np.sum((got[0, :, :3] - q[0, :3]) ** 2, axis=1)
    assert np.all(d2 == d2[0])  # eight corners of the cell around the query are equally near: any five of them
    m.close()


def test_map_clear_and_refill(oracle_mod, small_world):
    """lio_map_clear (fastlio_init's fresh IVox, laserMapping.cpp:1064): an emptied map takes a new content like a new map -- with and without the LRU list --
    and a content put back voxel by voxel keeps every voxel's push_back order (what bench.py's teacher-forced legs rely on: tie mode 2 on a lattice map
    must still return the reference's lists, order included)"""
    _dev()
    from lsd_amd import lio

    pts = small_world["map"][:120_000]
    rng = np.random.default_rng(12)
    q = pts[rng.choice(len(pts), 5000, replace=False)].copy()
    q[:, :3] += rng.normal(0, 0.2, (len(q), 3)).astype(np.float32)
    for lru in (False, True):
        m = lio.Map(resolution=0.5, stencil=19, max_points=600_000, max_voxels=300_000)
        if lru:
            m.set_lru(200_000, 100.0)
        m.add(pts[:50_000], 1.0)
        m.add(pts[50_000:], 2.0)
        want = m.knn(q)
        stats = m.stats()
        m.clear()
        assert m.stats() == (0, 0)
        got0, cnt0 = m.knn(q[:100])
        assert not cnt0.any()
        m.add(pts[:50_000], 1.0)
        m.add(pts[50_000:], 2.0)
        assert m.stats() == stats
        got = m.knn(q)
        assert np.array_equal(want[1], got[1]) and np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32)), lru
        m.close()
    # push_back order survives a dump-and-refill: lattice map (ties everywhere), lists in the reference's own order (tie mode 2) before and after
    lat = _tie_lattice(rng, 15000)
    ql = _tie_lattice(rng, 1500)
    iv = oracle_mod.IVox(res=0.5, stencil=19)
    iv.set_tie_mode(2)
    m = lio.Map(resolution=0.5, stencil=19, max_points=200_000, max_voxels=100_000)
    m.set_tie_mode(2)
    for lo, hi in ((0, 4000), (4000, 9000), (9000, 15000)):
        iv.add(lat[lo:hi], 0.0)
        m.add(lat[lo:hi], 0.0)
    ref_pts, ref_cnt, _ = iv.knn(ql)
    got_pts, got_cnt = m.knn(ql)
    assert np.array_equal(ref_cnt, got_cnt) and np.array_equal(ref_pts.view(np.uint32), got_pts.view(np.uint32))
    dump = iv.dump()  # voxel by voxel, push_back order inside a voxel
    m.clear()
    m.add(dump, 0.0)
    got2, cnt2 = m.knn(ql)
    assert np.array_equal(ref_cnt, cnt2) and np.array_equal(ref_pts.view(np.uint32), got2.view(np.uint32))
    m.close()


def test_knn_pruned_sweep_adversarial(oracle_mod):
    """the sweep visits the stencil voxels nearest-first and skips those that cannot beat five known candidates: maps and queries built
    to make that decision as hard as possible -- queries on voxel faces / edges / corners (+- one f32 ulp), nearest neighbours living in
    the far (edge) cells while the near cells hold only far points, fewer than five candidates, other resolutions, far-from-origin
    coordinates (the bound's slack scales with |q|)"""
    _dev()
    from lsd_amd import lio

    rng = np.random.default_rng(77)
    few = none = 0
    for res, origin in ((0.5, 0.0), (0.5, 4096.0), (1.0, -2500.0), (0.2, 0.0)):
        n = 60_000
        # sparse-to-dense mix: a dense slab, a sparse cloud, and clumps hugging voxel boundaries
        slab = np.c_[rng.uniform(-8, 8, (n // 2, 2)) * res * 2, rng.normal(0, 0.02, n // 2)]
        cloud = rng.uniform(-8, 8, (n // 4, 3)) * res * 2
        k = rng.integers(-16, 16, (n // 4, 3)).astype(np.float64)
        hug = (k + 0.5) * res + rng.choice([-1, 1], (n // 4, 3)) * rng.uniform(0, 0.02, (n // 4, 3)) * res
        pts = np.concatenate([slab, cloud, hug]) + origin
        pts = np.c_[pts, np.zeros(len(pts))].astype(np.float32)
        # queries: on faces, edges and corners of voxels (k + 0.5) * res, nudged by one ulp either way, plus random ones
        kq = rng.integers(-14, 14, (6000, 3)).astype(np.float64)
        onb = rng.random((6000, 3)) < 0.6
        q = (kq + np.where(onb, 0.5, rng.uniform(-0.5, 0.5, (6000, 3)))) * res + origin
        q = q.astype(np.float32)
        nudge = rng.integers(-1, 2, q.shape)
        q = np.where(nudge > 0, np.nextafter(q, np.float32(np.inf)), np.where(nudge < 0, np.nextafter(q, np.float32(-np.inf)), q)).astype(np.float32)
        q = np.c_[q, np.zeros(len(q), np.float32)]
        iv = oracle_mod.IVox(res=res, stencil=19)
        iv.add(pts)
        m = lio.Map(resolution=res, stencil=19, max_points=200_000, max_voxels=200_000)
        m.add(pts)
        for st in (7, 19, 27):
            iv.set_stencil(st)
            m.set_stencil(st)
            ref_pts, ref_cnt, _ = iv.knn(q)
            got_pts, got_cnt = m.knn(q)
            assert np.array_equal(ref_cnt, got_cnt), (res, origin, st)
            few += int(((ref_cnt > 0) & (ref_cnt < 5)).sum())
            none += int((ref_cnt == 0).sum())
            full = ref_cnt == 5
            assert np.array_equal(ref_pts[full][..., :3].view(np.uint32), got_pts[full][..., :3].view(np.uint32)), (res, origin, st)
            for c in range(1, 5):
                sel = ref_cnt == c
                assert np.array_equal(ref_pts[sel][:, :c, :3].view(np.uint32), got_pts[sel][:, :c, :3].view(np.uint32)), (res, origin, st, c)
        m.close()
    assert few > 20  # the fewer-than-five path was exercised too


def _make_pair(oracle_mod, small_world, stencil=19):
    from lsd_amd import lio, synth

    pts = small_world["map"]
    ds = oracle_mod.voxel_downsample(small_world["raw"], 0.5)
    state = synth.state_from_pose(small_world["guess_pos"], small_world["guess_q"])
    o = oracle_mod.Lio(res=0.5, stencil=stencil, capacity=1 << 40, threads=8)
    o.map_add(pts)
    o.set_state(state)
    o.set_cov(oracle_mod.init_cov())
    o.set_flags(ekf_inited=True, first_scan=False)
    o.set_ds(ds)
    e = lio.Engine(resolution=0.5, stencil=stencil, max_points=1_000_000, max_voxels=500_000, max_raw=1 << 18, max_ds=100000)
    e.map_add(pts)
    e.set_state(state)
    e.set_cov(lio.init_cov())
    e.set_flags(ekf_inited=True, first_scan=False)
    e.set_ds(ds)
    return o, e, state, ds


def test_linearize_matches_oracle(oracle_mod, small_world):
    _dev()
    from lsd_amd import lio

    o, e, state, ds = _make_pair(oracle_mod, small_world)
    ref = o.linearize(converge=True)
    got = lio.linearize(e.map, e.scan, state, redo_knn=True)
    mt = e.scan.get_match()
    assert got["n_ds"] == len(ds)
    assert np.array_equal(ref["nn_cnt"], mt["nn_cnt"])
    assert np.array_equal(ref["nn"][..., :3].view(np.uint32), mt["nn"][..., :3].view(np.uint32))
    assert np.array_equal(ref["selected"], mt["selected"])
    sel = ref["selected"].astype(bool)
    assert np.array_equal(ref["normvec"][sel].view(np.uint32), mt["normvec"][sel].view(np.uint32))  # plane + residual bit-exact
    assert got["n_eff"] == ref["n_eff"] and ref["n_eff"] > 1000
    assert np.allclose(got["JtJ"], ref["JtJ"], rtol=1e-10, atol=1e-9)
    assert np.allclose(got["Jtr"], ref["Jtr"], rtol=1e-10, atol=1e-9)
    assert abs(got["sum_abs_res"] - ref["sum_abs_res"]) < 1e-9 * max(1.0, ref["sum_abs_res"])
    # second pass without a neighbour search at a nudged state: the gate state carries over
    st2 = state.copy()
    st2[0] += 0.01
    o.set_state(st2)
    ref2 = o.linearize(converge=False)
    got2 = lio.linearize(e.map, e.scan, st2, redo_knn=False)
    mt2 = e.scan.get_match()
    assert np.array_equal(ref2["selected"], mt2["selected"])
    assert got2["n_eff"] == ref2["n_eff"]
    assert np.allclose(got2["JtJ"], ref2["JtJ"], rtol=1e-10, atol=1e-9)


def test_iterated_update_pose_parity(oracle_mod, small_world, loop_mode):
    _dev()
    from lsd_amd import synth

    o, e, state, ds = _make_pair(oracle_mod, small_world)
    lo = o.update()
    lg = e.update()
    assert len(lo) == len(lg)
    for a, b in zip(lo, lg):
        assert (a["knn"], a["n_eff"], a["valid"], a["degenerate"]) == (b["knn"], b["n_eff"], b["valid"], b["degenerate"])
        assert np.allclose(a["JtJ"], b["JtJ"], rtol=1e-9, atol=1e-8)
        assert np.allclose(a["dx"], b["dx"], rtol=0, atol=1e-9)
    so, sg = o.get_state(), e.get_state()
    assert np.linalg.norm(so[:3] - sg[:3]) < 1e-4
    assert synth.quat_angle(so[3:7], sg[3:7]) < 1e-5
    assert np.abs(so - sg).max() < 1e-8
    assert np.allclose(o.get_cov(), e.get_cov(), rtol=1e-7, atol=1e-12)
    # and the update actually registered the scan: closer to the truth than the guess was
    assert np.linalg.norm(sg[:3] - small_world["true_pos"]) < 0.03
    assert synth.quat_angle(sg[3:7], small_world["true_q"]) < 2e-3
    # map_incremental with the final state inserts the same set
    na = o.map_incremental()
    nb = e.map_incremental(ekf_inited=True)
    assert na == nb
    assert e.map.stats() == (o.map_num_points, o.map_num_voxels)


def test_process_scan_sequence(oracle_mod, scene, loop_mode):
    _dev()
    from lsd_amd import lio, synth

    o = oracle_mod.Lio(res=0.5, stencil=75, capacity=1 << 40, threads=8)
    e = lio.Engine(resolution=0.5, stencil=75, max_points=2_000_000, max_voxels=1_000_000, max_raw=1 << 18, max_ds=100000)
    s0 = synth.state_from_pose([0.0, 0.0, 1.8], [0, 0, 0, 1.0])
    for h in (o, e):
        h.set_state(s0)
        h.set_cov(oracle_mod.init_cov())
    pos = np.array([0.0, 0.0, 1.8])
    rcs = []
    for k in range(14):
        pos = pos + np.array([0.25, 0.05, 0.0])
        q = synth.quat_from_rotvec([0, 0, 0.01 * k])
        raw, _ = synth.make_scan(scene, pos, q, seed=100 + k, n_az=450)
        t = 0.1 * k
        ra = o.process_scan(raw, t)
        rb = e.process_scan(raw, t)
        rcs.append((ra, rb))
        assert ra == rb, (k, ra, rb)
        so, sg = o.get_state(), e.get_state()
        assert np.linalg.norm(so[:3] - sg[:3]) < 1e-4, k
        assert synth.quat_angle(so[3:7], sg[3:7]) < 1e-5, k
        assert e.map.stats() == (o.map_num_points, o.map_num_voxels), k
        # widen the prior like the IMU propagation would, so the filter keeps following the motion
        for h in (o, e):
            P = h.get_cov()
            P[:6, :6] += np.eye(6) * 1e-2
            h.set_cov(P)
    assert rcs[0] == (0, 0) and rcs[1] == (1, 1) and rcs[-1] == (3, 3)
    assert abs(o.travel - e.travel) < 1e-6


@pytest.mark.gpu
def test_map_overflow_is_reported_although_the_insert_is_not_waited_for(scene):
    """lio_engine_process_scan returns when the state is final; map_incremental runs on the map's stream behind it.  A map that runs out of
    room must still fail loudly: by the call that overflows it or by the next one that touches the map, never silently."""
    _dev()
    from lsd_amd import capi, lio, synth

    e = lio.Engine(resolution=0.5, stencil=19, max_points=12_000, max_voxels=6_000, max_raw=1 << 18, max_ds=100000)
    e.set_state(synth.state_from_pose([0.0, 0.0, 1.8], [0, 0, 0, 1.0]))
    pos = np.array([0.0, 0.0, 1.8])
    raised_at = None
    for k in range(40):
        pos = pos + np.array([1.5, 0.3, 0.0])   # new ground every scan: the map keeps growing
        raw, _ = synth.make_scan(scene, pos, synth.quat_from_rotvec([0, 0, 0.02 * k]), seed=300 + k, n_az=450)
        try:
            e.process_scan(raw, 0.1 * k)
            e.map.stats()
        except capi.LioError as ex:
            raised_at = k
            assert "capacity" in str(ex) or "exceed" in str(ex), str(ex)
            break
    assert raised_at is not None and raised_at >= 1
