"""Pins the CPU oracle against the reference's OWN code (oracle/_ref/libref_harness.so = esti_plane and iVox
compiled from /root/reference by `make -C oracle ref`).  Bit-exact for esti_plane; neighbour sets and voxel
bookkeeping for iVox.  Skipped only where the harness has not been built (it travels to the GPU box prebuilt)."""
import numpy as np
import pytest

import ref as refmod

pytestmark = pytest.mark.skipif(not refmod.available(), reason="oracle/_ref/libref_harness.so not built (needs /root/reference)")


def _plane_sets(rng, n):
    """5-point neighbourhoods: noisy planes of random orientation/offset, plus generic and degenerate sets"""
    out = []
    for i in range(n):
        kind = i % 5
        nrm = rng.normal(size=3)
        nrm /= np.linalg.norm(nrm)
        c = rng.uniform(-80, 80, 3)
        u = np.cross(nrm, rng.normal(size=3))
        u /= np.linalg.norm(u)
        v = np.cross(nrm, u)
        ab = rng.uniform(-0.6, 0.6, (5, 2))
        sig = [0.0, 0.005, 0.02, 0.08, 0.3][kind]
        pts = c + ab[:, :1] * u + ab[:, 1:] * v + rng.normal(0, sig, (5, 1)) * nrm
        if i % 37 == 0:
            pts[3] = pts[2]  # duplicate point
        if i % 41 == 0:
            pts = c + ab[:, :1] * u  # collinear: rank deficient
        out.append(np.concatenate([pts, np.zeros((5, 1))], 1).astype(np.float32))
    return out


def test_esti_plane_bit_exact(oracle_mod):
    rng = np.random.default_rng(0)
    n_ok = 0
    for pts in _plane_sets(rng, 4000):
        ok_r, p_r = refmod.esti_plane(pts)
        ok_o, p_o = oracle_mod.esti_plane(pts)
        both_finite = np.all(np.isfinite(p_r)) and np.all(np.isfinite(p_o))
        assert ok_r == ok_o
        if both_finite:
            assert np.array_equal(p_r.view(np.uint32), p_o.view(np.uint32)), (pts, p_r, p_o)
        n_ok += ok_r
    assert 500 < n_ok < 3900  # both gate outcomes are exercised


def test_ivox_knn_sets(oracle_mod):
    rng = np.random.default_rng(1)
    pts = np.concatenate([rng.uniform(-6, 6, (30000, 2)), rng.normal(0, 0.05, (30000, 1)), rng.uniform(0, 255, (30000, 1))], 1).astype(np.float32)
    wall = np.concatenate([rng.uniform(-6, 6, (8000, 1)), np.full((8000, 1), 3.0) + rng.normal(0, 0.03, (8000, 1)), rng.uniform(0, 3, (8000, 1)),
                           rng.uniform(0, 255, (8000, 1))], 1).astype(np.float32)
    pts = np.concatenate([pts, wall])
    q = pts[rng.choice(len(pts), 3000, replace=False)].copy()
    q[:, :3] += rng.normal(0, 0.15, (len(q), 3)).astype(np.float32)
    q[:50, 2] += 40.0  # nothing nearby
    q[50:100, 2] += 1.2  # only a few candidates in range of the stencil
    for stencil in (19, 75, 7, 27, 1):
        r = refmod.IVox(stencil=stencil)
        o = oracle_mod.IVox(stencil=stencil)
        r.add(pts[:20000], 0.0)
        r.add(pts[20000:], 3.0)
        o.add(pts[:20000], 0.0)
        o.add(pts[20000:], 3.0)
        assert r.num_voxels == o.num_voxels
        nn_r, cnt_r = r.knn(q)
        nn_o, cnt_o, _ = o.knn(q)
        assert np.array_equal(cnt_r, cnt_o), stencil
        nn_r = refmod.canonical(nn_r, cnt_r, q)
        assert np.array_equal(nn_r[..., :3].view(np.uint32), nn_o[..., :3].view(np.uint32)), stencil
        # the reference's one ordering promise: element 0 is the nearest (ivox3d.h:163)
        raw, _ = r.knn(q)
        has = cnt_r > 0
        assert np.array_equal(raw[has, 0, :3].view(np.uint32), nn_o[has, 0, :3].view(np.uint32))


def test_ivox_lru_eviction_bookkeeping(oracle_mod):
    """capacity / max_distance eviction (ivox3d.h:251-254): same voxel count after every batch"""
    rng = np.random.default_rng(2)
    r = refmod.IVox(stencil=19, capacity=300, max_distance=10.0)
    o = oracle_mod.IVox(stencil=19, capacity=300, max_distance=10.0)
    travel = 0.0
    for step in range(40):
        c = np.array([step * 1.5, 0.0, 0.0])
        pts = np.concatenate([c + rng.uniform(-4, 4, (400, 3)) * [1, 1, 0.1], rng.uniform(0, 255, (400, 1))], 1).astype(np.float32)
        travel += 1.5
        r.add(pts, travel)
        o.add(pts, travel)
        assert r.num_voxels == o.num_voxels, step
    q = np.concatenate([np.array([[50.0, 0, 0]]) + rng.uniform(-3, 3, (200, 3)) * [1, 1, 0.05], np.zeros((200, 1))], 1).astype(np.float32)
    nn_r, cnt_r = r.knn(q)
    nn_o, cnt_o, _ = o.knn(q)
    assert np.array_equal(cnt_r, cnt_o)
    assert np.array_equal(refmod.canonical(nn_r, cnt_r, q)[..., :3].view(np.uint32), nn_o[..., :3].view(np.uint32))


def test_ivox_lru_revisits_drop_and_recreate_like_the_reference(oracle_mod):
    """The point-by-point order of AddPoints (ivox3d.h:231-256) where it shows: a course that keeps coming back, so that voxels at the back of the list are
    touched by the very batch that is evicting around them -- the compiled ivox3d.h drops such a voxel with all it held and re-creates it from the
    batch's points; the oracle's list (the statements the HIP map is held to, tests/test_lru_gpu.py) must do the same: voxel counts after every batch,
    and the neighbours of queries all over the visited ground -- which are the voxels' CONTENT -- every third batch."""
    rng = np.random.default_rng(4)
    for cap, maxd, course in ((900, 1.0, [(-1) ** k * (2.0 + 0.6 * k) for k in range(30)]), (1200, 3.0, [-12.0, 0.0, 12.0] * 8), (500, 0.0, list(rng.uniform(-10, 10, 24)))):
        r = refmod.IVox(stencil=19, capacity=cap, max_distance=maxd)
        o = oracle_mod.IVox(stencil=19, capacity=cap, max_distance=maxd)
        travel = 0.0
        for b, cx in enumerate(course):
            travel += 1.5
            pts = np.concatenate([np.array([cx, 0.0, 0.0]) + rng.uniform(-4, 4, (500, 3)) * [1, 1, 0.1], rng.uniform(0, 255, (500, 1))], 1).astype(np.float32)
            r.add(pts, travel)
            o.add(pts, travel)
            assert r.num_voxels == o.num_voxels, (cap, b)
            if b % 3 == 2:
                q = np.concatenate([np.array([[cx, 0.0, 0.0]]) + rng.uniform(-8, 8, (300, 3)) * [1, 1, 0.05], np.zeros((300, 1))], 1).astype(np.float32)
                nn_r, cnt_r = r.knn(q)
                nn_o, cnt_o, _ = o.knn(q)
                assert np.array_equal(cnt_r, cnt_o), (cap, b)
                assert np.array_equal(refmod.canonical(nn_r, cnt_r, q).view(np.uint32), nn_o.view(np.uint32)), (cap, b)


def test_calc_dist(oracle_mod):
    rng = np.random.default_rng(3)
    for _ in range(200):
        a = rng.uniform(-100, 100, 4).astype(np.float32)
        b = rng.uniform(-100, 100, 4).astype(np.float32)
        d = (a[:3] - b[:3]).astype(np.float32)
        want = np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        assert np.float32(refmod.calc_dist(a, b)) == want


def test_undistort_point_and_Exp_bit_exact(oracle_mod):
    """so3_math.h Exp(ang_vel, dt) (the reference's template) and the per-point compensation expression of
    ImuProcess::UndistortPcl evaluated by real Eigen (scalar build): the oracle reproduces both bit for bit"""
    ref = refmod
    from lsd_amd import synth

    rng = np.random.default_rng(0)
    for _ in range(5000):
        q = lambda: synth.quat_from_rotvec(rng.normal(size=3) * 0.5)
        Rm = synth.quat_to_R(q())
        vel, pos, acc, gyr = rng.normal(size=3) * 5, rng.normal(size=3) * 50, rng.normal(size=3) * 3, rng.normal(size=3) * 0.5
        dt = rng.uniform(0, 0.02)
        p = (rng.normal(size=3) * 30).astype(np.float32)
        epos, erot, ril, til = pos + rng.normal(size=3) * 0.5, q(), q(), rng.normal(size=3) * 0.3
        a = oracle_mod.undistort_point(Rm, vel, pos, acc, gyr, dt, p, epos, erot, ril, til)
        b = ref.undistort_point(Rm, vel, pos, acc, gyr, dt, p, epos, erot, ril, til)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(oracle_mod.so3_Exp(gyr, dt).view(np.uint64), ref.so3_Exp(gyr, dt).view(np.uint64))
    # below the 1e-7 rad/s gate Exp is the identity
    assert np.array_equal(oracle_mod.so3_Exp([1e-9, 0, 0], 0.1), np.eye(3)) and np.array_equal(ref.so3_Exp([1e-9, 0, 0], 0.1), np.eye(3))


def _tie_lattice(rng, n, step=0.0625, half=1.5):
    """points on a dyadic lattice (every coordinate a multiple of 2^-4): squared distances between them are exact in f32, so equal distances
    are everywhere -- with a distinct intensity per point so that WHICH of two equally distant points was kept can be read off the result"""
    ijk = rng.integers(-int(half / step), int(half / step) + 1, (n, 3))
    return np.concatenate([ijk * step, np.arange(n)[:, None] + 1.0], 1).astype(np.float32)


def test_tie_at_the_fifth_nearest_boundary_is_resolved_as_the_reference_resolves_it(oracle_mod):
    """Candidates at EXACTLY the same f32 squared distance compete for the fifth place of a query's neighbour list.  The reference keeps
    whichever std::nth_element leaves in front (ivox3d_node.hpp:107-127, ivox3d.h:156-164) given ITS candidate sequence: stencil order, push_back
    order inside a voxel, every voxel cut to five first.  The oracle makes the same calls on the same sequence (same libstdc++): its list must
    equal the compiled ivox3d.h's element for element, intensity included, in the reference's own order -- and its canonical-order list must be
    that same set.  (Until round 6 the oracle and the kernels took the five smallest in (d2, x, y, z): another valid 5-NN set, but another set.)"""
    rng = np.random.default_rng(5)
    n_boundary = 0
    for trial in range(6):
        pts = _tie_lattice(rng, 6000 + 3000 * trial)
        q = _tie_lattice(rng, 1500)
        q[:, :3] += np.float32(0.03125) * (trial % 2)  # half of the trials: queries between the lattice planes
        for stencil in (19, 7, 27, 75, 1):
            r, o, c = refmod.IVox(stencil=stencil), oracle_mod.IVox(stencil=stencil), oracle_mod.IVox(stencil=stencil)
            c.set_tie_mode(0)
            cuts = sorted(rng.choice(len(pts), 3, replace=False).tolist())  # the map arrives in four batches: push_back order spans batches
            for lo, hi in zip([0] + cuts, cuts + [len(pts)]):
                for m in (r, o, c):
                    m.add(pts[lo:hi], float(lo))
            nn_r, cnt_r = r.knn(q)
            nn_a, cnt_a = o.knn_as_reference(q)
            assert np.array_equal(cnt_r, cnt_a)
            assert np.array_equal(nn_r.view(np.uint32), nn_a.view(np.uint32)), (trial, stencil)
            nn_o, cnt_o, _ = o.knn(q)
            assert np.array_equal(cnt_o, cnt_r)
            # the same SET (intensities identify the points), in the canonical order
            assert np.array_equal(np.sort(nn_o[..., 3], axis=1), np.sort(nn_r[..., 3], axis=1)), (trial, stencil)
            assert np.array_equal(refmod.canonical(nn_r, cnt_r, q)[..., :3].view(np.uint32), nn_o[..., :3].view(np.uint32))
            nn_c, _, _ = c.knn(q)
            n_boundary += int(np.any(np.sort(nn_c[..., 3], axis=1) != np.sort(nn_r[..., 3], axis=1), axis=1).sum())
    assert n_boundary > 100, "the lattice did not produce boundary ties: the test would not distinguish the two definitions"
