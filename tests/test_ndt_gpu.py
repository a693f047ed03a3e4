"""HIP localization matcher (NDT-P2D) vs the CPU oracle, through the C ABI.  The voxel statistics go through cosf / sinf /
atan2f (closed-form eigen-solver), which differ in the last bits between libm and the GPU's OCML, so this path is compared
with tolerances -- tight ones for the sums (the per-pair terms are otherwise the same IEEE operations), the north-star
1e-4 m / 1e-5 rad for the aligned pose."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _world():
    from lsd_amd import synth

    scene = synth.Scene(half=60.0, n_boxes=20, seed=3)
    mp = scene.sample_surface(200_000, seed=4, sigma=0.02)
    true_pos, true_q = np.array([0.5, 1.0, 1.8]), synth.quat_from_rotvec([0, 0, -0.2])
    raw, _ = synth.make_scan(scene, true_pos, true_q, seed=5, n_az=600)
    gp, gq = synth.perturb_pose(true_pos, true_q, seed=6, max_t=0.5, max_deg=3.0)

    def T_of(pos, q):
        T = np.eye(4)
        T[:3, :3] = synth.quat_to_R(q)
        T[:3, 3] = pos
        return T

    return mp, raw, T_of(true_pos, true_q), T_of(gp, gq)


def _rot_angle(A, B):
    R = A[:3, :3] @ B[:3, :3].T
    return float(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))


@pytest.mark.parametrize("method", [7, 1, 27])
def test_ndt_matches_oracle(method):
    import ndt as ondt
    import oracle
    from lsd_amd import lio

    mp, raw, T_true, T_guess = _world()
    ds = oracle.voxel_downsample(raw, 0.5)
    o = ondt.Ndt(1.0, method)
    o.set_target(mp)
    o.set_source(ds)
    g = lio.Ndt(resolution=1.0, search_method=method, max_points=400_000, max_voxels=200_000, max_source_points=100_000)
    g.set_target(mp)
    assert g.num_voxels == o.num_voxels
    # voxel statistics: same counts, means bit-exact (plain f32 sums in input order), inverse covariances to f32-libm tolerance
    rng = np.random.default_rng(0)
    for p in mp[rng.choice(len(mp), 200, replace=False)]:
        n_o, mean_o, _, cinv_o = o.voxel_at(p[:3])
        n_g, mean_g, cinv_g = g.voxel_at(p[:3])
        assert n_o == n_g
        assert np.array_equal(mean_o.view(np.uint32), mean_g.view(np.uint32))
        assert np.allclose(cinv_o, cinv_g, rtol=2e-3, atol=0.5)
    s = lio.Scan(max_raw=1 << 17, max_ds=100000)
    s.set_ds(ds)
    lo = o.linearize(T_guess)
    lg = g.linearize(s, T_guess)
    assert lo["n_corr"] == lg["n_corr"] and lo["n_corr"] > 5000
    assert np.allclose(lg["H"], lo["H"], rtol=1e-3, atol=1e-3 * np.abs(lo["H"]).max())
    # only the lower triangle of H is accumulated on the device (it is all Eigen::LDLT / ldlt_solve6 read): H comes back mirrored, exactly symmetric
    assert np.array_equal(lg["H"], lg["H"].T)
    assert np.allclose(lg["b"], lo["b"], rtol=1e-3, atol=1e-3 * np.abs(lo["b"]).max())
    assert abs(lg["err"] - lo["err"]) < 1e-3 * lo["err"]
    # error-only evaluation on the cached pairs at another transform (LM trial)
    T2 = T_guess.copy()
    T2[:3, 3] += [0.05, -0.02, 0.01]
    assert abs(g.linearize(s, T2, update_corr=False, with_derivatives=False)["err"] - o.compute_error(T2)) < 1e-3 * lo["err"]
    # full alignment
    To, conv_o, it_o = o.align(T_guess)
    Tg, conv_g, it_g = g.align(s, T_guess)
    assert conv_o and conv_g and it_o == it_g
    assert np.linalg.norm(Tg[:3, 3] - To[:3, 3]) < 1e-4 and _rot_angle(Tg, To) < 1e-5
    if method == 7:  # the reference's configuration; DIRECT1 / DIRECT27 at resolution 1.0 land 0.1-0.2 m off (oracle and HIP alike)
        assert np.linalg.norm(Tg[:3, 3] - T_true[:3, 3]) < 0.1 < np.linalg.norm(T_guess[:3, 3] - T_true[:3, 3])


def test_ndt_target_replacement_and_downsampled_source():
    """setInputTarget twice (local-map refresh, localization.cpp:351-352) and the real source path: upload + VoxelGrid"""
    import ndt as ondt
    import oracle
    from lsd_amd import lio

    mp, raw, T_true, T_guess = _world()
    g = lio.Ndt(resolution=1.0, search_method=7, max_points=400_000, max_voxels=200_000)
    g.set_target(mp[:50_000])
    first = g.num_voxels
    g.set_target(mp)
    o = ondt.Ndt(1.0, 7)
    o.set_target(mp)
    assert g.num_voxels == o.num_voxels and first < g.num_voxels
    s = lio.Scan(max_raw=1 << 17, max_ds=100000)
    s.upload(raw)
    s.voxel_downsample(0.5)
    o.set_source(oracle.voxel_downsample(raw, 0.5))
    Tg, conv, _ = g.align(s, T_guess)
    To, _, _ = o.align(T_guess)
    assert conv and np.linalg.norm(Tg[:3, 3] - To[:3, 3]) < 1e-4 and _rot_angle(Tg, To) < 1e-5


def test_pose_estimator_loop_tracks_a_drive(scene):
    """hdl_localization's per-scan loop on the device matcher: UKF predict (IMU) -> NDT align from the predicted pose ->
    gate -> correct, over a short drive against a prebuilt target; the filter stays on the trajectory"""
    from lsd_amd import capi, lio, synth

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible")
    tr = synth.Trajectory(t_static=0.2, speed=4.0)
    target = scene.sample_surface(3_000_000, seed=21, sigma=0.01)
    target = target[np.linalg.norm(target[:, :2] - tr.pos(0.0)[:2], axis=1) < 70.0]
    ndt = lio.Ndt(resolution=1.0, search_method=7, max_points=len(target) + 1, max_voxels=400_000, max_source_points=1 << 17)
    ndt.set_target(target)
    scan = lio.Scan(max_raw=1 << 18, max_ds=1 << 17)
    R0, p0 = tr.R(0.0), tr.pos(0.0)
    q0 = synth.quat_from_rotvec([0, 0, tr.heading])  # (x, y, z, w)
    est = lio.PoseEstimator(p0, [q0[3], q0[0], q0[1], q0[2]], stamp_us=0, cool_time=0.0)
    # a settled filter: the reference starts the quaternion block at variance 0.1 (sigma points far from unit length), which
    # makes its first seconds wander; the loop mechanics are what is tested here
    est.set(cov=np.eye(23, dtype=np.float32) * 1e-3)
    imu = synth.imu_stream(tr, 0.0, 2.2, rate=100.0)
    ii, worst, n_ok = 0, 0.0, 0
    for k in range(1, 20):
        tb = k * 0.1
        accs, gyrs = [], []
        while ii < len(imu) and imu[ii][0] <= tb:
            accs.append(imu[ii][2])
            gyrs.append(imu[ii][1])
            ii += 1
        est.predict(int(round(tb * 1e6)), np.mean(accs, 0), np.mean(gyrs, 0))  # one step per frame with the mean IMU (hdl_localization_nodelet.cpp:218-220)
        raw, _ = synth.make_scan(scene, tr.pos(tb), tr.quat(tb), seed=k, n_az=900, fov_deg=(-24.8, 2.0))
        scan.upload(raw)
        scan.voxel_downsample(0.2)
        if k % 3 == 0:
            # a GNSS observation of this frame (6-D, then 2-D): match_gps = guess -> lio_ndt_align -> observe, the same three calls by hand
            Tg = np.eye(4)
            Tg[:3, :3], Tg[:3, 3] = tr.R(tb), tr.pos(tb) + [0.2, -0.1, 0.0]
            gps = (Tg, 1.0, 6 if k % 6 == 0 else 2)
            G = est.guess(gps)
            Ta, conv, it2 = ndt.align(scan, G.astype(np.float64))
            ok2, obs2, cov2 = est.observe(G, Ta.astype(np.float32), conv, gps)
            ok, obs, cov, it = est.match_gps(ndt, scan, gps)
            assert ok == ok2 and it == it2 and np.array_equal(obs, obs2) and np.array_equal(cov, cov2)
            assert np.linalg.norm(obs[:3] - tr.pos(tb)) < 0.3
        else:
            ok, obs, it = est.match(ndt, scan)
        n_ok += int(ok)
        est.correct(int(round(tb * 1e6)), obs)
        T = est.matrix()
        dp = np.linalg.norm(T[:3, 3] - tr.pos(tb))
        dr = np.arccos(min(1.0, (np.trace(T[:3, :3].astype(np.float64).T @ tr.R(tb)) - 1) / 2))
        worst = max(worst, dp)
        assert dp < 0.3 and dr < 0.05, (k, dp, dr)
    assert n_ok >= 15, n_ok
    print("pose estimator loop: worst position error %.3f m, %d / 19 matches accepted" % (worst, n_ok))


def test_local_map_assembly_on_device(oracle_mod):
    """Localization::runUpdateLocalMap with the key frames resident in HBM: selection (radius, nearest first, thinning, point cap), VoxelGrid and
    target build against the REFERENCE'S OWN loop body (localization.cpp:305-312,325-372 cut out and compiled in oracle/ref_localmap.cpp; its
    results on tests/localmap_cases.py are recorded in tests/golden/localmap.npz): same return code for every pose of the drive -- replaced /
    nothing to do / replaced / out of map / nearest key frame too far -- and a downsampled local map that is bit-identical to the one the
    reference's code hands its localizer (live against the harness when it travelled with the snapshot); the matcher built from it aligns a scan"""
    import localmap_cases as lc
    from lsd_amd import capi, lio, synth

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "localmap.npz"))
    scene = lc.scene()
    frames, poses = lc.key_frames(scene)
    lm = lio.LocalMap(max_total_points=6_000_000, max_local_points=200_000, max_keyframe_points=80_000)
    for k, (w, p) in enumerate(zip(frames, poses)):
        assert lm.add_keyframe(w, p) == k
    ndt = lio.Ndt(resolution=1.0, search_method=7, max_points=400_000, max_voxels=200_000, max_source_points=1 << 17)
    R = None
    import ref_localmap

    if ref_localmap.available():
        R = ref_localmap.RefLocalMap(resolution=lc.LEAF, key_frame_distance=lc.KEY_FRAME_DISTANCE)
        for w, p in zip(frames, poses):
            R.add_keyframe(w, p)
    for k, (pose, what) in enumerate(lc.LM_POSES):
        rc, nk, npts = lm.update(ndt, np.array(pose), key_frame_distance=lc.KEY_FRAME_DISTANCE, leaf=lc.LEAF)
        assert rc == g["codes"][k], (what, rc)
        if rc == 1:
            got = lm.download()
            assert npts == len(got) and np.array_equal(lc.digest(got), g["digests"][k]), what
        if rc in (2, 3):
            assert ndt.num_voxels == 0, what  # the target is dropped (mLocalMap = nullptr handed to the localizer)
        if R is not None:
            assert R.update(pose) == rc, what
            if rc == 1:
                assert np.array_equal(R.local_map().view(np.uint32), lm.download().view(np.uint32)), what
        if k == 0:  # the target built from the first local map localises a fresh scan
            p0 = np.array(pose)
            tq = synth.quat_from_rotvec([0, 0, 0.1])
            raw, _ = synth.make_scan(scene, p0, tq, seed=999, n_az=900, fov_deg=(-24.8, 2.0))
            src = lio.Scan(max_raw=1 << 18, max_ds=1 << 17)
            src.upload(raw)
            src.voxel_downsample(0.2)
            T0 = np.eye(4)
            T0[:3, :3] = synth.quat_to_R(synth.quat_from_rotvec([0, 0, 0.12]))
            T0[:3, 3] = p0 + [0.15, -0.1, 0.02]
            T, conv, it = ndt.align(src, T0)
            assert conv and np.linalg.norm(T[:3, 3] - p0) < 0.05


def test_fitness_score_exact_nearest_neighbours(scene):
    """getFitnessScore(max_range): exact nearest neighbours by rings of target voxels vs brute force in the same f32 arithmetic"""
    from lsd_amd import capi, lio, synth

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible")
    rng = np.random.default_rng(12)
    target = scene.sample_surface(400_000, seed=31, sigma=0.01)
    target = target[np.linalg.norm(target[:, :2], axis=1) < 35.0][:40_000]
    ndt = lio.Ndt(resolution=1.0, search_method=7, max_points=len(target) + 1, max_voxels=100_000, max_source_points=1 << 16)
    ndt.set_target(target)
    src_pts = np.concatenate([target[rng.choice(len(target), 2500, replace=False)] + rng.normal(0, 0.05, (2500, 4)).astype(np.float32),
                              np.concatenate([rng.uniform(-60, 60, (500, 2)), rng.uniform(5, 30, (500, 1)), np.zeros((500, 1))], 1).astype(np.float32)])
    scan = lio.Scan(max_raw=1 << 16, max_ds=1 << 16)
    scan.set_ds(src_pts)
    T = np.eye(4)
    T[:3, :3] = synth.quat_to_R(synth.quat_from_rotvec([0.01, -0.02, 0.05]))
    T[:3, 3] = [0.3, -0.2, 0.1]
    Tf = T.astype(np.float32)
    tp = np.stack([((Tf[r, 0] * src_pts[:, 0] + Tf[r, 1] * src_pts[:, 1]) + Tf[r, 2] * src_pts[:, 2]) + Tf[r, 3] for r in range(3)], 1)
    best = np.full(len(tp), np.inf, np.float32)
    for a in range(0, len(tp), 200):
        d = tp[a:a + 200, None, :] - target[None, :, :3]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        best[a:a + 200] = d2.min(1)
    for max_range in (25.0, 1.0, 1e-3):
        inl = best <= np.float32(max_range)
        score, n_in = ndt.fitness_score(scan, T, max_range)
        assert n_in == inl.sum(), (max_range, n_in, inl.sum())
        assert abs(score - best[inl].astype(np.float64).mean()) <= 1e-12 * max(1.0, score)
    far = src_pts.copy()
    far[:, 2] += 500.0
    scan.set_ds(far)
    score, n_in = ndt.fitness_score(scan, T, 25.0)
    assert n_in == 0 and score > 1e300  # PCL: std::numeric_limits<double>::max()


def test_overlap_score_of_the_map_merge_tools():
    """calc_fitness_score(cloud1, cloud2, relpose, max_range): both clouds through the range / floor filter (the source AFTER the transform), exact
    nearest neighbours, mean squared distance of the inliers and their share -- against the REFERENCE'S OWN filter + calc_fitness_score
    (overlap_merge.hpp:213-263 cut out and compiled in oracle/ref_localmap.cpp, recorded in tests/golden/localmap.npz; live when the harness
    travelled with the snapshot)"""
    import localmap_cases as lc
    from lsd_amd import capi, lio

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "localmap.npz"))
    cloud1, cloud2, T = lc.overlap_case()
    target = lio.overlap_filter(cloud1)
    assert 0 < len(target) < len(cloud1)
    ndt = lio.Ndt(resolution=1.0, search_method=7, max_points=len(target) + 1, max_voxels=200_000, max_source_points=1 << 16)
    ndt.set_target(target)
    scan = lio.Scan(max_raw=1 << 16, max_ds=1 << 16)
    scan.set_ds(cloud2)
    import ref_localmap

    n_inl = 0
    for max_range, (want_score, want_ratio) in zip(g["ranges"], g["fitness"]):
        score, ratio = ndt.overlap_score(scan, T, max_range)
        if want_ratio == 0.0:
            assert score > 1e300 and ratio == 0.0
            continue
        n_inl += 1
        # the reference adds the f32 squared distances one by one in point order, the device in a fixed tree: the mean agrees to rounding
        assert ratio == want_ratio and abs(score - want_score) <= 1e-12 * max(1.0, want_score), (max_range, score, want_score)
        if ref_localmap.available():
            ls, lr = ref_localmap.overlap_fitness(cloud1, cloud2, T, max_range)
            assert (ls, lr) == (want_score, want_ratio)
    assert n_inl >= 3
    # nothing survives the filter / nothing within range: (DBL_MAX, 0)
    up = cloud2.copy()
    up[:, 2] -= 100.0
    scan.set_ds(up)
    score, ratio = ndt.overlap_score(scan, T, 1.0)
    assert score > 1e300 and ratio == 0.0 and g["fitness_none"][0] > 1e300


def test_batched_alignments_equal_the_single_ones(scene):
    """lio_ndt_align_batch: B alignments per launch against one target, the Levenberg-Marquardt loop of LsqRegistration on the device (lsq.h compiled
    for it) -- the candidate alignments of the map-merge / loop-closure tools (overlap_merge.hpp:158-179) -- against lio_ndt_align job by job: same
    convergence flags and iteration counts, poses equal to the rounding of the device's libm; more jobs than slots, sources of different sizes, a
    hopeless guess that runs into the iteration limit"""
    from lsd_amd import capi, lio, synth

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible")
    target = scene.sample_surface(2_000_000, seed=41, sigma=0.01)
    target = np.ascontiguousarray(target[np.linalg.norm(target[:, :2], axis=1) < 90.0])
    ndt = lio.Ndt(resolution=1.0, search_method=7, max_points=len(target) + 1, max_voxels=400_000, max_source_points=1 << 16)
    ndt.set_target(target)
    rng = np.random.default_rng(5)
    scans, guesses, truth = [], [], []
    for k in range(70):  # more than the 64 slots of a launch
        pos = np.array([rng.uniform(-6, 6), rng.uniform(-6, 6), 1.8])
        q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
        raw, _ = synth.make_scan(scene, pos, q, seed=700 + k, n_az=300 + 40 * (k % 5), fov_deg=(-24.8, 2.0))
        sc = lio.Scan(max_raw=1 << 17, max_ds=1 << 16)
        sc.upload(raw)
        sc.voxel_downsample(0.3)
        gp, gq = synth.perturb_pose(pos, q, seed=800 + k, max_t=0.4, max_deg=2.5)
        if k == 13:
            gp = gp + [25.0, -20.0, 0.0]  # far off: the alignment runs until it gives up
        G, T = np.eye(4), np.eye(4)
        G[:3, :3], G[:3, 3] = synth.quat_to_R(gq), gp
        T[:3, :3], T[:3, 3] = synth.quat_to_R(q), pos
        scans.append(sc)
        guesses.append(G)
        truth.append(T)
    single = [ndt.align(sc, G) for sc, G in zip(scans, guesses)]
    batch = ndt.align_batch(scans, guesses)
    worst = 0.0
    for k, ((Ts, cs, its), (Tb, cb, itb, evals, rc)) in enumerate(zip(single, batch)):
        assert rc == 0 and cb == cs and itb == its, (k, cs, cb, its, itb)
        assert evals >= itb + 2 or not cb  # one linearisation, then one (speculative) evaluation per LM trial: at least one per iteration
        worst = max(worst, float(np.abs(Tb - Ts).max()))
        assert np.abs(Tb - Ts).max() < 1e-9, (k, np.abs(Tb - Ts).max())
        if k != 13:
            assert cb and np.linalg.norm(Tb[:3, 3] - truth[k][:3, 3]) < 0.05, k
    print("batched vs single NDT alignments: worst |dT| %.2e" % worst)
    # run-to-run identical
    again = ndt.align_batch(scans, guesses)
    for a, b in zip(batch, again):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    # one job list over SEVERAL targets (a new key frame against each of its candidate frames, overlap_merge.hpp:158-179: setInputTarget per candidate)
    halves = [np.ascontiguousarray(target[target[:, 0] < 20.0]), np.ascontiguousarray(target[target[:, 0] > -20.0])]
    others = []
    for h in halves:
        t2 = lio.Ndt(resolution=1.0, search_method=7, max_points=len(h) + 1, max_voxels=400_000, max_source_points=1 << 16)
        t2.set_target(h)
        others.append(t2)
    tg = [None if k % 3 == 0 else others[k % 3 - 1] for k in range(12)]
    mixed = ndt.align_batch(scans[:12], guesses[:12], targets=tg)
    for k, (Tb, cb, itb, evals, rc) in enumerate(mixed):
        Ts, cs, its = (ndt if tg[k] is None else tg[k]).align(scans[k], guesses[k])
        assert rc == 0 and (cb, itb) == (cs, its) and np.abs(Tb - Ts).max() < 1e-9, k


def test_batched_voxel_downsample_equals_the_single_scans(scene):
    """lio_scan_voxel_downsample_batch: one set of launches for n scans -- every scan's output bit-identical (points AND order) to its own
    lio_scan_voxel_downsample, for clouds of different sizes, a leaf that needs another number of radix passes, and again on re-use"""
    from lsd_amd import lio, synth

    rng = np.random.default_rng(3)
    clouds = []
    for k, n_az in enumerate((450, 1875, 120, 900, 33)):
        raw, _ = synth.make_scan(scene, np.array([2.0 * k, -1.0 * k, 1.8]), synth.quat_from_rotvec([0, 0, 0.3 * k]), seed=700 + k, n_az=n_az, max_range=150.0)
        clouds.append(raw[:, :4].astype(np.float32))
    clouds.append(clouds[0][:7].copy())  # fewer points than one tile
    single = lio.Scan(max_raw=1 << 18, max_ds=200000)
    batch = [lio.Scan(max_raw=1 << 18, max_ds=200000) for _ in clouds]
    for leaf in (0.2, 0.5, 2.0, 0.2):
        want = []
        for c in clouds:
            single.upload(c)
            single.voxel_downsample(leaf)
            want.append(single.get_ds())
        for b, c in zip(batch, clouds):
            b.upload(c)
        counts = lio.Scan.voxel_downsample_batch(batch, leaf)
        for b, w, cnt in zip(batch, want, counts):
            got = b.get_ds()
            assert cnt == len(w) == len(got) and np.array_equal(got.view(np.uint32), w.view(np.uint32)), (leaf, cnt, len(w))
    with pytest.raises(Exception):
        lio.Scan.voxel_downsample_batch([batch[0], batch[0]], 0.5)
    # an empty scan among the others; a scan whose output does not fit its max_ds fails loudly and leaves the others intact
    empty = lio.Scan(max_raw=1 << 18, max_ds=200000)
    small = lio.Scan(max_raw=1 << 18, max_ds=64)
    batch[0].upload(clouds[0])
    small.upload(clouds[1])
    single.upload(clouds[0])
    single.voxel_downsample(0.5)
    want0 = single.get_ds()
    with pytest.raises(Exception, match="max_ds"):
        lio.Scan.voxel_downsample_batch([batch[0], empty, small], 0.5)
    assert np.array_equal(batch[0].get_ds().view(np.uint32), want0.view(np.uint32))
    counts = lio.Scan.voxel_downsample_batch([batch[0], empty], 0.5)
    assert counts == [len(want0), 0]


def test_speculative_trial_evaluation_is_bit_identical(scene, monkeypatch):
    """lio_ndt_align fetches a trial pose's cost (on the pairs cached at the linearisation point) and the linearisation an accepted step continues
    from in ONE launch (ndt_cost_spec_kernel + lsq_align_spec); LIO_NDT_SPEC=0 keeps LsqRegistration's two evaluations per iteration.  Same
    statements, same accumulation order: poses, convergence flags and iteration counts are the same BITS -- accepted steps, rejected trials (a
    hopeless guess), one-offset and 27-offset neighbourhoods"""
    from lsd_amd import capi, lio, synth

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible")
    target = scene.sample_surface(1_500_000, seed=43, sigma=0.01)
    target = np.ascontiguousarray(target[np.linalg.norm(target[:, :2], axis=1) < 80.0])
    rng = np.random.default_rng(11)
    cases = []
    for k in range(10):
        pos = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), 1.8])
        q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
        raw, _ = synth.make_scan(scene, pos, q, seed=900 + k, n_az=400, fov_deg=(-24.8, 2.0))
        gp, gq = synth.perturb_pose(pos, q, seed=950 + k, max_t=0.5, max_deg=3.0)
        if k == 4:
            gp = gp + [20.0, 15.0, 0.0]  # far off: rejected trials, lambda growing, gives up
        G = np.eye(4)
        G[:3, :3], G[:3, 3] = synth.quat_to_R(gq), gp
        cases.append((raw, G))
    for method in (7, 1, 27):
        res, lin = {}, {}
        for flag in ("1", "0"):
            monkeypatch.setenv("LIO_NDT_SPEC", flag)  # read by lio_ndt_create
            ndt = lio.Ndt(resolution=1.0, search_method=method, max_points=len(target) + 1, max_voxels=400_000, max_source_points=1 << 16)
            ndt.set_target(target)
            sc = lio.Scan(max_raw=1 << 17, max_ds=1 << 16)
            out, lins = [], []
            for raw, G in cases:
                sc.upload(raw)
                sc.voxel_downsample(0.3)
                lins.append(ndt.linearize(sc, G))
                out.append(ndt.align(sc, G))
            res[flag], lin[flag] = out, lins
            ndt.close()
        for k, ((Ta, ca, ia), (Tb, cb, ib)) in enumerate(zip(res["1"], res["0"])):
            assert (ca, ia) == (cb, ib) and np.array_equal(Ta, Tb), (method, k, ca, cb, ia, ib, np.abs(Ta - Tb).max())
        for k, (la, lb) in enumerate(zip(lin["1"], lin["0"])):  # (the pair count travels with the sums since round 4: the same count either way)
            assert la["n_corr"] == lb["n_corr"] > 0 and np.array_equal(la["H"], lb["H"]) and la["err"] == lb["err"], (method, k)
        assert sum(int(c) for _, c, _ in res["1"]) >= 8
