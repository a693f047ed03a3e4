"""The CPU oracle against the committed golden vectors (tests/golden/*.npz, made by tools/make_golden.py).
esti_plane.npz and ivox_knn.npz were produced by the REFERENCE'S OWN CODE; they pin the oracle wherever the
reference tree is not mounted (e.g. on the GPU box)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_esti_plane_golden(oracle_mod):
    d = np.load(os.path.join(G, "esti_plane.npz"))
    for pts, ok, pabcd in zip(d["points"], d["ok"], d["pabcd"]):
        o, r = oracle_mod.esti_plane(pts)
        assert o == bool(ok)
        if np.all(np.isfinite(pabcd)):
            assert np.array_equal(r.view(np.uint32), pabcd.view(np.uint32))
    assert 0 < d["ok"].sum() < len(d["ok"])


def test_ivox_knn_golden(oracle_mod):
    d = np.load(os.path.join(G, "ivox_knn.npz"))
    for st in (19, 75):
        iv = oracle_mod.IVox(stencil=st)
        iv.add(d["map"][:5000], 0.0)
        iv.add(d["map"][5000:], 1.0)
        assert iv.num_voxels == int(d[f"voxels{st}"])
        nn, cnt, _ = iv.knn(d["queries"])
        assert np.array_equal(cnt, d[f"cnt{st}"])
        assert np.array_equal(nn[..., :3].view(np.uint32), d[f"nn{st}"].view(np.uint32))


def test_voxelgrid_golden(oracle_mod):
    d = np.load(os.path.join(G, "voxelgrid.npz"))
    ds = oracle_mod.voxel_downsample(d["raw"], float(d["leaf"]))
    assert np.array_equal(ds.view(np.uint32), d["ds"].view(np.uint32))
    # first-principles properties of the filter: every centroid lies in its voxel, voxels are unique and ascending
    raw = d["raw"][np.isfinite(d["raw"]).all(1)]
    inv = np.float32(1.0) / np.float32(d["leaf"])
    mn = np.floor(raw[:, :3].min(0) * inv)
    ijk = np.floor(ds[:, :3] * inv) - mn
    div = np.floor(raw[:, :3].max(0) * inv) - mn + 1
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    assert np.all(np.diff(idx) > 0)
    rijk = np.floor(raw[:, :3] * inv) - mn
    ridx = np.unique(rijk[:, 0] + rijk[:, 1] * div[0] + rijk[:, 2] * div[0] * div[1])
    assert len(ridx) == len(ds)


def _lio(oracle_mod, d, state):
    o = oracle_mod.Lio(stencil=19, capacity=1 << 40, threads=4)
    o.map_add(d["map"])
    o.set_state(state)
    o.set_cov(oracle_mod.init_cov())
    o.set_flags(ekf_inited=True, first_scan=False)
    o.set_ds(d["ds"])
    return o


def test_linearize_and_update_golden(oracle_mod):
    d = np.load(os.path.join(G, "linearize.npz"))
    u = np.load(os.path.join(G, "update.npz"))
    o = _lio(oracle_mod, d, d["state"])
    lin = o.linearize(converge=True)
    assert np.array_equal(lin["selected"], d["selected"]) and lin["n_eff"] == int(d["n_eff"])
    sel = d["selected"].astype(bool)
    assert np.array_equal(lin["normvec"][sel].view(np.uint32), d["normvec"][sel].view(np.uint32))
    assert np.array_equal(lin["nn"][..., :3].view(np.uint32), d["nn"].view(np.uint32))
    assert np.allclose(lin["JtJ"], d["JtJ"], rtol=1e-13, atol=0) and np.allclose(lin["Jtr"], d["Jtr"], rtol=1e-12, atol=1e-12)
    o2 = _lio(oracle_mod, d, u["state0"])
    logs = o2.update()
    assert [l["knn"] for l in logs] == list(u["knn"]) and [l["n_eff"] for l in logs] == list(u["n_eff"])
    assert np.allclose(o2.get_state(), u["state1"], rtol=0, atol=1e-12)
    assert np.allclose(o2.get_cov(), u["P1"], rtol=1e-9, atol=1e-15)
    # and the registration means something: the update moved the guess onto the true pose
    from lsd_amd import synth

    assert np.linalg.norm(u["state1"][:3] - u["true_pos"]) < 0.03 < np.linalg.norm(u["state0"][:3] - u["true_pos"])
    assert synth.quat_angle(u["state1"][3:7], u["true_q"]) < 3e-3


def test_ndt_golden():
    import ndt as ondt

    d = np.load(os.path.join(G, "ndt.npz"))
    n = ondt.Ndt(1.0, 7)
    n.set_target(d["map"])
    n.set_source(d["ds"])
    assert n.num_voxels == int(d["n_voxels"])
    lin = n.linearize(d["T_guess"])
    assert lin["n_corr"] == int(d["n_corr"])
    assert np.allclose(lin["H"], d["H"], rtol=1e-12) and np.allclose(lin["b"], d["b"], rtol=1e-12, atol=1e-9)
    T, conv, its = n.align(d["T_guess"])
    assert conv == bool(d["converged"]) and its == int(d["iterations"])
    assert np.allclose(T, d["T_aligned"], atol=1e-12)
    # the matcher pulls the guess onto the true pose (NDT at 1 m resolution: decimetre accuracy)
    assert np.linalg.norm(T[:3, 3] - d["T_true"][:3, 3]) < 0.1 < np.linalg.norm(d["T_guess"][:3, 3] - d["T_true"][:3, 3])


def test_ndt_cost_decreases_and_gradient_matches_finite_differences():
    """first principles: b is the gradient of the robustified cost wrt a left-multiplied twist, up to the treatment of
    the Cauchy weight as a constant (so compare the direction); an LM step must not increase the cost"""
    import ndt as ondt

    d = np.load(os.path.join(G, "ndt.npz"))
    n = ondt.Ndt(1.0, 7)
    n.set_target(d["map"])
    n.set_source(d["ds"])
    lin = n.linearize(d["T_guess"])
    g = np.zeros(6)
    eps = 1e-4
    for k in range(6):
        a = np.zeros(6)
        a[k] = eps
        g[k] = (n.compute_error(ondt.se3_exp(a) @ d["T_guess"]) - n.compute_error(ondt.se3_exp(-a) @ d["T_guess"])) / (2 * eps)
    cosang = float(g @ (2 * lin["b"]) / (np.linalg.norm(g) * np.linalg.norm(2 * lin["b"])))
    assert cosang > 0.8
    step = np.linalg.solve(lin["H"] + 1e-6 * np.eye(6), -lin["b"])
    assert n.compute_error(ondt.se3_exp(step) @ d["T_guess"]) < lin["err"]
