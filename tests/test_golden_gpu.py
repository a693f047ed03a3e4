"""The HIP path against the committed golden vectors (two of them produced by the reference's own code)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ivox_knn_golden_gpu():
    from lsd_amd import lio

    d = np.load(os.path.join(G, "ivox_knn.npz"))
    for st in (19, 75):
        m = lio.Map(stencil=st, max_points=100_000, max_voxels=50_000)
        m.add(d["map"][:5000], 0.0)
        m.add(d["map"][5000:], 1.0)
        assert m.num_voxels == int(d[f"voxels{st}"])
        nn, cnt = m.knn(d["queries"])
        assert np.array_equal(cnt, d[f"cnt{st}"])
        assert np.array_equal(nn[..., :3].view(np.uint32), d[f"nn{st}"].view(np.uint32))


def test_voxelgrid_golden_gpu():
    from lsd_amd import lio

    d = np.load(os.path.join(G, "voxelgrid.npz"))
    s = lio.Scan(max_raw=1 << 16, max_ds=1 << 16)
    s.upload(d["raw"])
    n = s.voxel_downsample(float(d["leaf"]))
    assert n == len(d["ds"])
    assert np.array_equal(s.get_ds().view(np.uint32), d["ds"].view(np.uint32))


def test_linearize_update_golden_gpu():
    """esti_plane is exercised through the linearisation: plane parameters and gates bit-exact, sums to 1e-10"""
    from lsd_amd import lio, synth

    d = np.load(os.path.join(G, "linearize.npz"))
    u = np.load(os.path.join(G, "update.npz"))
    e = lio.Engine(stencil=19, max_points=200_000, max_voxels=100_000, max_raw=1 << 16, max_ds=1 << 16)
    e.map_add(d["map"])
    e.set_flags(ekf_inited=True, first_scan=False)
    e.set_ds(d["ds"])
    e.set_state(d["state"])
    e.set_cov(lio.init_cov())
    got = lio.linearize(e.map, e.scan, d["state"], redo_knn=True)
    mt = e.scan.get_match()
    assert np.array_equal(mt["selected"], d["selected"]) and got["n_eff"] == int(d["n_eff"])
    sel = d["selected"].astype(bool)
    assert np.array_equal(mt["normvec"][sel].view(np.uint32), d["normvec"][sel].view(np.uint32))
    assert np.array_equal(mt["nn"][..., :3].view(np.uint32), d["nn"].view(np.uint32))
    assert np.allclose(got["JtJ"], d["JtJ"], rtol=1e-10, atol=1e-9) and np.allclose(got["Jtr"], d["Jtr"], rtol=1e-10, atol=1e-9)
    e.scan.reset()
    e.set_ds(d["ds"])
    e.set_state(u["state0"])
    e.set_cov(u["P0"])
    logs = e.update()
    assert [l["knn"] for l in logs] == list(u["knn"]) and [l["n_eff"] for l in logs] == list(u["n_eff"])
    s1 = e.get_state()
    assert np.linalg.norm(s1[:3] - u["state1"][:3]) < 1e-4 and synth.quat_angle(s1[3:7], u["state1"][3:7]) < 1e-5
    assert np.abs(s1 - u["state1"]).max() < 1e-8
    assert np.allclose(e.get_cov(), u["P1"], rtol=1e-6, atol=1e-12)


def test_ndt_golden_gpu():
    from lsd_amd import lio

    d = np.load(os.path.join(G, "ndt.npz"))
    g = lio.Ndt(resolution=1.0, search_method=7, max_points=100_000, max_voxels=50_000, max_source_points=1 << 16)
    g.set_target(d["map"])
    assert g.num_voxels == int(d["n_voxels"])
    s = lio.Scan(max_raw=1 << 16, max_ds=1 << 16)
    s.set_ds(d["ds"])
    lin = g.linearize(s, d["T_guess"])
    assert lin["n_corr"] == int(d["n_corr"])
    assert np.allclose(lin["H"], d["H"], rtol=1e-3, atol=1e-3 * np.abs(d["H"]).max())
    assert np.allclose(lin["b"], d["b"], rtol=1e-3, atol=1e-3 * np.abs(d["b"]).max())
    T, conv, its = g.align(s, d["T_guess"])
    assert conv == bool(d["converged"]) and its == int(d["iterations"])
    R = T[:3, :3] @ d["T_aligned"][:3, :3].T
    assert np.linalg.norm(T[:3, 3] - d["T_aligned"][:3, 3]) < 1e-4 and np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)) < 1e-5
