"""Fine matcher (GICP) on the device (lio_gicp_*, csrc/gicp.hip) against the reference: the vectors fast_gicp::FastGICP itself produced
(tests/golden/gicp.npz) and, at a larger size, the reference harness run side by side (oracle/_ref/libref_gicp.so travels to the GPU box)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import gicp as OG  # noqa: E402
import gicp_cases  # noqa: E402
import ref_gicp  # noqa: E402
from lsd_amd import lio  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(HERE, "golden", "gicp.npz"))


def _order(dev_pts, pts):
    """perm with dev_pts[i] == pts[perm[i]] (the device keeps a cloud in hash-grid order); the clouds have no duplicate points"""
    key = lambda a: np.ascontiguousarray(a[:, :3]).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel()
    ka, kb = key(dev_pts), key(pts)
    ob = np.argsort(kb)
    pos = np.searchsorted(kb[ob], ka)
    perm = ob[pos]
    assert np.array_equal(pts[perm, :3], dev_pts[:, :3])
    return perm


def _check_against(c, ref, grid):
    """ref: dict cov_tgt, cov_src, corr, err, H, b, err2, T, iterations, converged in INPUT order"""
    g = lio.Gicp(grid_resolution=grid, max_points=max(len(c["target"]), len(c["source"])), k=c["k"])
    g.set_target(c["target"])
    g.set_source(c["source"])
    tp, tcov = g.download(0)
    sp, scov = g.download(1)
    pt, ps = _order(tp, c["target"]), _order(sp, c["source"])
    for cov, r, cloud, perm in ((tcov, ref["cov_tgt"], c["target"], pt), (scov, ref["cov_src"], c["source"], ps)):
        d = np.abs(cov - r[perm]).reshape(len(cov), -1).max(1)
        if len(cloud) <= 6000:  # where the k-th neighbour is tied the neighbour set is not unique
            _, dk = OG.knn(np.asarray(cloud, np.float32), c["k"])
            d = d[(dk[:, c["k"] - 1] < dk[:, c["k"]])[perm]]
        assert np.quantile(d, 0.99) < 1e-9 and d.max() < 1e-5, (np.quantile(d, 0.99), d.max())
    maxd = c["max_corr_dist"]
    r = g.linearize(c["guess"], max_corr_dist=maxd)
    corr = g.correspondences()
    # device correspondences index the device's target order, source rows in the device's source order
    want = ref["corr"][ps]
    got = np.where(corr >= 0, pt[np.maximum(corr, 0)], -1)
    assert np.array_equal(got, want), int((got != want).sum())
    assert r["n_corr"] == int((want >= 0).sum())
    assert abs(r["err"] - ref["err"]) < 1e-8 * abs(ref["err"])
    assert np.abs(r["H"] - ref["H"]).max() < 1e-8 * np.abs(ref["H"]).max()
    assert np.abs(r["b"] - ref["b"]).max() < 1e-8 * np.abs(ref["b"]).max()
    T2 = c["guess"].copy()
    T2[:3, 3] += [0.01, -0.02, 0.005]
    r2 = g.linearize(T2, max_corr_dist=maxd, update_corr=False, with_derivatives=False)
    assert abs(r2["err"] - ref["err2"]) < 1e-8 * abs(ref["err2"])
    T, conv, it = g.align(c["guess"].astype(np.float32).astype(np.float64), max_corr_dist=maxd)
    assert conv == bool(ref["converged"]) and it == int(ref["iterations"]), (conv, it)
    assert np.abs(T[:3, 3] - ref["T"][:3, 3]).max() < 1e-4       # BASELINE.json tolerance: 1e-4 m, 1e-5 rad
    assert np.abs(T[:3, :3] - ref["T"][:3, :3]).max() < 1e-5
    # run-to-run identical (fixed fold order)
    T_again, _, _ = g.align(c["guess"].astype(np.float32).astype(np.float64), max_corr_dist=maxd)
    assert np.array_equal(T, T_again)
    g.close()


@pytest.mark.parametrize("name", list(gicp_cases.CASES))
@pytest.mark.parametrize("grid", [1.0, 0.6])
def test_gicp_vs_reference_vectors(name, grid):
    c = gicp_cases.make(name)
    ref = {k: GOLD[name + "/" + k] for k in ("cov_tgt", "cov_src", "corr", "err", "H", "b", "err2", "T", "iterations", "converged")}
    _check_against(c, ref, grid)


@pytest.mark.skipif(not ref_gicp.available(), reason="oracle/_ref/libref_gicp.so not built")
def test_gicp_vs_reference_at_merge_size():
    """two 64 x 1875 scans thinned to 0.3 m (about 30k points each), the reference harness run next to the device on the same input"""
    from lsd_amd import synth
    sc = synth.Scene(half=60.0, n_boxes=30, seed=9)
    pa, qa = np.array([0.5, -1.0, 1.8]), synth.quat_from_rotvec([0, 0, 0.2])
    pb, qb = np.array([2.0, -0.2, 1.8]), synth.quat_from_rotvec([0.01, -0.02, 0.4])
    ra, _ = synth.make_scan(sc, pa, qa, seed=21, max_range=80.0)
    rb, _ = synth.make_scan(sc, pb, qb, seed=22, max_range=80.0)
    c = dict(target=gicp_cases._thin(ra[:, :4].astype(np.float32), 0.3), source=gicp_cases._thin(rb[:, :4].astype(np.float32), 0.3), k=20, max_corr_dist=2.0)
    truth = np.linalg.inv(gicp_cases._pose(pa, qa)) @ gicp_cases._pose(pb, qb)
    c["guess"] = truth @ gicp_cases._pose([0.2, -0.1, 0.05], synth.quat_from_rotvec([0.005, 0.01, -0.02]))
    h = ref_gicp.RefGicp(k=20, max_corr_dist=2.0, num_threads=4)
    ref = dict(cov_tgt=h.set_target(c["target"]), cov_src=h.set_source(c["source"]))
    e, H, b, corr, _, _ = h.linearize(c["guess"])
    T2 = c["guess"].copy()
    T2[:3, 3] += [0.01, -0.02, 0.005]
    ref.update(err=e, H=H, b=b, corr=corr, err2=h.compute_error(T2))
    T, it, conv = h.align(c["guess"].astype(np.float32))
    ref.update(T=T, iterations=it, converged=conv)
    assert np.abs(T[:3, 3] - truth[:3, 3]).max() < 0.02
    _check_against(c, ref, 1.0)


@pytest.mark.parametrize("name,sm", [("room_small", 1), ("room_small", 7), ("room_fine", 1), ("room_fine", 27)])
def test_vgicp_vs_reference_vectors(name, sm):
    """lio_gicp_* in voxel mode = fast_gicp::FastVGICP (fast_vgicp_impl.hpp:72-204): Gaussian voxels of the target, voxel correspondences,
    weighted cost / H / b, whole alignments -- against the vectors the reference wrote (configured as select_registration_method("FAST_VGICP"))"""
    c = gicp_cases.make(name)
    key = f"vgicp/{name}/{sm}/"
    g = lio.Gicp(grid_resolution=1.0, max_points=max(len(c["target"]), len(c["source"])), k=c["k"])
    g.set_voxel_mode(1.0, sm)
    g.set_target(c["target"])
    g.set_source(c["source"])
    _, tcov = g.download(0)
    for p, n, m, C in zip(GOLD[key + "probe"], GOLD[key + "vox_n"], GOLD[key + "vox_mean"], GOLD[key + "vox_cov"]):
        n2, m2, C2 = g.voxel_at(p)
        assert n2 == n and np.abs(m2 - m).max() < 1e-12 and np.abs(C2 - C).max() < 1e-6
    r = g.linearize(c["guess"])
    assert r["n_corr"] == int(GOLD[key + "n_corr"])
    assert abs(r["err"] - GOLD[key + "err"]) < 1e-7 * abs(GOLD[key + "err"])
    assert np.abs(r["H"] - GOLD[key + "H"]).max() < 1e-7 * np.abs(GOLD[key + "H"]).max()
    assert np.abs(r["b"] - GOLD[key + "b"]).max() < 1e-7 * np.abs(GOLD[key + "b"]).max()
    T2 = c["guess"].copy()
    T2[:3, 3] += [0.01, -0.02, 0.005]
    r2 = g.linearize(T2, update_corr=False, with_derivatives=False)
    assert abs(r2["err"] - GOLD[key + "err2"]) < 1e-7 * abs(GOLD[key + "err2"])
    T, conv, it = g.align(c["guess"].astype(np.float32).astype(np.float64), transformation_epsilon=0.1, rotation_epsilon_deg=0.1)
    assert conv == bool(GOLD[key + "converged"]) and it == int(GOLD[key + "iterations"])
    assert np.abs(T[:3, 3] - GOLD[key + "T"][:3, 3]).max() < 1e-4 and np.abs(T[:3, :3] - GOLD[key + "T"][:3, :3]).max() < 1e-5
    # back to the kd-tree form on the same object
    g.set_voxel_mode(0.0, 1)
    r3 = g.linearize(c["guess"], max_corr_dist=c["max_corr_dist"])
    assert abs(r3["err"] - GOLD[name + "/err"]) < 1e-8 * abs(GOLD[name + "/err"])
    g.close()
