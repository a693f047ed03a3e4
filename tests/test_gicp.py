"""Fine matcher (GICP), CPU side: the numpy restatement oracle/gicp.py against the vectors the reference itself wrote
(tests/golden/gicp.npz, fast_gicp::FastGICP compiled from /root/reference by oracle/ref_gicp.cpp, tools/make_gicp_golden.py), and -- where
the harness is present -- the harness against those vectors again (the fixtures are what the reference computes today)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import gicp as OG  # noqa: E402
import gicp_cases  # noqa: E402
import ref_gicp  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "gicp.npz"))


def _nontie_rows(cloud, k):
    """points whose k-th and (k+1)-th neighbour distances differ: the neighbour SET is then unique (the tie order of a kd-tree is not)"""
    _, d = OG.knn(np.asarray(cloud, np.float32), k)
    return d[:, k - 1] < d[:, k]


@pytest.mark.parametrize("name", ["room_small", "room_k10"])
def test_oracle_vs_reference_vectors(name):
    c = gicp_cases.make(name)
    g = OG.Gicp(k=c["k"], max_corr_dist=c["max_corr_dist"])
    cov_t, cov_s = g.set_target(c["target"]), g.set_source(c["source"])
    for cov, ref, cloud in ((cov_t, GOLD[name + "/cov_tgt"], c["target"]), (cov_s, GOLD[name + "/cov_src"], c["source"])):
        ok = _nontie_rows(cloud, c["k"])
        assert ok.mean() > 0.99
        d = np.abs(cov - ref).reshape(len(cov), -1).max(1)
        # the regularised covariance depends on the direction of least spread only: ill-conditioned where the two smallest singular values
        # almost coincide -- a handful of points, bounded loosely; everywhere else to rounding
        assert np.quantile(d[ok], 0.99) < 1e-9 and d[ok].max() < 1e-5, (np.quantile(d[ok], 0.99), d[ok].max())
    e, H, b = g.linearize(c["guess"])
    assert np.array_equal(g.corr, GOLD[name + "/corr"])
    assert np.array_equal(g.sq, GOLD[name + "/sq"])
    has = g.corr >= 0
    assert np.abs(g.maha[has] - GOLD[name + "/maha"][has]).max() < 1e-6 * np.abs(GOLD[name + "/maha"][has]).max()
    assert abs(e - GOLD[name + "/err"]) < 1e-7 * abs(GOLD[name + "/err"])
    assert np.abs(H - GOLD[name + "/H"]).max() < 1e-7 * np.abs(GOLD[name + "/H"]).max()
    assert np.abs(b - GOLD[name + "/b"]).max() < 1e-7 * np.abs(GOLD[name + "/b"]).max()
    T2 = c["guess"].copy()
    T2[:3, 3] += [0.01, -0.02, 0.005]
    assert abs(g.compute_error(T2) - GOLD[name + "/err2"]) < 1e-7 * abs(GOLD[name + "/err2"])


def test_oracle_align_vs_reference_vectors():
    name = "room_small"
    c = gicp_cases.make(name)
    g = OG.Gicp(k=c["k"], max_corr_dist=c["max_corr_dist"])
    g.set_target(c["target"])
    g.set_source(c["source"])
    T, it, conv = g.align(c["guess"].astype(np.float32))
    assert conv == bool(GOLD[name + "/converged"]) and it == int(GOLD[name + "/iterations"])
    assert np.abs(T - GOLD[name + "/T"]).max() < 1e-5


@pytest.mark.skipif(not ref_gicp.available(), reason="oracle/_ref/libref_gicp.so not built (needs /root/reference: make -C oracle ref)")
@pytest.mark.parametrize("name", list(gicp_cases.CASES))
def test_reference_reproduces_vectors(name):
    c = gicp_cases.make(name)
    g = ref_gicp.RefGicp(k=c["k"], max_corr_dist=c["max_corr_dist"], num_threads=1)
    assert np.array_equal(g.set_target(c["target"]), GOLD[name + "/cov_tgt"])
    assert np.array_equal(g.set_source(c["source"]), GOLD[name + "/cov_src"])
    e, H, b, corr, sq, maha = g.linearize(c["guess"])
    assert np.array_equal(corr, GOLD[name + "/corr"]) and np.array_equal(sq, GOLD[name + "/sq"])
    assert e == float(GOLD[name + "/err"]) and np.array_equal(H, GOLD[name + "/H"]) and np.array_equal(b, GOLD[name + "/b"])
    T, it, conv = g.align(c["guess"].astype(np.float32))
    assert np.array_equal(T, GOLD[name + "/T"]) and it == int(GOLD[name + "/iterations"]) and conv == bool(GOLD[name + "/converged"])


@pytest.mark.skipif(not ref_gicp.available(), reason="oracle/_ref/libref_gicp.so not built")
def test_kdtree_standin_is_exact():
    """the grid search standing in for pcl::search::KdTree returns the brute-force neighbour sets, for any cell size"""
    c = gicp_cases.make("room_small")
    covs = []
    for cell in (0.7, 1.0, 3.0):
        g = ref_gicp.RefGicp(k=c["k"], num_threads=1, kdtree_cell=cell)
        covs.append(g.set_target(c["target"]))
    assert np.array_equal(covs[0], covs[1]) and np.array_equal(covs[1], covs[2])


VG = [("room_small", 1), ("room_small", 7), ("room_fine", 1), ("room_fine", 27)]


@pytest.mark.parametrize("name,sm", [("room_small", 1), ("room_small", 7)])
def test_oracle_vgicp_vs_reference_vectors(name, sm):
    """the voxelised variant (fast_gicp::FastVGICP): Gaussian voxels, voxel correspondences, weighted cost -- numpy restatement vs the reference"""
    c = gicp_cases.make(name)
    key = f"vgicp/{name}/{sm}/"
    v = OG.Vgicp(k=c["k"], resolution=1.0, search_method=sm)
    v.set_target(c["target"])
    v.set_source(c["source"])
    for p, n, m, C in zip(GOLD[key + "probe"], GOLD[key + "vox_n"], GOLD[key + "vox_mean"], GOLD[key + "vox_cov"]):
        n2, m2, C2 = v.voxel_at(p)
        assert n2 == n and np.abs(m2 - m).max() < 1e-12 and np.abs(C2 - C).max() < 1e-6
    e, H, b = v.linearize(c["guess"])
    assert len(v.vcorr) == int(GOLD[key + "n_corr"])
    assert abs(e - GOLD[key + "err"]) < 1e-6 * abs(GOLD[key + "err"])
    assert np.abs(H - GOLD[key + "H"]).max() < 1e-6 * np.abs(GOLD[key + "H"]).max()
    assert np.abs(b - GOLD[key + "b"]).max() < 1e-6 * np.abs(GOLD[key + "b"]).max()
    T2 = c["guess"].copy()
    T2[:3, 3] += [0.01, -0.02, 0.005]
    assert abs(v.compute_error(T2) - GOLD[key + "err2"]) < 1e-6 * abs(GOLD[key + "err2"])
    if sm == 1:
        T, it, conv = v.align(c["guess"].astype(np.float32))
        assert conv == bool(GOLD[key + "converged"]) and it == int(GOLD[key + "iterations"])
        assert np.abs(T - GOLD[key + "T"]).max() < 1e-5


@pytest.mark.skipif(not ref_gicp.available(), reason="oracle/_ref/libref_gicp.so not built")
@pytest.mark.parametrize("name,sm", VG)
def test_reference_vgicp_reproduces_vectors(name, sm):
    c = gicp_cases.make(name)
    key = f"vgicp/{name}/{sm}/"
    v = ref_gicp.RefVgicp(k=c["k"], resolution=1.0, search_method=sm, num_threads=1)
    v.set_target(c["target"])
    v.set_source(c["source"])
    e, H, b, nc = v.linearize(c["guess"])
    assert nc == int(GOLD[key + "n_corr"]) and e == float(GOLD[key + "err"]) and np.array_equal(H, GOLD[key + "H"]) and np.array_equal(b, GOLD[key + "b"])
    T, it, conv = v.align(c["guess"].astype(np.float32))
    assert np.array_equal(T, GOLD[key + "T"]) and it == int(GOLD[key + "iterations"])
