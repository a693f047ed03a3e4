"""The degeneracy branch of h_share_model_geometric (laserMapping.cpp:934-980) and the N_eff < 23 branch of the filter
(esekfom.hpp:1715-1744): scenes in which they FIRE.  CPU: the oracle against the reference's own translation units
(oracle/_ref/libref_fastlio.so).  GPU: the HIP path against the oracle through the C ABI."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenes  # noqa: E402


def _oracle_run(oracle_mod, case):
    o = oracle_mod.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=8)
    o.map_add(case["map"])
    o.set_state(case["guess"])
    o.set_cov(oracle_mod.init_cov())
    o.set_flags(ekf_inited=True, first_scan=False)
    ds = oracle_mod.voxel_downsample(case["raw"], 0.5)
    o.set_ds(ds)
    logs = o.update()
    return o, ds, logs


EXPECT = {  # per case: degenerate flag of the filter's passes, and per direction (ascending eigenvalue) of the last pass: dropped?
    "open_ground": (True, [True, True, False]),
    "box_6x3": (True, [True, False, False]),
    "box_12x4": (True, [True, False, False]),
    "box_20x6": (True, [True, False, False]),
}


@pytest.mark.parametrize("name", list(scenes.DEGENERATE_CASES))
def test_oracle_degeneracy_decisions(oracle_mod, name):
    """the scenes are what they claim to be (so that the GPU test below exercises each side of the decision)"""
    from lsd_amd import synth

    case = scenes.degenerate_case(name)
    o, ds, logs = _oracle_run(oracle_mod, case)
    assert all(p["degenerate"] == int(EXPECT[name][0]) for p in logs) and o.is_degenerate
    d = o.last_degeneracy()
    dropped = [bool(c < 250.0 and s < 50.0) for c, s in zip(d["contri"], d["strong"])]
    assert dropped == EXPECT[name][1]
    n_eff = logs[-1]["n_eff"]
    bound = d["eigval"] - 0.030138 * n_eff
    if name == "box_6x3":
        assert d["contri"][1] < 250.0 <= 250.0 and d["strong"][1] >= 50.0
    if name == "box_12x4":
        assert d["contri"][1] >= 250.0 and bound[1] < 250.0       # the shortcut cannot decide: the sums are evaluated
    if name == "box_20x6":
        assert bound[1] >= 250.0 and bound[2] >= 250.0 and bound[0] < 250.0
    # the projected normal equations have lost the dropped directions: J^T J [0:3, 0:3] has rank 3 - #dropped
    w = np.linalg.eigvalsh(logs[-1]["JtJ"][:3, :3])
    assert (w < 1e-6 * w.max()).sum() == sum(dropped)
    # and the pose did not move along them: what is left of the initial horizontal error stays (nothing observes it)
    so = o.get_state()
    if name == "open_ground":
        g = case["guess"]
        assert np.linalg.norm((so[:3] - g[:3])[:2]) < 0.02 and abs(so[2] - case["true_pos"][2]) < 0.01
        assert synth.quat_angle(so[3:7], case["true_q"]) < 0.02


@pytest.mark.parametrize("name", ["open_ground", "box_6x3", "box_12x4"])
def test_oracle_vs_reference_on_degenerate_scenes(oracle_mod, name):
    """the reference's own laserMapping.cpp (h_share_model with its degeneracy block, :934-980) on the same scenes"""
    ref_fastlio = pytest.importorskip("ref_fastlio")
    if not ref_fastlio.available():
        pytest.skip("oracle/_ref/libref_fastlio.so not built (needs /root/reference)")
    from lsd_amd import synth

    case = scenes.degenerate_case(name)
    o, ds, logs = _oracle_run(oracle_mod, case)
    R = ref_fastlio.RefFastLio()
    R.map_add(case["map"])
    R.set_nearby(18)
    R.set_canonical(True)
    R.calls(clear=True)
    rc, sr, Pr = R.register(case["raw"], case["guess"], oracle_mod.init_cov())
    assert rc == 3
    calls = R.calls()
    R.set_canonical(False)
    assert R.info()["degenerate"] is True
    assert [c["degenerate"] for c in calls] == [bool(p["degenerate"]) for p in logs]
    assert [c["n_eff"] for c in calls] == [p["n_eff"] for p in logs]
    for c, p in zip(calls, logs):  # the reference's projected h_x^T h_x / h_x^T h against the oracle's
        scale = np.abs(p["JtJ"]).max()
        assert np.abs(c["HtH"] - p["JtJ"]).max() < 1e-9 * scale
        assert np.abs(c["Hth"] - p["Jtr"]).max() < 1e-9 * max(1.0, np.abs(p["Jtr"]).max())
    so = o.get_state()
    assert np.linalg.norm(so[:3] - sr[:3]) < 1e-4 and synth.quat_angle(so[3:7], sr[3:7]) < 1e-5


def _sparse_case(n_az=6, n_beams=8):
    return scenes.degenerate_case("open_ground", n_az=n_az, n_beams=n_beams)


def test_oracle_sparse_scan_takes_the_dense_branch(oracle_mod):
    case = _sparse_case()
    o, ds, logs = _oracle_run(oracle_mod, case)
    assert 5 <= len(ds) and all(p["valid"] and 0 < p["n_eff"] < 23 for p in logs)
    assert o.is_degenerate  # N_eff < 23 < 50: every direction fails contri < 250 && strong < 50 -> h_x[:, 0:3] = 0
    assert np.abs(logs[-1]["JtJ"][:3, :]).max() == 0.0 and np.abs(logs[-1]["JtJ"][3:, 3:]).max() > 0.0


# ---------------------------------------------------------------------------------------------------------------------
def _dev():
    from lsd_amd import capi

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device")


def _hip_run(case, ds):
    from lsd_amd import lio

    e = lio.Engine(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=500_000, max_raw=1 << 18, max_ds=100000)
    e.map_add(case["map"])
    e.set_state(case["guess"])
    e.set_cov(lio.init_cov())
    e.set_flags(ekf_inited=True, first_scan=False)
    e.set_ds(ds)
    return e


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(scenes.DEGENERATE_CASES))
def test_hip_degenerate_projection_matches_oracle(oracle_mod, name, loop_mode):
    _dev()
    from lsd_amd import lio, synth

    case = scenes.degenerate_case(name)
    o, ds, logs = _oracle_run(oracle_mod, case)
    e = _hip_run(case, ds)
    # one forced evaluation of the six sums at the prior state against the oracle's f32 sums
    o2 = oracle_mod.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=8)
    o2.map_add(case["map"]); o2.set_state(case["guess"]); o2.set_cov(oracle_mod.init_cov()); o2.set_flags(ekf_inited=True, first_scan=False)
    o2.set_ds(ds)
    r0 = o2.linearize(converge=True)
    d0 = o2.last_degeneracy()
    e.scan.set_degeneracy_mode(1)
    g0 = lio.linearize(e.map, e.scan, case["guess"], redo_knn=True)
    e.scan.set_degeneracy_mode(0)
    assert g0["n_eff"] == r0["n_eff"]
    assert np.allclose(g0["eigval"], d0["eigval"], rtol=1e-9, atol=1e-9)
    assert np.allclose(g0["contri"], d0["contri"], rtol=0, atol=2e-2) and np.allclose(g0["strong"], d0["strong"], rtol=0, atol=2e-2)  # f32 running sums there
    e.scan.reset()
    e.set_ds(ds)
    lg = e.update()
    assert len(lg) == len(logs)
    for a, b in zip(logs, lg):
        assert (a["knn"], a["n_eff"], a["valid"], a["degenerate"]) == (b["knn"], b["n_eff"], b["valid"], b["degenerate"])
        scale = np.abs(a["JtJ"]).max()
        assert np.abs(a["JtJ"] - b["JtJ"]).max() < 1e-10 * scale            # after the projection M JtJ M^T
        assert np.abs(a["Jtr"] - b["Jtr"]).max() < 1e-10 * max(1.0, np.abs(a["Jtr"]).max())
        assert np.allclose(a["dx"], b["dx"], rtol=0, atol=1e-9)
    assert e.is_degenerate and o.is_degenerate
    so, sg = o.get_state(), e.get_state()
    assert np.linalg.norm(so[:3] - sg[:3]) < 1e-8 and synth.quat_angle(so[3:7], sg[3:7]) < 1e-8
    # the projected H^T H is singular: the oracle inverts (P/R)^-1 + H^T H twice (23 x 23), the product uses the 6 x 6 information form;
    # entries of the posterior that are ~1e-9 (against a diagonal of ~5e-5) agree to ~2e-12 absolute
    Po, Pe = o.get_cov(), e.get_cov()
    print(name, "max |dP|", np.abs(Po - Pe).max(), "max |P|", np.abs(Po).max())
    assert np.abs(Po - Pe).max() < 1e-8 * max(1.0, np.abs(Po).max())


@pytest.mark.gpu
def test_hip_sparse_scan_dense_branch_matches_oracle(oracle_mod, loop_mode):
    """N_eff < 23: lio_p2plane_rows + the dense gain of esekfom.hpp:1715-1744 on the device path, degeneracy-projected rows"""
    _dev()
    from lsd_amd import synth

    for n_az, n_beams in ((6, 8), (8, 6), (10, 4)):  # (8, 6): 23 rows (information form) in the early passes, 21 (dense) in the late ones
        case = _sparse_case(n_az, n_beams)
        o, ds, logs = _oracle_run(oracle_mod, case)
        assert any(p["valid"] and p["n_eff"] < 23 for p in logs)
        e = _hip_run(case, ds)
        lg = e.update()
        assert len(lg) == len(logs)
        for a, b in zip(logs, lg):
            assert (a["knn"], a["n_eff"], a["valid"], a["degenerate"]) == (b["knn"], b["n_eff"], b["valid"], b["degenerate"])
            assert np.abs(a["JtJ"] - b["JtJ"]).max() < 1e-10 * np.abs(a["JtJ"]).max()   # (rows projected, then summed / sums, then projected)
            assert np.abs(a["JtJ"][:3, :]).max() == 0.0 and np.abs(b["JtJ"][:3, :]).max() < 1e-12 * np.abs(a["JtJ"]).max()
            assert np.allclose(a["dx"], b["dx"], rtol=0, atol=1e-9)
        so, sg = o.get_state(), e.get_state()
        assert np.linalg.norm(so[:3] - sg[:3]) < 1e-8 and synth.quat_angle(so[3:7], sg[3:7]) < 1e-8
        Po, Pe = o.get_cov(), e.get_cov()
        print("sparse", n_az, n_beams, "max |dP|", np.abs(Po - Pe).max())
        assert np.abs(Po - Pe).max() < 1e-8 * max(1.0, np.abs(Po).max())
