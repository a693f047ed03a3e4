"""The wheel-speed rows of h_share_model (laserMapping.cpp:794-811, 994-1012): three velocity rows appended to the point-to-plane rows when
wheelspeed_en and the last INS sample of the scan is within 10 ms of its end.  `wheelspeed_en` is a constant false in the reference (its own
binaries never run this branch); the harness flips the file-scope variable so that the branch can be pinned.
CPU: the oracle against the reference's translation units on a drive with INS samples.  GPU: the product (host-driven filter loop over the
device linearisation) against the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import ref_fastlio  # noqa: E402
from test_fastlio_vs_ref import _sweep  # noqa: E402


def _ins_vel(tr, t):
    """what a wheel / INS sensor reports: the body-frame velocity at time t (forward, left; the reference zeroes the third component)"""
    v_w = (tr.pos(t + 1e-4) - tr.pos(t - 1e-4)) / 2e-4
    v_b = tr.R(t).T @ v_w
    return float(v_b[0]), float(v_b[1])


def _drive(oracle_mod, scene, n_scans, wheelspeed, bias=(0.0, 0.0)):
    from lsd_amd import synth

    tr = synth.Trajectory()
    imu = synth.imu_stream(tr, 0.0, n_scans * 0.1 + 0.2, rate=200.0)
    L = oracle_mod.Lio()
    L.frontend_config(extT=(0, 0, 0), extR_xyzw=np.array([0, 0, 0, 1.0]), filter_num=1, scan_period=0.1, undistort=True, max_point_num=-1)
    R = ref_fastlio.RefFastLio(extT=(0, 0, 0), extR=np.eye(3), filter_num=1, max_point_num=-1, scan_period=0.1, undistort=True)
    R.set_canonical(True)
    R.set_wheelspeed(wheelspeed)
    L.set_wheelspeed(wheelspeed)
    ii, out = 0, []
    for k in range(n_scans):
        us = k * 100000
        tb = us / 1e6
        pts, st = _sweep(scene, tr, k, True)
        while ii < len(imu) and imu[ii][0] <= tb + 0.12:
            L.imu_enqueue(*imu[ii])
            R.imu_enqueue(*imu[ii])
            ii += 1
        # one INS sample 2 ms before the end of the sweep (inside the 10 ms window of laserMapping.cpp:798), one 40 ms before it
        for dt_us in (60000, 98000):
            vx, vy = _ins_vel(tr, tb + dt_us / 1e6)
            vx, vy = vx + bias[0], vy + bias[1]
            R.ins_enqueue(True, us + dt_us, 0.0, 0.0, 0.0, vx, vy, 0.3, "Wheel")
            L.ins_enqueue(tb + dt_us / 1e6, [vx, vy, 0.0])
        L.pcl_enqueue(pts, st, tb)
        R.pcl_enqueue(pts, st, us)
        rc = L.frontend_main()
        assert R.main()
        sr, _, Pr = R.state()
        out.append(dict(rc=rc, ref=sr, ref_P=Pr, orc=L.get_state(), orc_P=L.get_cov(), info=R.info()))
    R.set_wheelspeed(False)
    return tr, out


@pytest.mark.skipif(not ref_fastlio.available(), reason="oracle/_ref/libref_fastlio.so not built (needs /root/reference)")
def test_oracle_vs_reference_with_wheelspeed_rows(oracle_mod, scene):
    tr, out = _drive(oracle_mod, scene, 12, True, bias=(0.05, -0.03))
    assert [o["rc"] for o in out[7:]] == [3] * 5
    d = [np.abs(o["ref"] - o["orc"]).max() for o in out]
    # same amplification of last-bit differences as without the rows (test_fastlio_vs_ref.py::test_drive_in_canonical_order)
    assert d[7] < 1e-13 and max(d[8:10]) < 1e-9 and max(d[10:12]) < 1e-6, d
    assert np.abs(out[7]["ref_P"] - out[7]["orc_P"]).max() < 1e-10
    # and the rows matter: with a biased wheel velocity the estimated velocity is pulled towards it
    _, off = _drive(oracle_mod, scene, 9, False)
    dv = np.abs(out[8]["orc"][14:17] - off[8]["orc"][14:17]).max()
    assert dv > 1e-4, dv


@pytest.mark.gpu
def test_hip_wheelspeed_rows_match_oracle(oracle_mod, scene):
    """lio_fastlio_set_wheelspeed: the product (device linearisation, host-driven filter over the 9-column active set) against the oracle on a
    14-scan drive with a biased wheel velocity; and the rows change the estimate"""
    from lsd_amd import capi, lio, synth
    from test_frontend_cpu import OracleFront

    if capi.lib().lio_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box")
    tr = synth.Trajectory()
    n = 14

    def drive(wheelspeed):
        e = lio.Engine(max_points=4_000_000, max_voxels=1 << 20, max_raw=1 << 18, max_ds=100000)
        e.fastlio_init(scan_period=0.1)
        e.fastlio_set_wheelspeed(wheelspeed)
        orc = OracleFront(oracle_mod, scan_period=0.1)
        orc.L.set_wheelspeed(wheelspeed)
        imu = synth.imu_stream(tr, 0.0, n * 0.1 + 0.2, rate=200.0)
        ii, out = 0, []
        for k in range(n):
            tb = k * 0.1
            pts, st = synth.make_sweep(scene, tr, tb, n_beams=32, n_az=600, seed=k, fov_deg=(-24.8, 2.0))
            while ii < len(imu) and imu[ii][0] <= tb + 0.12:
                e.fastlio_imu_enqueue(*imu[ii])
                orc.imu_enqueue(*imu[ii])
                ii += 1
            for dt in (0.060, 0.098):
                vx, vy = _ins_vel(tr, tb + dt)
                v = [vx + 0.05, vy - 0.03, 0.0]
                e.fastlio_ins_enqueue(tb + dt, v)
                orc.L.ins_enqueue(tb + dt, v)
            e.fastlio_pcl_enqueue(pts, st, tb)
            orc.pcl_enqueue(pts, st, tb)
            rc = (e.fastlio_main(), orc.main())
            out.append((rc, e.get_state(), orc.get_state(), e.get_cov(), orc.L.get_cov()))
        return out

    on = drive(True)
    assert all(a == b for (a, b), *_ in on)
    assert [a for (a, _), *_ in on][7:] == [capi.MAIN_UPDATED] * (n - 7)
    for k in range(7, n):
        _, sh, so, Ph, Po = on[k]
        dp, dr = np.linalg.norm(sh[0:3] - so[0:3]), synth.quat_angle(sh[3:7], so[3:7])
        assert dp < 1e-4 and dr < 1e-5, (k, dp, dr)            # BASELINE.json tolerance
        assert np.abs(sh[14:17] - so[14:17]).max() < 1e-3
    assert np.abs(on[7][1] - on[7][2]).max() < 1e-9 and np.abs(on[7][3] - on[7][4]).max() < 1e-8 * max(1.0, np.abs(on[7][4]).max())
    off = drive(False)
    assert np.abs(on[8][1][14:17] - off[8][1][14:17]).max() > 1e-4


@pytest.mark.parametrize("n_rows", [400, 20, 19, 10, 3])  # information form (N + 3 >= 23) and the dense gain branch (N + 3 < 23)
@pytest.mark.parametrize("degenerate", [False, True])
def test_host_filter_with_wheelspeed_rows_matches_the_oracle_filter(oracle_mod, n_rows, degenerate):
    """the product's host filter on the active index set {0..5, 12, 13, 14} (csrc/eskf.cpp) against the oracle's dense restatement of
    esekfom.hpp:1619-1931 (n x 15 Jacobian, two 23 x 23 inverses), both fed the same point-to-plane rows plus the wheel-speed rows"""
    from lsd_amd import lio
    from test_ikfom_vs_ref import _close, _cov, _plane_model, _state

    rng = np.random.default_rng(50 + n_rows)
    for trial in range(4):
        truth = _state(oracle_mod, rng, 0.2)
        s0 = oracle_mod.state_boxplus(truth, np.concatenate([rng.normal(size=6) * [0.2, 0.2, 0.2, 0.02, 0.02, 0.02], np.zeros(17)]))
        P0 = _cov(rng, 1e-3)
        model, calls = _plane_model(rng, n_rows, truth)
        v_ins = rng.normal(size=3) * [3.0, 0.5, 0.0]
        so, Po = oracle_mod.kf_update_ws(s0, P0, 0.001, lio.make_meas_fn(model), v_ins, degenerate, max_iter=4)
        n_o = len(calls)
        del calls[:]
        sp, Pp = lio.eskf_update_ws(s0, P0, 0.001, model, v_ins, degenerate, max_iter=4)
        assert len(calls) == n_o
        tol = 1e-9 if n_rows + 3 >= 23 else 1e-8
        assert _close(sp, so, tol) and _close(Pp, Po, tol * 10), (n_rows, trial, np.abs(sp - so).max(), np.abs(Pp - Po).max())
        # the rows do something: without them the velocity block of the result differs
        s_no, _ = lio.eskf_update(s0, P0, 0.001, model, max_iter=4)
        assert np.abs(s_no[14:17] - sp[14:17]).max() > 1e-9


@pytest.mark.parametrize("n_rows", [400, 10])          # information form, and the dense gain branch (rows + 3 (+ 3) < 23)
@pytest.mark.parametrize("empty_passes", [(1,), (1, 2), (0,), (2, 3)])
@pytest.mark.parametrize("wheel", [True, False])
def test_stale_rows_of_a_pass_without_effective_points_with_wheel_rows(oracle_mod, n_rows, empty_passes, wheel):
    """laserMapping.cpp:991: `ekfom_data_geo = ekfom_data` copies the shared struct, so a pass in which h_share_model_geometric finds no
    effective point (:888-893) re-uses the rows the PREVIOUS pass left there -- its point-to-plane rows and, with wheelspeed_en, the three
    wheel-speed rows that pass appended -- and appends this pass's wheel rows behind them (the weight then counts the stale wheel rows too).
    The product's keeper of those rows (StaleRows in csrc/engine.hip: run_update, the resumed update and this harness share it) against the
    oracle's dense restatement, scripted through the filter-level harnesses: the model reports NO_EFFECTIVE_POINTS in the given passes (pass 0 =
    nothing to fall back on: with wheel rows the three rows alone are the measurement, without them the pass is invalid)."""
    from lsd_amd import lio
    from test_ikfom_vs_ref import _close, _cov, _plane_model, _state

    rng = np.random.default_rng(900 + n_rows + 10 * sum(empty_passes) + wheel)
    for trial in range(3):
        truth = _state(oracle_mod, rng, 0.2)
        s0 = oracle_mod.state_boxplus(truth, np.concatenate([rng.normal(size=6) * [0.2, 0.2, 0.2, 0.02, 0.02, 0.02], np.zeros(17)]))
        P0 = _cov(rng, 1e-3)
        base, calls = _plane_model(rng, n_rows, truth)
        seen = []

        def model(s, converge):
            k = len(seen)
            seen.append(k)
            if k in empty_passes:
                return lio.NO_EFFECTIVE_POINTS
            return base(s, converge)

        v_ins = rng.normal(size=3) * [3.0, 0.5, 0.0]
        if wheel:
            so, Po = oracle_mod.kf_update_ws(s0, P0, 0.001, lio.make_meas_fn(model), v_ins, False, max_iter=4)
        else:
            so, Po = oracle_mod.kf_update(s0, P0, 0.001, lio.make_meas_fn(model), max_iter=4)
        n_o = len(seen)
        del seen[:]
        if wheel:
            sp, Pp = lio.eskf_update_ws(s0, P0, 0.001, model, v_ins, False, max_iter=4)
        else:
            sp, Pp = lio.eskf_update(s0, P0, 0.001, model, max_iter=4)
        assert len(seen) == n_o and n_o >= max(empty_passes) + 1, (n_o, len(seen))
        tol = 1e-9 if n_rows >= 23 else 1e-8
        assert _close(sp, so, tol) and _close(Pp, Po, tol * 10), (n_rows, empty_passes, wheel, trial, np.abs(sp - so).max(), np.abs(Pp - Po).max())
        # the scripted passes matter: the same update with every pass measured ends elsewhere
        del seen[:]
        s_all, _ = (lio.eskf_update_ws(s0, P0, 0.001, base, v_ins, False, max_iter=4) if wheel else lio.eskf_update(s0, P0, 0.001, base, max_iter=4))
        assert np.abs(s_all - sp).max() > 1e-12
