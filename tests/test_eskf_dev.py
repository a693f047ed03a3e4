"""The device-resident form of the iterated ESKF update (csrc/eskf_dev.h: the pass as data-parallel phases, one workgroup per scan
on the GPU) compiled for the HOST and driven by the same Python measurement models as the host filter (lio_eskf_update_cb, which
tests/test_ikfom_vs_ref.py pins to the reference's own IKFoM code) and, when it is built, as the reference filter itself.
Same convergence decisions, states and covariances to rounding.  CPU only; the GPU runs this very source (tests/test_batch_gpu.py, and every -m gpu test that calls lio_engine_update)."""
import numpy as np
import pytest

from test_ikfom_vs_ref import _close, _cov, _plane_model, _state


@pytest.mark.parametrize("n_rows", [400, 60, 23])
def test_device_form_matches_host_filter(oracle_mod, n_rows):
    from lsd_amd import lio

    rng = np.random.default_rng(20 + n_rows)
    for trial in range(8):
        truth = _state(oracle_mod, rng, 0.2)
        s0 = oracle_mod.state_boxplus(truth, np.concatenate([rng.normal(size=6) * [0.2, 0.2, 0.2, 0.02, 0.02, 0.02], np.zeros(17)]))
        P0 = _cov(rng, 1e-3)
        model, calls = _plane_model(rng, n_rows, truth)
        sp, Pp = lio.eskf_update(s0, P0, 0.001, model, max_iter=4)
        n_host = len(calls)
        flags_host = list(calls)
        del calls[:]
        sd, Pd, logs, status = lio.eskf_update_sums(s0, P0, 0.001, model, max_iter=4)
        assert status == 1 and len(calls) == n_host and list(calls) == flags_host and len(logs) == n_host
        assert [l["knn"] for l in logs] == [int(c) for c in flags_host]
        assert np.abs(sd - sp).max() < 1e-12 and np.abs(Pd - Pp).max() < 1e-13 * max(1.0, np.abs(Pp).max()), (np.abs(sd - sp).max(), np.abs(Pd - Pp).max())


def test_device_form_matches_reference_filter(oracle_mod):
    ref_ikfom = pytest.importorskip("ref_ikfom")
    if not ref_ikfom.available():
        pytest.skip("oracle/_ref/libref_ikfom.so not built (needs /root/reference)")
    from lsd_amd import lio

    rng = np.random.default_rng(31)
    for trial in range(6):
        truth = _state(oracle_mod, rng, 0.2)
        s0 = oracle_mod.state_boxplus(truth, np.concatenate([rng.normal(size=6) * [0.2, 0.2, 0.2, 0.02, 0.02, 0.02], np.zeros(17)]))
        P0 = _cov(rng, 1e-3)
        model, calls = _plane_model(rng, 400, truth)
        sr, Pr = ref_ikfom.update(s0, P0, 0.001, lio.make_meas_fn(model), max_iter=4)
        n_ref = len(calls)
        del calls[:]
        sd, Pd, logs, status = lio.eskf_update_sums(s0, P0, 0.001, model, max_iter=4)
        assert status == 1 and len(calls) == n_ref
        assert _close(sd, sr, 1e-9) and _close(Pd, Pr, 1e-8)


def test_invalid_passes_and_stale_measurements(oracle_mod):
    """a pass without effective points: before any valid one it is skipped (esekfom.hpp:1638-1641), after one the rows of the previous
    pass survive in the copied struct and are used again (laserMapping.cpp:991) -- both as the host loop does it"""
    from lsd_amd import lio

    rng = np.random.default_rng(9)
    truth = _state(oracle_mod, rng, 0.2)
    s0 = oracle_mod.state_boxplus(truth, np.concatenate([rng.normal(size=6) * 0.05, np.zeros(17)]))
    P0 = _cov(rng, 1e-3)
    base, calls = _plane_model(rng, 300, truth)
    for pattern in ([1, 0, 1, 0, 1], [0, 0, 1, 1, 1], [1, 1, 0, 0, 0], [0, 0, 0, 0, 0], [1, 0, 0, 0, 0]):
        k = [0]

        def flaky(s, converge):
            k[0] += 1
            return None if not pattern[(k[0] - 1) % 5] else base(s, converge)

        # the host engine's measure lambda re-uses the previous valid measurement; the plain host filter (lio_eskf_update_cb) does not
        # model that copy, so emulate it here exactly as engine.hip's run_update does
        prev = [None]

        def with_stale(s, converge):
            r = flaky(s, converge)
            if r is None:
                return prev[0]
            prev[0] = r
            return r

        k[0] = 0
        sp, Pp = lio.eskf_update(s0, P0, 0.001, with_stale, max_iter=4)
        n_host = k[0]
        k[0] = 0
        sd, Pd, logs, status = lio.eskf_update_sums(s0, P0, 0.001, flaky, max_iter=4)
        assert status == 1 and k[0] == n_host, (pattern, k[0], n_host)
        assert np.abs(sd - sp).max() < 1e-12, (pattern, np.abs(sd - sp).max())
        if any(pattern[:n_host]):
            assert np.abs(Pd - Pp).max() < 1e-13 * max(1.0, np.abs(Pp).max()), pattern


def test_fewer_than_23_rows_hands_over_to_the_host(oracle_mod):
    from lsd_amd import lio

    rng = np.random.default_rng(4)
    truth = _state(oracle_mod, rng, 0.2)
    s0 = oracle_mod.state_boxplus(truth, np.concatenate([rng.normal(size=6) * 0.05, np.zeros(17)]))
    P0 = _cov(rng, 1e-3)
    model, calls = _plane_model(rng, 15, truth)
    sd, Pd, logs, status = lio.eskf_update_sums(s0, P0, 0.001, model, max_iter=4)
    assert status == 2 and len(calls) == 1 and np.array_equal(sd, s0) and np.array_equal(Pd, P0)


def test_degeneracy_projection_in_the_device_form(oracle_mod):
    """rows whose normals leave one direction unobserved: the eigenvalue bound does not decide, the six sums are asked for, the direction
    is projected out of the 6 x 6 normal equations and the state does not move along it"""
    from lsd_amd import lio, synth

    rng = np.random.default_rng(12)
    truth = _state(oracle_mod, rng, 0.05)
    s0 = oracle_mod.state_boxplus(truth, np.concatenate([[0.1, 0.1, 0.1, 0.01, -0.01, 0.02], np.zeros(17)]))
    P0 = np.eye(23) * 1e-2
    n = 600
    pb = rng.uniform(-20, 20, (n, 3))
    nrm = np.zeros((n, 3))
    nrm[:, 2] = 1.0
    nrm[: n // 2, 1] = 1.0
    nrm[: n // 2, 2] = 0.0   # normals along y and z only: nothing observes x
    nrm += rng.normal(0, 0.02, (n, 3)) * [0.0, 1.0, 1.0]
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    Rt, Rl = synth.quat_to_R(truth[3:7]), synth.quat_to_R(truth[7:11])
    pw = (pb @ Rl.T + truth[11:14]) @ Rt.T + truth[0:3]
    d = -np.sum(nrm * pw, 1)

    def model(s, converge):
        R, Rli = synth.quat_to_R(s[3:7]), synth.quat_to_R(s[7:11])
        pi = pb @ Rli.T + s[11:14]
        w = pi @ R.T + s[0:3]
        return np.concatenate([nrm, np.cross(pi, nrm @ R)], 1), -(np.sum(nrm * w, 1) + d)

    sd, Pd, logs, status = lio.eskf_update_sums(s0, P0, 0.001, model, max_iter=4, degenerate_detect=True)
    assert status == 1 and all(l["degenerate"] == 1 for l in logs)
    assert np.abs(logs[-1]["JtJ"][0, :]).max() < 1e-9 * np.abs(logs[-1]["JtJ"]).max()   # x is gone from the normal equations
    assert abs(sd[0] - s0[0]) < 1e-6 and abs(sd[1] - truth[1]) < 1e-3 and abs(sd[2] - truth[2]) < 1e-3
    # without detection the same rows give a (rank-deficient but regularised by the prior) full update
    s2, _, logs2, _ = lio.eskf_update_sums(s0, P0, 0.001, model, max_iter=4, degenerate_detect=False)
    assert all(l["degenerate"] == 0 for l in logs2)
