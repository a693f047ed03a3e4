"""IMU front half on the CPU: the product's host-side forward propagation (csrc/eskf.cpp Eskf::predict, through the
C ABI, no GPU needed) against the oracle, the oracle's propagation against first principles (a numerical Jacobian of the
state transition and an analytic trajectory; the pins against the reference's own filter and translation units are
tests/test_ikfom_vs_ref.py and tests/test_fastlio_vs_ref.py), and the oracle's whole front half tracking a synthetic drive."""
import numpy as np


def _random_state(oracle_mod, rng):
    s = oracle_mod.default_state()
    s[23:26] = [0.3, -0.2, -9.8]
    s[23:26] *= 9.809 / np.linalg.norm(s[23:26])
    return oracle_mod.state_boxplus(s, rng.normal(size=23) * 0.1)


def test_host_predict_bit_identical_to_oracle(oracle_mod):
    from lsd_amd import lio

    rng = np.random.default_rng(0)
    s = _random_state(oracle_mod, rng)
    A = rng.normal(size=(23, 23))
    P = A @ A.T * 0.01
    Q = np.array([1e-4] * 6 + [1e-5] * 6)  # process_noise_cov(), use-ikfom.hpp:36-44 (the oracle's default)
    L = oracle_mod.Lio()
    for _ in range(100):
        dt = rng.uniform(0, 0.01)
        acc = rng.normal(size=3) * 3 + [0, 0, 9.8]
        gyr = rng.normal(size=3) * 0.3
        L.set_state(s)
        L.set_cov(P)
        L.predict(dt, acc, gyr)
        s, P = lio.state_predict(s, P, dt, Q, acc, gyr)
        # the product skips exact zeros of F_x / F_w in the original summation order: every bit survives
        assert np.array_equal(s, L.get_state())
        assert np.array_equal(P, L.get_cov())


def test_predict_covariance_matches_numerical_jacobian(oracle_mod):
    """P' = F P F^T + W Q W^T with F the derivative of x' = x [+] f(x) dt in box coordinates: the reference's F_x1 +
    f_x_final dt is a first-order-in-dt form of it, so for a small dt it must agree with finite differences"""
    from lsd_amd import lio

    rng = np.random.default_rng(1)
    s = _random_state(oracle_mod, rng)
    acc = np.array([0.4, -0.3, 9.7])
    gyr = np.array([0.2, -0.1, 0.3])
    dt, eps = 1e-3, 1e-6
    zeroQ = np.zeros(12)
    base, P1 = lio.state_predict(s, np.eye(23), dt, zeroQ, acc, gyr)
    F = np.zeros((23, 23))
    for j in range(23):
        d = np.zeros(23)
        d[j] = eps
        sp, _ = lio.state_predict(oracle_mod.state_boxplus(s, d), np.eye(23), dt, zeroQ, acc, gyr)
        sm, _ = lio.state_predict(oracle_mod.state_boxplus(s, -d), np.eye(23), dt, zeroQ, acc, gyr)
        F[:, j] = (oracle_mod.state_boxminus(sp, base) - oracle_mod.state_boxminus(sm, base)) / (2 * eps)
    assert np.abs(F @ F.T - P1).max() < 5e-6, np.abs(F @ F.T - P1).max()
    # the noise term: W Q W^T with W = dt * df/dw -- gyro noise enters rot with -1, acc noise enters vel through -R
    Q = np.arange(1, 13) * 1e-3
    _, P2 = lio.state_predict(s, np.zeros((23, 23)), dt, Q, acc, gyr)
    assert np.allclose(np.diag(P2)[3:6], Q[0:3] * dt * dt, rtol=1e-3)
    assert np.allclose(np.diag(P2)[15:18], Q[6:9] * dt * dt, rtol=1e-12)
    assert np.allclose(np.diag(P2)[18:21], Q[9:12] * dt * dt, rtol=1e-12)
    from lsd_amd import synth
    R = synth.quat_to_R(s[3:7])
    assert np.allclose(P2[12:15, 12:15], R @ np.diag(Q[3:6]) @ R.T * dt * dt, rtol=1e-9, atol=1e-18)


def test_predict_integrates_an_analytic_trajectory(oracle_mod):
    from lsd_amd import lio, synth

    tr = synth.Trajectory()
    s = lio.default_state()
    s[23:26] = [0, 0, -9.81 * 9.809 / 9.81]
    P = lio.init_cov()
    Q = np.array([0.1] * 6 + [1e-4] * 6)
    R0, p0 = tr.R(0.0), tr.pos(0.0)
    dt, t = 0.005, 1.0
    g0, a0 = tr.imu(t)
    while t < 3.0 - 1e-9:
        g1, a1 = tr.imu(t + dt)
        s, P = lio.state_predict(s, P, dt, Q, 0.5 * (a0 + a1) * 9.809 / 9.81, 0.5 * (g0 + g1))
        g0, a0 = g1, a1
        t += dt
    Rt, pt = R0.T @ tr.R(t), R0.T @ (tr.pos(t) - p0)
    Re = synth.quat_to_R(s[3:7])
    assert np.arccos(min(1.0, (np.trace(Re.T @ Rt) - 1) / 2)) < 1e-5
    # pos += vel * dt uses the velocity BEFORE the step (explicit Euler, as the reference integrates): a lag of dt / 2 * dv
    vel = lambda u: (tr.pos(u + 1e-4) - tr.pos(u - 1e-4)) / 2e-4
    lag = 0.5 * dt * (R0.T @ (vel(t) - vel(1.0)))
    assert np.linalg.norm(s[0:3] - (pt - lag)) < 2e-3
    assert np.linalg.norm(s[14:17] - R0.T @ vel(t)) < 2e-3
    assert np.all(np.linalg.eigvalsh((P + P.T) / 2) > -1e-12)


def run_drive(front, scene, tr, n_scans, rate=200.0, n_beams=64, n_az=1875, on_scan=None):
    """feed a synthetic drive into an object with imu_enqueue / pcl_enqueue / main / get_state (oracle or product)"""
    from lsd_amd import synth

    imu = synth.imu_stream(tr, 0.0, n_scans * 0.1 + 0.2, rate=rate)
    ii, out = 0, []
    for k in range(n_scans):
        tb = k * 0.1
        pts, st = synth.make_sweep(scene, tr, tb, n_beams=n_beams, n_az=n_az, seed=k, fov_deg=(-24.8, 2.0))
        while ii < len(imu) and imu[ii][0] <= tb + 0.1 + 0.02:
            front.imu_enqueue(*imu[ii])
            ii += 1
        front.pcl_enqueue(pts, st, tb)
        rc = front.main()
        out.append((rc, front.get_state()))
        if on_scan:
            on_scan(k, rc, pts, st)
    return out


def pose_error(tr, s, t):
    from lsd_amd import synth

    R0, p0 = tr.R(0.0), tr.pos(0.0)
    Rt, pt = R0.T @ tr.R(t), R0.T @ (tr.pos(t) - p0)
    Re = synth.quat_to_R(s[3:7])
    return np.linalg.norm(s[0:3] - pt), np.arccos(min(1.0, (np.trace(Re.T @ Rt) - 1) / 2))


class OracleFront:
    def __init__(self, oracle_mod, **cfg):
        self.L = oracle_mod.Lio()
        self.L.frontend_config(**cfg)
        self.imu_enqueue = self.L.imu_enqueue
        self.pcl_enqueue = self.L.pcl_enqueue
        self.main = self.L.frontend_main
        self.get_state = self.L.get_state


def test_oracle_front_half_tracks_a_drive(oracle_mod, scene):
    """sync_packages + IMU_init + forward propagation + undistortion + scan matching + map growth on 24 sweeps of a
    moving 64-beam lidar: the estimate stays within centimetres of the analytic trajectory"""
    from lsd_amd import synth

    tr = synth.Trajectory()
    res = run_drive(OracleFront(oracle_mod, scan_period=0.1), scene, tr, 24)
    rcs = [r for r, _ in res]
    assert rcs[0] == 0 and rcs[1:6] == [4] * 5 and rcs[6] == 1 and all(r == 3 for r in rcs[7:]), rcs
    assert abs(res[5][1][25] + 9.809) < 1e-6  # gravity from the quiet IMU samples, S2 length 9.809
    for k in range(7, 24):
        dp, dr = pose_error(tr, res[k][1], (k + 1) * 0.1)
        assert dp < 0.03 and dr < 5e-3, (k, dp, dr)
    # without motion compensation the same drive is visibly worse once the vehicle moves
    res0 = run_drive(OracleFront(oracle_mod, scan_period=0.1, undistort=False), scene, tr, 24)
    e1 = max(pose_error(tr, res[k][1], (k + 1) * 0.1)[0] for k in range(18, 24))
    e0 = max(pose_error(tr, res0[k][1], (k + 1) * 0.1)[0] for k in range(18, 24))
    assert e0 > 2 * e1, (e0, e1)
