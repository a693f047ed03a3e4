"""Known-answer tests of the CPU oracle written from first principles (the reference ships no tests or vectors
for this path: SURVEY.md section 4).  Each test states the property it derives its expected value from."""
import numpy as np
from scipy.spatial.transform import Rotation as Rot


def test_esti_plane_recovers_an_exact_plane(oracle_mod):
    # five points of the plane n.x + d = 0 with |n| = 1  ->  pabcd = (n, d) up to f32 rounding, all residuals ~0
    rng = np.random.default_rng(0)
    for _ in range(50):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        d = rng.uniform(2, 30)  # esti_plane solves A x = -1: the plane must not pass through the origin
        u = np.cross(n, [1, 0, 0.3])
        u /= np.linalg.norm(u)
        v = np.cross(n, u)
        ab = rng.uniform(-0.5, 0.5, (5, 2))
        pts = -d * n + ab[:, :1] * u + ab[:, 1:] * v
        ok, p = oracle_mod.esti_plane(np.concatenate([pts, np.zeros((5, 1))], 1))
        assert ok
        assert np.allclose(p[:3], n, atol=2e-4) and abs(p[3] - d) < 2e-3 * d
        assert abs(np.linalg.norm(p[:3]) - 1) < 1e-6


def test_esti_plane_rejects_a_rough_patch(oracle_mod):
    pts = np.array([[10, 0, 0], [10.5, 0, 0.0], [10, 0.5, 0], [10.5, 0.5, 0], [10.25, 0.25, 0.6]], np.float32)
    ok, _ = oracle_mod.esti_plane(np.concatenate([pts, np.zeros((5, 1), np.float32)], 1))
    assert not ok  # one point 0.6 m off a 0.5 m patch: some residual exceeds 0.1


def test_voxelgrid_hand_case(oracle_mod):
    # leaf 1: points (0.2,0.2,0.2),(0.8,0.4,0.6) share voxel (0,0,0); (-0.5,0.1,0.1) is voxel (-1,0,0) and comes
    # FIRST (lower linear index); (1.5,0.5,0.5) is voxel (1,0,0); (0.5,1.5,0.5) is voxel (0,1,0) (y stride = 3)
    pts = np.array([[0.2, 0.2, 0.2, 10], [1.5, 0.5, 0.5, 20], [0.8, 0.4, 0.6, 30], [-0.5, 0.1, 0.1, 40], [0.5, 1.5, 0.5, 50]], np.float32)
    ds = oracle_mod.voxel_downsample(pts, 1.0)
    want = np.array([[-0.5, 0.1, 0.1, 40], [0.5, 0.3, 0.4, 20], [1.5, 0.5, 0.5, 20], [0.5, 1.5, 0.5, 50]], np.float32)
    assert ds.shape == (4, 4) and np.allclose(ds, want, atol=1e-6)


def test_voxelgrid_overflow_guard_returns_input(oracle_mod):
    pts = np.array([[0, 0, 0, 1], [1e6, 1e6, 1e6, 2], [5, 5, 5, 3]], np.float32)
    assert np.array_equal(oracle_mod.voxel_downsample(pts, 0.01), pts)


def test_ivox_key_rounding_and_stencil_membership(oracle_mod):
    # Pos2Grid rounds half away from zero: 0.25 -> key 1 (voxel centred on 0.5), -0.25 -> key -1; 0.24 -> key 0
    iv = oracle_mod.IVox(res=0.5, stencil=1)
    iv.add(np.array([[0.25, 0, 0, 0], [-0.25, 0, 0, 0], [0.24, 0, 0, 0]], np.float32))
    assert iv.num_voxels == 3
    # NEARBY6 sees the face neighbour but not the edge neighbour; NEARBY18 sees both; "NEARBY74" reaches two cells in x/y
    m = np.array([[0.5, 0, 0, 1], [0.5, 0.5, 0, 2], [1.0, 0, 0, 3], [0, 0, 1.0, 4]], np.float32)
    q = np.zeros((1, 4), np.float32)
    for st, want in ((1, 0), (7, 1), (19, 2), (27, 2), (75, 3)):
        iv = oracle_mod.IVox(res=0.5, stencil=st)
        iv.add(m)
        _, cnt, _ = iv.knn(q)
        assert cnt[0] == want, (st, cnt)


def test_ivox_range_limit(oracle_mod):
    # d^2 < 5.0 (strict): a point at distance sqrt(5) is excluded, one just inside is kept (needs the 75 stencil)
    iv = oracle_mod.IVox(res=2.0, stencil=75)
    iv.add(np.array([[2.2, 0, 0, 0], [np.sqrt(5.0) + 1e-3, 0, 0, 0]], np.float32))
    _, cnt, _ = iv.knn(np.zeros((1, 4), np.float32))
    assert cnt[0] == 1


def test_so3_boxplus_boxminus(oracle_mod):
    rng = np.random.default_rng(1)
    s = oracle_mod.default_state()
    for _ in range(20):
        d = np.zeros(23)
        d[3:6] = rng.normal(size=3) * 0.3
        s2 = oracle_mod.state_boxplus(s, d)
        R = Rot.from_quat(s2[3:7]).as_matrix()
        assert np.allclose(R, Rot.from_rotvec(d[3:6]).as_matrix(), atol=1e-12)  # q (+) v = q * exp(v)
        assert np.allclose(oracle_mod.state_boxminus(s2, s), d, atol=1e-12)


def test_A_matrix_is_the_so3_right_jacobian_series(oracle_mod):
    # A(v) = I + (1-cos|v|)/|v|^2 [v]x + (1 - sin|v|/|v|)/|v|^2 [v]x^2   (mtkmath.hpp:235-247)
    rng = np.random.default_rng(2)
    for _ in range(10):
        v = rng.normal(size=3) * 0.5
        th = np.linalg.norm(v)
        K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        want = np.eye(3) + (1 - np.cos(th)) / th**2 * K + (1 - np.sin(th) / th) / th**2 * K @ K
        assert np.allclose(oracle_mod.A_matrix(v), want, atol=1e-13)
    assert np.array_equal(oracle_mod.A_matrix(np.zeros(3)), np.eye(3))


def test_s2_boxplus_keeps_the_length_and_boxminus_inverts_it(oracle_mod):
    rng = np.random.default_rng(3)
    s = oracle_mod.default_state()
    s[23:26] = np.array([0.3, -0.2, -9.8])
    s[23:26] *= oracle_mod.G_LEN / np.linalg.norm(s[23:26])
    for _ in range(10):
        d = np.zeros(23)
        d[21:23] = rng.normal(size=2) * 0.05
        s2 = oracle_mod.state_boxplus(s, d)
        assert abs(np.linalg.norm(s2[23:26]) - oracle_mod.G_LEN) < 1e-12
        assert np.allclose(oracle_mod.state_boxminus(s2, s)[21:23], d[21:23], atol=1e-9)


def test_normal_equations_match_finite_differences(oracle_mod, small_world):
    """JtJ/Jtr are the Gauss-Newton terms of 0.5 * sum r_i(x)^2 over the selected points: the gradient -J^T h = J^T r
    must match a central finite difference of that cost with the correspondences frozen"""
    from lsd_amd import synth

    ds = oracle_mod.voxel_downsample(small_world["raw"], 0.5)[::8]
    state = synth.state_from_pose(small_world["guess_pos"], small_world["guess_q"])
    o = oracle_mod.Lio(stencil=19, capacity=1 << 40, threads=4)
    o.map_add(small_world["map"][::3])
    o.set_state(state)
    o.set_flags(ekf_inited=True, first_scan=False)
    o.set_ds(ds)
    lin = o.linearize(converge=True)
    sel = lin["selected"].astype(bool)
    nv = lin["normvec"][sel].astype(np.float64)
    pb = ds[sel, :3].astype(np.float64)

    def cost(dpos, drot):
        R = Rot.from_quat(state[3:7]).as_matrix() @ Rot.from_rotvec(drot).as_matrix()
        pw = pb @ R.T + state[:3] + dpos
        r = (nv[:, :3] * pw).sum(1) + (nv[:, 3] - (nv[:, :3] * (pb @ Rot.from_quat(state[3:7]).as_matrix().T + state[:3])).sum(1))
        return 0.5 * (r**2).sum()

    g = np.zeros(6)
    eps = 1e-6
    for k in range(6):
        d = np.zeros(6)
        d[k] = eps
        g[k] = (cost(d[:3], d[3:]) - cost(-d[:3], -d[3:])) / (2 * eps)
    assert np.allclose(-lin["Jtr"], g, rtol=2e-4, atol=1e-5)
    assert np.all(np.linalg.eigvalsh(lin["JtJ"]) > 0)


def test_update_recovers_a_known_perturbation(oracle_mod, small_world):
    from lsd_amd import synth

    ds = oracle_mod.voxel_downsample(small_world["raw"], 0.5)
    o = oracle_mod.Lio(stencil=19, capacity=1 << 40, threads=8)
    o.map_add(small_world["map"])
    o.set_state(synth.state_from_pose(small_world["guess_pos"], small_world["guess_q"]))
    o.set_cov(oracle_mod.init_cov())
    o.set_flags(ekf_inited=True, first_scan=False)
    o.set_ds(ds)
    logs = o.update()
    s = o.get_state()
    assert 2 <= len(logs) <= 5 and logs[0]["knn"] == 1
    assert np.linalg.norm(s[:3] - small_world["true_pos"]) < 0.03
    assert synth.quat_angle(s[3:7], small_world["true_q"]) < 2e-3
    P = o.get_cov()
    assert np.allclose(P, P.T, atol=1e-9) and np.all(np.diag(P)[:6] < 1e-3)  # the pose block collapsed
