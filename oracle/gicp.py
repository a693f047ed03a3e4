"""CPU restatement (numpy) of the reference's fine matcher fast_gicp::FastGICP<PointXYZI, PointXYZI> -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/slam/thirdparty/fast_gicp/include/fast_gicp/gicp/impl/fast_gicp_impl.hpp:
  covariances()   calculate_covariances  :244-303 (k nearest neighbours, covariance / k, PLANE regularisation = SVD with the singular
                                                   values replaced by (1, 1, 1e-3))
  correspondences()  update_correspondences  :118-157 (nearest target point of trans_f * a, f32; Mahalanobis (C_B + T C_A T^T)^-1)
  linearize()     linearize / compute_error  :159-242
  align()         LsqRegistration::computeTransformation / step_lm / is_converged  lsq_registration_impl.hpp:71-208, se3_exp of so3.hpp
  Vgicp           fast_gicp::FastVGICP: fast_vgicp_voxel.hpp:95-110,129-167 (ADDITIVE Gaussian voxels, key floor(x / res - 0.5) in f64),
                  fast_vgicp_impl.hpp:72-204 (voxel correspondences DIRECT1 / 7 / 27, weight sqrt(points in the voxel))
Pinned against the reference itself (oracle/_ref/libref_gicp.so, oracle/ref_gicp.cpp) by tests/test_gicp.py and through the vectors that
harness wrote into tests/golden/gicp_*.npz (tools/make_gicp_golden.py).  Brute-force neighbour search: small clouds only."""
import numpy as np


def _d2_f32(q, pts):
    """((dx*dx + dy*dy) + dz*dz) in f32, the order of FLANN's L2_Simple and of the device kernel"""
    e = pts[None, :, :3].astype(np.float32) - q[:, None, :3].astype(np.float32)
    return (e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1]) + e[..., 2] * e[..., 2]


def knn(cloud, k):
    """indices [n, k] of the k nearest neighbours of every point within its own cloud, ascending (d2, index)"""
    d2 = _d2_f32(cloud, cloud)
    order = np.lexsort((np.broadcast_to(np.arange(len(cloud)), d2.shape), d2), axis=1)
    return order[:, :k], np.take_along_axis(d2, order[:, :k + 1], axis=1)


def covariances(cloud, k=20):
    cloud = np.asarray(cloud, np.float32).reshape(-1, 4)
    idx, _ = knn(cloud, k)
    nb = cloud[idx][..., :3].astype(np.float64)           # [n, k, 3]
    nb = nb - nb.mean(axis=1, keepdims=True)
    cov = np.einsum("nka,nkb->nab", nb, nb) / k
    out = np.zeros_like(cov)
    for i in range(len(cov)):
        U, _, Vt = np.linalg.svd(cov[i])
        out[i] = U @ np.diag([1.0, 1.0, 1e-3]) @ Vt
    return out


def transform_f(T, pts):
    """trans.cast<float>() * [x y z 1]: Eigen folds a row's four products pairwise (checked against the reference build)"""
    Tf = np.asarray(T, np.float64).astype(np.float32)
    p = np.asarray(pts, np.float32)
    out = np.empty((len(p), 3), np.float32)
    for r in range(3):
        out[:, r] = (Tf[r, 0] * p[:, 0] + Tf[r, 1] * p[:, 1]) + (Tf[r, 2] * p[:, 2] + Tf[r, 3])
    return out


def correspondences(src, tgt, cov_src, cov_tgt, T, max_corr_dist):
    T = np.asarray(T, np.float64)
    q = transform_f(T, src)
    d2 = _d2_f32(q, tgt)
    j = np.lexsort((np.broadcast_to(np.arange(len(tgt)), d2.shape), d2), axis=1)[:, 0]
    sq = d2[np.arange(len(src)), j]
    thr = np.float32(max_corr_dist) * np.float32(max_corr_dist) if max_corr_dist < 1e18 else np.float32(np.inf)
    corr = np.where(sq < thr, j, -1).astype(np.int32)
    maha = np.zeros((len(src), 3, 3))
    R = T[:3, :3]
    for i in np.nonzero(corr >= 0)[0]:
        maha[i] = np.linalg.inv(cov_tgt[corr[i]] + R @ cov_src[i] @ R.T)
    return corr, sq, maha


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def cost(src, tgt, corr, maha, T, derivatives=True):
    T = np.asarray(T, np.float64)
    H, b, err = np.zeros((6, 6)), np.zeros(6), 0.0
    for i in np.nonzero(corr >= 0)[0]:
        ta = T[:3, :3] @ src[i, :3].astype(np.float64) + T[:3, 3]
        e = tgt[corr[i], :3].astype(np.float64) - ta
        err += e @ maha[i] @ e
        if derivatives:
            J = np.hstack([_skew(ta), -np.eye(3)])
            H += J.T @ maha[i] @ J
            b += J.T @ maha[i] @ e
    return err, H, b


def se3_exp(a):
    """so3.hpp se3_exp: a = (omega, v)"""
    w, v = np.asarray(a[:3], np.float64), np.asarray(a[3:], np.float64)
    th = np.sqrt(w @ w)
    O = _skew(w)
    T = np.eye(4)
    if th < 1e-10:
        Rm, V = np.eye(3), np.eye(3)
    else:
        O2 = O @ O
        Rm = np.eye(3) + np.sin(th) / th * O + (1 - np.cos(th)) / th**2 * O2
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * O + (th - np.sin(th)) / th**3 * O2
    T[:3, :3], T[:3, 3] = Rm, V @ v
    return T


class Gicp:
    def __init__(self, k=20, max_corr_dist=2.0, transformation_epsilon=0.01, rotation_epsilon=1e-2, max_iterations=64):
        self.k, self.maxd, self.teps, self.reps, self.max_iter = k, max_corr_dist, transformation_epsilon, rotation_epsilon, max_iterations
        self.lm_max_iterations, self.lm_init_lambda_factor = 10, 1e-9

    def set_target(self, xyzi):
        self.tgt = np.asarray(xyzi, np.float32).reshape(-1, 4)
        self.cov_tgt = covariances(self.tgt, self.k)
        return self.cov_tgt

    def set_source(self, xyzi):
        self.src = np.asarray(xyzi, np.float32).reshape(-1, 4)
        self.cov_src = covariances(self.src, self.k)
        return self.cov_src

    def linearize(self, T):
        self.corr, self.sq, self.maha = correspondences(self.src, self.tgt, self.cov_src, self.cov_tgt, T, self.maxd)
        return cost(self.src, self.tgt, self.corr, self.maha, T, True)

    def compute_error(self, T):
        return cost(self.src, self.tgt, self.corr, self.maha, T, False)[0]

    def _converged(self, delta, scale=1.0):
        c = np.clip((np.trace(delta[:3, :3]) - 1) / 2, -1, 1)
        Rdeg = np.degrees(np.arccos(c))
        return max(Rdeg / (self.reps * scale), np.abs(delta[:3, 3]).max() / (self.teps * scale)) < 1

    def align(self, guess):
        x0 = np.asarray(guess, np.float32).astype(np.float64)
        lam, converged, it = -1.0, False, 0
        for i in range(self.max_iter):
            if converged:
                break
            it = i
            y0, H, b = self.linearize(x0)
            if lam < 0:
                lam = self.lm_init_lambda_factor * np.abs(np.diag(H)).max()
            nu, ok = 2.0, False
            for _ in range(self.lm_max_iterations):
                d = np.linalg.solve(H + lam * np.eye(6), -b)
                delta = se3_exp(d)
                xi = delta @ x0
                yi = self.compute_error(xi)
                rho = (y0 - yi) / (d @ (lam * d - b))
                if rho < 0:
                    if self._converged(delta, 10.0):
                        ok = True
                        break
                    lam, nu = nu * lam, 2 * nu
                    continue
                x0 = xi
                lam = lam * max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
                ok = True
                break
            if not ok:
                break
            converged = self._converged(delta)
        return x0.astype(np.float32), it, converged


OFFSETS = {1: [(0, 0, 0)], 7: [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)],
           27: [(i - 1, j - 1, k - 1) for i in range(3) for j in range(3) for k in range(3)]}


class Vgicp(Gicp):
    """fast_gicp::FastVGICP<PointXYZI, PointXYZI> (select_registration_method("FAST_VGICP"): resolution 1.0, epsilons 0.1 / 0.1)"""

    def __init__(self, k=20, resolution=1.0, search_method=1, transformation_epsilon=0.1, rotation_epsilon=0.1, max_iterations=64):
        super().__init__(k=k, max_corr_dist=np.inf, transformation_epsilon=transformation_epsilon, rotation_epsilon=rotation_epsilon, max_iterations=max_iterations)
        self.res, self.offsets, self.voxels = float(resolution), OFFSETS[search_method], None

    def set_target(self, xyzi):
        self.voxels = None
        return super().set_target(xyzi)

    def _coord(self, x):
        return tuple(np.floor(np.asarray(x, np.float64) / self.res - 0.5).astype(np.int64))

    def _build(self):
        acc = {}
        for i in range(len(self.tgt)):  # create_voxelmap: append in input order, then finalize
            c = self._coord(self.tgt[i, :3].astype(np.float64))
            v = acc.setdefault(c, [0, np.zeros(3), np.zeros((3, 3))])
            v[0] += 1
            v[1] = v[1] + self.tgt[i, :3].astype(np.float64)
            v[2] = v[2] + self.cov_tgt[i]
        self.voxels = {c: (n, m / n, C / n) for c, (n, m, C) in acc.items()}

    def voxel_at(self, p):
        if self.voxels is None:
            self._build()
        v = self.voxels.get(self._coord(np.asarray(p, np.float32).astype(np.float64)))
        return (0, None, None) if v is None else v

    def linearize(self, T, derivatives=True, update=True):
        if self.voxels is None:
            self._build()
        T = np.asarray(T, np.float64)
        R = T[:3, :3]
        if update:
            self.vcorr = []
            for i in range(len(self.src)):
                ta = R @ self.src[i, :3].astype(np.float64) + T[:3, 3]
                c = self._coord(ta)
                for o in self.offsets:
                    v = self.voxels.get((c[0] + o[0], c[1] + o[1], c[2] + o[2]))
                    if v is not None:
                        self.vcorr.append((i, v, np.linalg.inv(v[2] + R @ self.cov_src[i] @ R.T)))
        H, b, err = np.zeros((6, 6)), np.zeros(6), 0.0
        for i, v, M in self.vcorr:
            ta = R @ self.src[i, :3].astype(np.float64) + T[:3, 3]
            e = v[1] - ta
            w = np.sqrt(v[0])
            err += w * (e @ M @ e)
            if derivatives:
                J = np.hstack([_skew(ta), -np.eye(3)])
                H += w * (J.T @ M @ J)
                b += w * (J.T @ M @ e)
        return err, H, b

    def compute_error(self, T):
        return self.linearize(T, derivatives=False, update=False)[0]
