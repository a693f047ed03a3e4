"""Pins the pieces of the NDT oracle that exist as compilable reference code: fast_gicp's se3_exp (so3.hpp) and the
Eigen calls behind the PLANE covariance regularisation (computeDirect, 3x3 inverse) -- through oracle/_ref."""
import numpy as np
import pytest

import ndt as ondt
import ref as refmod

pytestmark = pytest.mark.skipif(not refmod.available(), reason="oracle/_ref/libref_harness.so not built (needs /root/reference)")


def test_se3_exp_matches_reference():
    rng = np.random.default_rng(0)
    for k in range(200):
        a = rng.normal(size=6) * [0.3, 0.3, 0.3, 1, 1, 1]
        if k % 10 == 0:
            a[:3] *= 1e-7  # small-angle branch
        assert np.allclose(ondt.se3_exp(a), refmod.se3_exp(a), rtol=0, atol=1e-15)


def _covs(rng, n):
    out = []
    for i in range(n):
        # sample covariances of planar-ish patches of 8..40 points at map-scale coordinates, computed in f32 like the map build
        nrm = rng.normal(size=3)
        nrm /= np.linalg.norm(nrm)
        u = np.cross(nrm, rng.normal(size=3))
        u /= np.linalg.norm(u)
        v = np.cross(nrm, u)
        m = rng.integers(8, 40)
        c = rng.uniform(-60, 60, 3) * (0.05 if i % 3 == 0 else 1.0)
        pts = (c + rng.uniform(-0.5, 0.5, (m, 1)) * u + rng.uniform(-0.5, 0.5, (m, 1)) * v + rng.normal(0, 0.02, (m, 1)) * nrm).astype(np.float32)
        s = pts.sum(0, dtype=np.float32)
        S = (pts[:, :, None] * pts[:, None, :]).sum(0, dtype=np.float32)
        mean = s / np.float32(m)
        out.append(((S - mean[:, None] * s[None, :]) / np.float32(m)).astype(np.float32))
    return out


def test_direct_eigen_solver_matches_eigen():
    rng = np.random.default_rng(1)
    worst = 0.0
    for C in _covs(rng, 500):
        w_o, V_o = ondt.eig3_direct(C)
        w_r, V_r = refmod.eig3_direct(C)
        assert np.allclose(w_o, w_r, rtol=1e-4, atol=1e-6 * max(1.0, np.abs(C).max()))
        # eigenvectors up to sign (the normal direction = column 0 is the one that matters downstream)
        d = abs(float(V_o[:, 0] @ V_r[:, 0]))
        worst = max(worst, 1 - d)
    assert worst < 1e-4


def test_plane_regularisation_matches_eigen():
    rng = np.random.default_rng(2)
    for C in _covs(rng, 500):
        R_o, I_o = ondt.regularize_plane(C)
        R_r, I_r = refmod.regularize_plane(C)
        assert np.allclose(R_o, R_r, rtol=0, atol=2e-3), (R_o, R_r)      # entries in [0, 1]
        assert np.allclose(I_o, I_r, rtol=2e-2, atol=2.0)                  # entries up to 1000: f32 inverse of a 1e3-conditioned matrix
        # and both are what the construction promises: eigenvalues (1e-3, 1, 1)
        ev = np.linalg.eigvalsh((R_o.astype(np.float64) + R_o.T) / 2)
        assert np.allclose(ev, [1e-3, 1, 1], atol=5e-3)
