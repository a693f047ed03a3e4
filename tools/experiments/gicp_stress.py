"""random scan pairs through lio_gicp_* with the reference's own fast_gicp::FastGICP (oracle/_ref/libref_gicp.so) run beside it on the same input: point
covariances, correspondences, cost / H / b, whole alignments -- the checks of tests/test_gicp_gpu.py::_check_against on scenes, thinnings, neighbour counts,
correspondence ranges and guesses drawn at random"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import gicp_cases
import ref_gicp
from lsd_amd import synth
from test_gicp_gpu import _check_against


def main(n_cfg=10, seed0=0):
    bad = 0
    for c_ in range(n_cfg):
        rng = np.random.default_rng(seed0 * 31337 + c_)
        sc = synth.Scene(half=float(rng.choice([30.0, 60.0])), n_boxes=int(rng.choice([8, 30])), seed=int(rng.integers(1, 1000)))
        pa = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), 1.8])
        qa = synth.quat_from_rotvec([0, 0, rng.uniform(-3, 3)])
        pb = pa + np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), 0.0])
        qb = synth.quat_from_rotvec([rng.normal(0, 0.02), rng.normal(0, 0.02), rng.uniform(-3, 3)])
        n_az = int(rng.choice([300, 900, 1875]))
        ra, _ = synth.make_scan(sc, pa, qa, seed=int(rng.integers(1, 1 << 30)), max_range=80.0, n_az=n_az)
        rb, _ = synth.make_scan(sc, pb, qb, seed=int(rng.integers(1, 1 << 30)), max_range=80.0, n_az=n_az)
        thin = float(rng.choice([0.2, 0.3, 0.6]))
        k = int(rng.choice([10, 20]))
        maxd = float(rng.choice([1.0, 2.0, 5.0]))
        c = dict(target=gicp_cases._thin(ra[:, :4].astype(np.float32), thin), source=gicp_cases._thin(rb[:, :4].astype(np.float32), thin), k=k, max_corr_dist=maxd)
        if min(len(c["target"]), len(c["source"])) < 200:
            continue
        truth = np.linalg.inv(gicp_cases._pose(pa, qa)) @ gicp_cases._pose(pb, qb)
        c["guess"] = truth @ gicp_cases._pose(rng.normal(0, 0.1, 3), synth.quat_from_rotvec(rng.normal(0, 0.01, 3)))
        h = ref_gicp.RefGicp(k=k, max_corr_dist=maxd, num_threads=4)
        ref = dict(cov_tgt=h.set_target(c["target"]), cov_src=h.set_source(c["source"]))
        e, H, b, corr, _, _ = h.linearize(c["guess"])
        T2 = c["guess"].copy()
        T2[:3, 3] += [0.01, -0.02, 0.005]
        ref.update(err=e, H=H, b=b, corr=corr, err2=h.compute_error(T2))
        T, it, conv = h.align(c["guess"].astype(np.float32))
        ref.update(T=T, iterations=it, converged=conv)
        try:
            _check_against(c, ref, float(rng.choice([1.0, 0.6])))
        except AssertionError as ex:
            bad += 1
            print("MISMATCH cfg", c_, dict(n_t=len(c["target"]), n_s=len(c["source"]), k=k, maxd=maxd, thin=thin), repr(ex)[:300])
    print("configurations", n_cfg, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
