"""Writes tests/golden/gicp.npz from the reference itself: fast_gicp::FastGICP compiled from /root/reference (oracle/_ref/libref_gicp.so,
`make -C oracle ref`).  Runs only where /root/reference is mounted; the vectors travel, the reference does not.

  python tools/make_gicp_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lidar-slam-detection_amd", "python"))
import ref_gicp  # noqa: E402
import gicp_cases  # noqa: E402


def main():
    out = {}
    for name in gicp_cases.CASES:
        c = gicp_cases.make(name)
        g = ref_gicp.RefGicp(k=c["k"], max_corr_dist=c["max_corr_dist"], num_threads=1)
        out[name + "/cov_tgt"] = g.set_target(c["target"])
        out[name + "/cov_src"] = g.set_source(c["source"])
        e, H, b, corr, sq, maha = g.linearize(c["guess"])
        out[name + "/err"], out[name + "/H"], out[name + "/b"] = np.float64(e), H, b
        out[name + "/corr"], out[name + "/sq"], out[name + "/maha"] = corr, sq, maha
        T2 = c["guess"].copy()
        T2[:3, 3] += [0.01, -0.02, 0.005]
        out[name + "/err2"] = np.float64(g.compute_error(T2))
        T, it, conv = g.align(c["guess"].astype(np.float32))
        out[name + "/T"], out[name + "/iterations"], out[name + "/converged"] = T, np.int32(it), np.bool_(conv)
        print(name, "n_src", len(c["source"]), "n_corr", int((corr >= 0).sum()), "err", e, "iterations", it, "converged", conv)
        print("   |T - truth| t", np.abs(T[:3, 3] - c["truth"][:3, 3]).max())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gicp.npz"), **out)


if __name__ == "__main__":
    main()
