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
    # the voxelised variant (FastVGICP) on the same clouds, as select_registration_method("FAST_VGICP") configures it, per search method
    for name, sm in (("room_small", 1), ("room_small", 7), ("room_fine", 1), ("room_fine", 27)):
        c = gicp_cases.make(name)
        v = ref_gicp.RefVgicp(k=c["k"], resolution=1.0, search_method=sm, num_threads=1)
        v.set_target(c["target"])
        v.set_source(c["source"])
        e, H, b, nc = v.linearize(c["guess"])
        key = f"vgicp/{name}/{sm}/"
        out[key + "err"], out[key + "H"], out[key + "b"], out[key + "n_corr"] = np.float64(e), H, b, np.int32(nc)
        T2 = c["guess"].copy()
        T2[:3, 3] += [0.01, -0.02, 0.005]
        out[key + "err2"] = np.float64(v.compute_error(T2))
        probes = c["target"][:: max(1, len(c["target"]) // 40), :3]
        vox = [v.voxel_at(p) for p in probes]
        out[key + "probe"], out[key + "vox_n"] = probes, np.array([x[0] for x in vox], np.int32)
        out[key + "vox_mean"], out[key + "vox_cov"] = np.array([x[1] for x in vox]), np.array([x[2] for x in vox])
        T, it, conv = v.align(c["guess"].astype(np.float32))
        out[key + "T"], out[key + "iterations"], out[key + "converged"] = T, np.int32(it), np.bool_(conv)
        print("vgicp", name, "search", sm, "n_corr", nc, "err", e, "iterations", it, "converged", conv, "|T - truth| t", np.abs(T[:3, 3] - c["truth"][:3, 3]).max())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gicp.npz"), **out)


if __name__ == "__main__":
    main()
