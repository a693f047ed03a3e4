"""Per-stage timing of lio_fastlio_main (IMU front half + scan matching + map growth) on a synthetic drive.
Usage: python tools/bench_frontend.py [--scans 40] [--rate 200]   (GPU box)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-slam-detection_amd", "python"))
from lsd_amd import capi, lio, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=40)
    ap.add_argument("--rate", type=float, default=200.0)
    ap.add_argument("--timing", type=int, default=1)
    a = ap.parse_args()
    scene, tr = synth.Scene(seed=1), synth.Trajectory()
    e = lio.Engine(max_points=8_000_000, max_voxels=1 << 21, max_raw=1 << 18, max_ds=100000)
    e.fastlio_init(scan_period=0.1)
    e.enable_timing(bool(a.timing))
    imu = synth.imu_stream(tr, 0.0, a.scans * 0.1 + 0.2, rate=a.rate)
    sweeps = [synth.make_sweep(scene, tr, k * 0.1, seed=k, fov_deg=(-24.8, 2.0)) for k in range(a.scans)]
    ii, rows = 0, []
    for k, (pts, st) in enumerate(sweeps):
        tb = k * 0.1
        while ii < len(imu) and imu[ii][0] <= tb + 0.12:
            e.fastlio_imu_enqueue(*imu[ii])
            ii += 1
        t0 = time.perf_counter()
        e.fastlio_pcl_enqueue(pts, st, tb)
        t1 = time.perf_counter()
        rc = e.fastlio_main()
        t2 = time.perf_counter()
        if rc == capi.MAIN_UPDATED:
            tm = e.timings()
            rows.append(dict(n_raw=len(pts), enqueue_us=(t1 - t0) * 1e6, main_us=(t2 - t1) * 1e6, imu_host_us=tm["imu_host_us"], undistort_us=tm["undistort_us"],
                             downsample_us=tm["downsample_us"], knn_us=tm["knn_us"], linearize_us=tm["linearize_us"], insert_us=tm["insert_us"],
                             host_solve_us=tm["host_solve_us"], n_ds=tm["n_ds"], n_pass=tm["n_pass"]))
    rows = rows[3:]  # warm-up
    med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    s = e.get_state()
    R0, p0 = tr.R(0.0), tr.pos(0.0)
    te = a.scans * 0.1
    med["final_pos_err_m"] = float(np.linalg.norm(s[0:3] - R0.T @ (tr.pos(te) - p0)))
    med["scans"] = len(rows)
    med["timing_events"] = a.timing
    print(json.dumps(med))


if __name__ == "__main__":
    main()
