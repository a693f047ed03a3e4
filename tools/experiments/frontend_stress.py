"""random drives through the lio_fastlio_* entry points against the oracle's front half (sync_packages, IMU_init, forward propagation, per-point motion
compensation, registration, map_incremental): random speeds and oscillations, IMU rates 100 ... 400 Hz, dropped IMU packets, scans without an IMU sample,
point_filter_num, max_point_num, sweep sizes -- return codes equal sweep by sweep and the same points survive the blind filter (64 drives x 20 sweeps: always);
both sides run free, so last-bit differences of the poses grow along a drive through the f32 map and the compensation: 1e-7 m on the first registered sweeps,
1e-4 ... 2e-4 m in the undistorted clouds and up to 1.6e-3 m in a pose by sweeps 16 - 19 (the per-sweep figures from equal state and map are bench.py's);
flagged here: 1e-2 m"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle
from lsd_amd import capi, lio, synth


def main(n_cfg=16, seed0=0):
    bad = sweeps = 0
    for c in range(n_cfg):
        rng = np.random.default_rng(seed0 * 6007 + c)
        scene = synth.Scene(half=float(rng.choice([40.0, 80.0])), n_boxes=int(rng.choice([10, 30])), seed=int(rng.integers(1, 1000)))
        tr = synth.Trajectory(speed=float(rng.uniform(0.0, 15.0)), tau=float(rng.uniform(0.5, 3.0)), sway=float(rng.uniform(0, 3.0)), yaw_amp=float(rng.uniform(0, 1.2)),
                              pitch_amp=float(rng.uniform(0, 0.08)), roll_amp=float(rng.uniform(0, 0.08)), heading=float(rng.uniform(-3, 3)))
        rate = float(rng.choice([100.0, 200.0, 400.0]))
        pfn = int(rng.choice([1, 1, 2, 3]))
        mpn = int(rng.choice([-1, -1, 30000]))
        cfg = dict(scan_period=0.1, filter_num=pfn, max_point_num=mpn, undistort=bool(rng.random() < 0.85))
        n_az = int(rng.choice([200, 600, 1875]))
        drop = float(rng.choice([0.0, 0.0, 0.02, 0.1]))
        hole = int(rng.integers(9, 20)) if rng.random() < 0.4 else -1  # one sweep during which no IMU packet arrives
        e = lio.Engine(max_points=4_000_000, max_voxels=1 << 20, max_raw=1 << 18, max_ds=100000)
        e.fastlio_init(**cfg)
        L = oracle.Lio()
        L.frontend_config(**cfg)
        n = 20
        imu = synth.imu_stream(tr, 0.0, n * 0.1 + 0.2, rate=rate)
        ii, ok = 0, True
        for k in range(n):
            tb = k * 0.1
            pts, st = synth.make_sweep(scene, tr, tb, n_beams=64, n_az=n_az, seed=int(rng.integers(1, 1 << 30)), fov_deg=(-24.8, 2.0))
            while ii < len(imu) and imu[ii][0] <= tb + 0.1 + 0.02:
                if not (rng.random() < drop or (k == hole and imu[ii][0] > tb)):
                    e.fastlio_imu_enqueue(*imu[ii])
                    L.imu_enqueue(*imu[ii])
                ii += 1
            e.fastlio_pcl_enqueue(pts, st, tb)
            L.pcl_enqueue(pts, st, tb)
            ra, rb = e.fastlio_main(), L.frontend_main()
            sweeps += 1
            sa, sb = e.get_state(), L.get_state()
            dp, dr = float(np.linalg.norm(sa[:3] - sb[:3])), float(synth.quat_angle(sa[3:7], sb[3:7]))
            msg = None
            if ra != rb:
                msg = ("return codes", ra, rb)
            elif ra == capi.MAIN_UPDATED:
                a, b = e.undistorted(), L.get_undistorted()
                keep = np.isfinite(a[:, 0])
                if keep.sum() != len(b):
                    msg = ("points after the blind filter", int(keep.sum()), len(b))
                elif np.abs(a[keep][:, :3] - b[:, :3]).max() > 1e-2:
                    msg = ("undistorted clouds", float(np.abs(a[keep][:, :3] - b[:, :3]).max()))
                elif dp > 1e-2 or dr > 1e-3:  # (free-running on both sides: a last-bit difference grows along the drive; the per-sweep figures are bench.py's teacher-forced legs)
                    msg = ("poses", dp, dr)
            if msg:
                bad += 1
                ok = False
                print("MISMATCH cfg", c, "sweep", k, dict(rate=rate, cfg=cfg, n_az=n_az, drop=drop, hole=hole, speed=round(tr.speed, 1)), msg)
                break
        e.close()
    print("configurations", n_cfg, "sweeps compared", sweeps, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
