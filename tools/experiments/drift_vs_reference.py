"""Config 3's lawnmower drive through the HIP front half AND the reference's own FastLIO build (oracle/_ref/libref_fastlio_release.so) on the
same sweeps: position error of each against the generating trajectory, and their distance from each other, every 50 sweeps."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "lidar-slam-detection_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from lsd_amd import capi, lio, synth, synth_gpu
import ref_fastlio

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
speed = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
dev = torch.device("cuda", 0)
scene = synth.Scene(half=500.0, n_boxes=1500, seed=3, box_size=(4.0, 30.0), keep_clear=8.0)
tr = synth_gpu.Lawnmower(speed=speed)
sweeper = synth_gpu.Sweeper(scene, tr, dev, fov_deg=(-24.8, 2.0), max_range=100.0, seed=1000)
imu_t, imu_g, imu_a = synth_gpu.imu_stream(tr, 0.0, n * 0.1 + 0.3, rate=100.0, seed=1000, gyr_sigma=1e-3, acc_sigma=1e-2)
e = lio.Engine(resolution=0.5, stencil=75, max_points=13_000_000, max_voxels=1 << 23, max_raw=1 << 18, max_ds=100000, device=0)
e.map.set_lru((1 << 23) - 100_000, 1e9)
e.fastlio_init(scan_period=0.1)
ref_fastlio.use_release_build()
R = ref_fastlio.RefFastLio(scan_period=0.1)
R.set_logging(False)
R0, p0 = tr.R(0.0), tr.pos(0.0)
ii = jj = 0
print("sweep  driven_m   gpu_err_m  ref_err_m  gpu_vs_ref_m   gpu_err_track(x along, y left, z up)   yaw_err_gpu_deg yaw_err_ref_deg")
for k in range(n):
    p, st = sweeper.sweep(k)
    tb = (k * 100000) / 1000000.0  # (the double the reference forms from its integer microsecond header stamp: k * 0.1 differs from it in the last bit for some k, and a point or an IMU sample exactly on a boundary then falls on the other side)
    while ii < len(imu_t) and imu_t[ii] <= tb + 0.12:
        e.fastlio_imu_enqueue(imu_t[ii], imu_g[ii], imu_a[ii]); ii += 1
    e.fastlio_pcl_enqueue(p, st, tb)
    e.fastlio_main()
    while jj < len(imu_t) and imu_t[jj] <= tb + 0.12:
        R.imu_enqueue(imu_t[jj], imu_g[jj], imu_a[jj]); jj += 1
    R.pcl_enqueue(p, st, k * 100000)
    R.main()
    if k % 50 == 49:
        tk = (k + 1) * 0.1
        truth = R0.T @ (tr.pos(tk) - p0)
        Rt = R0.T @ tr.R(tk)
        sg, sr = e.get_state(), R.get_state()
        eg, er = sg[0:3] - truth, sr[0:3] - truth
        def yaw_err(s):
            Rs = synth.quat_to_R(s[3:7])
            d = Rt.T @ Rs
            return np.degrees(np.arctan2(d[1, 0], d[0, 0]))
        print(f"{k + 1:5d} {float(tr._d(tk)):9.1f} {np.linalg.norm(eg):10.3f} {np.linalg.norm(er):10.3f} {np.linalg.norm(sg[0:3] - sr[0:3]):12.4f}    {np.round(Rt.T @ eg, 3)}   {yaw_err(sg):8.3f} {yaw_err(sr):8.3f}", flush=True)
