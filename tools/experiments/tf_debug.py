"""Debug helper (GPU box): the config-3 drive's first sweeps, a tie-mode-2 HIP engine teacher-forced beside the pinned build of the reference,
one line per sweep: |dpos|, voxel counts on both sides, the engine's tie statistics.  python tools/experiments/tf_debug.py [n_sweeps] [speed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle")]
import torch

import ref_fastlio
from lsd_amd import capi, lio, synth, synth_gpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
speed = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
tie_mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
force_map = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0)
scene = synth.Scene(half=500.0, n_boxes=1500, seed=3, box_size=(4.0, 30.0), keep_clear=8.0)
tr = synth_gpu.Lawnmower(speed=speed)
sweeper = synth_gpu.Sweeper(scene, tr, dev, fov_deg=(-24.8, 2.0), max_range=100.0, seed=1000)
imu_t, imu_g, imu_a = synth_gpu.imu_stream(tr, 0.0, n * 0.1 + 0.3, rate=100.0, seed=1000, gyr_sigma=1e-3, acc_sigma=1e-2)
R = ref_fastlio.RefFastLio(scan_period=0.1)
R.set_logging(False)
e = lio.Engine(resolution=0.5, stencil=75, max_points=6_000_000, max_voxels=(1 << 21), max_raw=1 << 18, max_ds=100000, device=0)
e.map.set_lru((1 << 21) - 100_000, 1e9)
e.fastlio_init(scan_period=0.1)
e.map.set_tie_mode(tie_mode)
jj = 0
for k in range(n):
    p, st = sweeper.sweep(k)
    o = np.argsort(st, kind="stable")
    p, st = np.ascontiguousarray(p[o]), st[o].astype(np.int64)
    thin = int(np.ceil(len(p) / 90000.0))
    if thin > 1:
        p, st = np.ascontiguousarray(p[::thin]), st[::thin]
    ii = np.arange(len(st))
    st = (np.maximum.accumulate(st - ii) + ii).astype(np.uint32)
    tb = (k * 100000) / 1000000.0  # (the double the reference forms from its integer microsecond header stamp: k * 0.1 differs from it in the last bit for some k, and a point or an IMU sample exactly on a boundary then falls on the other side)
    while jj < len(imu_t) and imu_t[jj] <= tb + 0.12:
        R.imu_enqueue(imu_t[jj], imu_g[jj], imu_a[jj])
        e.fastlio_imu_enqueue(imu_t[jj], imu_g[jj], imu_a[jj])
        jj += 1
    R.pcl_enqueue(p, st, k * 100000)
    upd = R.main()
    e.fastlio_pcl_enqueue(p, st, tb)
    rc = e.fastlio_main()
    e.flush()
    s_ref, _, P_ref = R.state()
    s = e.get_state()
    info = R.info()
    tm = e.timings()
    extra = ""
    if rc == capi.MAIN_UPDATED:
        a, b = e.scan.get_ds(), R.down_body()
        if a.shape == b.shape:
            dd = np.abs(a[:, :3] - b[:, :3]).max(axis=1)
            extra = " ds_differ %d max %.2e" % (int((dd > 0).sum()), float(dd.max()))
        else:
            extra = " ds_shape %s %s" % (a.shape, b.shape)
        ua, ub = e.undistorted(), R.undistorted()
        ua = ua[np.isfinite(ua[:, 0])]
        if ua.shape[0] == ub.shape[0]:
            du = np.abs(ua[:, :3] - ub[:, :3]).max(axis=1)
            extra += " und_differ %d max %.2e" % (int((du > 0).sum()), float(du.max()))
        else:
            extra += " und_n %d %d" % (ua.shape[0], ub.shape[0])
    print(k, "rc", rc, upd, "dpos %.3e drot %.3e" % (np.linalg.norm(s[:3] - s_ref[:3]), synth.quat_angle(s[3:7], s_ref[3:7])), "voxels", e.map.stats()[1], R.map_voxels(),
          "n_ds", tm["n_ds"], info["feats_down_size"], "n_eff", tm.get("n_eff_last"), info["effct_feat_num"], "passes", tm["n_pass"], tm["n_knn_pass"], "ties", e.map.tie_stats(),
          extra, flush=True)
    if rc == capi.MAIN_UPDATED:
        e.set_state(s_ref)
        e.set_cov(P_ref)
    if force_map and e.map.stats()[1] > 0:
        e.map.clear()
        e.map.add(R.map_dump(), float(info["travel_distance"]))
