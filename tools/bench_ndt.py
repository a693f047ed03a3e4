#!/usr/bin/env python
"""Localization-matcher timing (BASELINE.json config 4): 64x1875 scans, VoxelGrid(leaf) + NDT-P2D LM alignment against a
map resident in HBM.  Two targets: the reference's semantic (a <= 200k-point local map, localization.cpp:305-308) and the
prebuilt dense map.  Prints one JSON line per case.   python tools/bench_ndt.py [--dense-points 50000000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-slam-detection_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dense-points", type=int, default=50_000_000)
    ap.add_argument("--scans", type=int, default=100)
    ap.add_argument("--leaf", type=float, default=0.2)
    ap.add_argument("--ref-max-points", type=int, default=5_000_000, help="largest target also built by the reference's own kernels (0 = skip that leg)")
    args = ap.parse_args()
    import torch

    from lsd_amd import lio, synth

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import ref_ndt_cuda as ref

        if not ref.available() or args.ref_max_points <= 0:
            ref = None
    except Exception:
        ref = None

    dev = torch.device("cuda", 0)
    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
    rng = np.random.default_rng(7)
    scans, scans_host = [], []
    for k in range(8):
        pos = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), 1.8])
        q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
        raw, _ = synth.make_scan(scene, pos, q, seed=50 + k, fov_deg=(-24.8, 2.0), max_range=150.0)
        gp, gq = synth.perturb_pose(pos, q, seed=70 + k, max_t=0.5, max_deg=3.0)
        T, G = np.eye(4), np.eye(4)
        T[:3, :3], T[:3, 3] = synth.quat_to_R(q), pos
        G[:3, :3], G[:3, 3] = synth.quat_to_R(gq), gp
        scans.append((torch.from_numpy(raw).to(dev), len(raw), T, G))
        scans_host.append(raw)
    s = lio.Scan(max_raw=1 << 18, max_ds=200000)
    for name, npts in (("local_200k", 200_000), ("dense_5M", 5_000_000), ("dense", args.dense_points)):
        if name == "local_200k":  # <= 200k points within 30 m of the sensor, like the reference's local map
            pts = scene.sample_surface(3_200_000, seed=2, sigma=0.01)
            pts = pts[np.linalg.norm(pts[:, :2], axis=1) < 30.0][:200_000]
        else:
            pts = scene.sample_surface(npts, seed=2, sigma=0.01)
        n = lio.Ndt(resolution=1.0, search_method=7, max_points=len(pts), max_voxels=max(len(pts) // 4, 200_000), max_source_points=200000)
        d = torch.from_numpy(pts).to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n.set_target_device(d.data_ptr(), len(pts))
        nvox = n.num_voxels
        t_build = time.perf_counter() - t0
        del d
        errs, its, nds = [], [], []
        for w in range(8):
            s.set_device(scans[w][0].data_ptr(), scans[w][1])
            s.voxel_downsample(args.leaf)
            n.align(s, scans[w][3])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.scans):
            d_raw, n_raw, T, G = scans[i % len(scans)]
            s.set_device(d_raw.data_ptr(), n_raw)
            nds.append(s.voxel_downsample(args.leaf))
            Ta, conv, it = n.align(s, G)
            its.append(it + 1)
            errs.append(float(np.linalg.norm(Ta[:3, 3] - T[:3, 3])))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"case": name, "target_points": len(pts), "target_voxels": nvox, "target_build_ms": round(1e3 * t_build, 2),
                          "ms_per_scan": round(1e3 * dt / args.scans, 4), "scans_per_s": round(args.scans / dt, 1),
                          "registered_points_per_s": round(120000 * args.scans / dt, 1), "n_ds_avg": float(np.mean(nds)),
                          "lm_iterations_avg": float(np.mean(its)), "pos_err_m_median": float(np.median(errs)), "leaf": args.leaf}), flush=True)
        # ---- the reference's own kernels on this GPU (fast_gicp::cuda::NDTCudaCore compiled for gfx950, oracle/ref_ndt_cuda.hip -- test
        # infrastructure, timed here only as the baseline): target build and one linearisation (correspondences + cost + H + b)
        if ref is not None and len(pts) <= args.ref_max_points:
            import oracle as orc

            ds_host = [orc.voxel_downsample(scans_host[w], args.leaf) for w in range(2)]
            core = ref.NdtCudaCore(1.0, 7)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            core.set_target(pts)
            t_ref_build = time.perf_counter() - t0
            t_lin_ref, t_lin_hip = [], []
            for w in range(2):
                core.set_source(ds_host[w])
                s.set_ds(ds_host[w])
                for rep in range(6):
                    t0 = time.perf_counter()
                    lr = core.linearize(scans[w][3])
                    t1 = time.perf_counter()
                    lg = n.linearize(s, scans[w][3])
                    t2 = time.perf_counter()
                    if rep:
                        t_lin_ref.append(t1 - t0)
                        t_lin_hip.append(t2 - t1)
            # the whole alignment through the reference's registration object (fast_gicp::NDTCuda + LsqRegistration's LM loop)
            reg = ref.NdtCudaRegistration(1.0, 7)
            reg.set_target(pts)
            t_al_ref, t_al_hip, dpos = [], [], []
            for w in range(2):
                reg.set_source(ds_host[w])
                s.set_ds(ds_host[w])
                for rep in range(5):
                    t0 = time.perf_counter()
                    Tr, conv_r, it_r = reg.align(scans[w][3])
                    t1 = time.perf_counter()
                    Tg, conv_g, it_g = n.align(s, scans[w][3])
                    t2 = time.perf_counter()
                    if rep:
                        t_al_ref.append(t1 - t0)
                        t_al_hip.append(t2 - t1)
                dpos.append(float(np.linalg.norm(Tr[:3, 3] - Tg[:3, 3])))
            reg.close()
            print(json.dumps({"case": name + "_vs_reference_kernels", "reference_align_ms": round(1e3 * float(np.median(t_al_ref)), 3),
                              "hip_align_ms": round(1e3 * float(np.median(t_al_hip)), 3), "align_iterations": [it_r + 1, it_g + 1], "align_dpos_m": max(dpos), "target_points": len(pts), "reference_build_ms": round(1e3 * t_ref_build, 2),
                              "hip_build_ms": round(1e3 * t_build, 2), "reference_linearize_ms": round(1e3 * float(np.median(t_lin_ref)), 3),
                              "hip_linearize_ms": round(1e3 * float(np.median(t_lin_hip)), 3), "pairs_reference": lr["n_corr"], "pairs_hip": lg["n_corr"],
                              "cost_rel_diff": abs(lg["err"] / lr["err"] - 1), "note": "linearize: synchronous calls on both sides, same downsampled source; reference_build_ms includes its host-to-device copy of the cloud, hip_build_ms starts from a device-resident cloud"}), flush=True)
            core.close()
        n.close()
    # the step before the matcher in the localisation mode: constant-velocity motion compensation (slam_utils.cpp:163-191), stamps resident
    d_raw, n_raw, _, _ = scans[0]
    d_st = torch.from_numpy(rng.integers(0, 100000, n_raw).astype(np.uint32).view(np.int32)).to(dev)
    D = np.eye(4, dtype=np.float32)
    D[:3, :3] = synth.quat_to_R(synth.quat_from_rotvec([0.002, -0.001, 0.03])).astype(np.float32)
    D[:3, 3] = [1.2, 0.05, 0.0]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            s.set_device(d_raw.data_ptr(), n_raw)
            s.undistort_delta(d_st.data_ptr(), D, 0.1, on_device=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps({"case": "undistort_delta", "points": n_raw, "us_per_scan": round(1e6 * dt / 200, 2),
                      "algorithmic_GBps": round(n_raw * 36 / (dt / 200) / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
