#!/usr/bin/env python
"""Generate tests/golden/*.npz: small frozen input/output vectors for the LIO hot path.

Run in the build container (needs /root/reference for the pinned pieces):
    make -C oracle all ref && python tools/make_golden.py

Provenance of every expected output is recorded in the file ("source" field):
  * esti_plane.npz, ivox_knn.npz ...... the REFERENCE'S OWN CODE (oracle/_ref/libref_harness.so = esti_plane and
                                        faster_lio::IVox compiled from /root/reference, scalar Eigen build)
  * voxelgrid.npz, linearize.npz, update.npz ... the CPU oracle (oracle/lio_oracle.cpp), whose esti_plane and iVox
                                        are themselves checked bit-for-bit against the files above; PCL VoxelGrid
                                        is not in the tree and stays oracle-defined.
  * ukf.npz ........................... the reference's UKF + pose system (oracle/_ref/libref_ukf.so)
  * fastlio_drive.npz ................. the reference's whole FastLIO translation units (oracle/_ref/libref_fastlio.so)
  * undistort_delta.npz ............... the reference's slam_utils.cpp undistortPoints (oracle/_ref/libref_slam_utils.so)
  * pose_estimator.npz ................ the reference's hdl_localization::PoseEstimator (oracle/_ref/libref_pose_estimator.so)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
import ref as refmod  # noqa: E402
from lsd_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def fastlio_drive():
    """a 14-scan synthetic drive through the reference's OWN FastLIO translation units (oracle/_ref/libref_fastlio.so), once with
    its neighbour lists put in canonical order, once untouched; replayed by tests/test_fastlio_golden.py (oracle on the CPU, HIP
    path on the GPU box, where /root/reference does not exist)"""
    import ref_fastlio
    import test_fastlio_vs_ref as tf

    if not ref_fastlio.available():
        raise SystemExit("oracle/_ref/libref_fastlio.so missing: run `make -C oracle ref` where /root/reference is mounted")
    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)  # tests/conftest.py `scene`
    out = {}
    for name, canonical, distinct in (("canonical", True, True), ("native", False, False)):
        rec = dict(und6=None, vox=[], init=[], n_eff=[], start=[], odom_e=[])

        def on_scan(k, L, R, o):
            if k == 6:
                rec["und6"] = R.undistorted()[:, :4].copy()
                rec["ds6"] = R.down_body()
            rec["vox"].append(R.map_voxels())
            rec["init"].append(R.is_init())
            rec["n_eff"].append(R.info()["effct_feat_num"])
            rec["start"].append(o["ref_start"])
            rec["odom_e"].append(R.odometry()[1])

        _, _, _, res = tf._drive(oracle, scene, 14, canonical=canonical, distinct=distinct, on_scan=on_scan)
        out[name + "_state"] = np.stack([r["ref"] for r in res])
        out[name + "_P"] = np.stack([r["ref_P"] for r in res])
        out[name + "_start"] = np.stack(rec["start"])
        out[name + "_odom_e"] = np.stack(rec["odom_e"])
        out[name + "_voxels"] = np.array(rec["vox"])
        out[name + "_is_init"] = np.array(rec["init"])
        out[name + "_n_eff"] = np.array(rec["n_eff"])
        if canonical:  # (the untouched run's cloud is the same set in std::sort's order)
            out["und6"] = rec["und6"]
            out["ds6"] = rec["ds6"]
    np.savez_compressed(os.path.join(OUT, "fastlio_drive.npz"), n_scans=14, **out,
                        source="laserMapping.cpp + IMU_Processing.hpp + preprocess.cpp + iVox + IKFoM compiled whole from /root/reference "
                               "(oracle/ref_fastlio.cpp); pcl::VoxelGrid = the oracle's restatement; inputs = tests/test_fastlio_vs_ref.py _drive")


def undistort_delta_cases(rng, n=3000):
    """inputs of the constant-velocity compensation: rotations from 3 rad down to exactly none, with and without translation"""
    pts = (rng.normal(size=(n, 4)) * 30).astype(np.float32)
    pts[7, :3] = np.nan
    st = rng.integers(0, 100001, n).astype(np.uint32)
    st[:5] = [0, 100000, 1, 99999, 50000]
    deltas = []
    for case, scale in enumerate([3.0, 1e-1, 1e-3, 1e-6, 1e-8, 0.0, 2.5, 1e-2]):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = synth.quat_to_R(synth.quat_from_rotvec(rng.normal(size=3) * scale)).astype(np.float32)
        T[:3, 3] = rng.normal(size=3) * (1.0 if case % 3 else 0.0)
        deltas.append(T)
    return pts, st, np.stack(deltas)


def undistort_poses_cases(rng, n=3000):
    """inputs of the pose-list compensation: 2 .. 12 poses, time-ordered and unordered clouds, a first pose at / after the scan start, a pose
    before the header stamp (its interval 'never ends'), stamps past the last pose"""
    pts = (rng.normal(size=(n, 4)) * 30).astype(np.float32)
    cases = []
    for case in range(8):
        m = [2, 3, 6, 12][case % 4]
        header = 1_700_000_000_000_000 + 7 * case
        ps = (header + np.sort(rng.integers(0, 110000 if case % 2 else 90000, m))).astype(np.uint64)
        if case % 3 == 0:
            ps[0] = header
        if case == 5:
            ps[1] = header - 5
        Ts = np.stack([np.eye(4)] * m)
        for i in range(1, m):
            Ts[i, :3, :3] = synth.quat_to_R(synth.quat_from_rotvec(rng.normal(size=3) * [1e-1, 1e-3, 0.0][case % 3]))
            Ts[i, :3, 3] = rng.normal(size=3) * 0.5
        st = rng.integers(0, 100001, n).astype(np.uint32)
        if case < 6:
            st = np.sort(st)
        cases.append(dict(header=header, pose_stamps=ps, pose_T=Ts, stamp_us=st))
    return pts, cases


def slam_utils():
    """undistortPoints(delta_pose, ...) through the reference's OWN slam_utils.cpp (oracle/_ref/libref_slam_utils.so)"""
    import ref_slam_utils as rs

    if not rs.available():
        raise SystemExit("oracle/_ref/libref_slam_utils.so missing: run `make -C oracle ref` where /root/reference is mounted")
    pts, st, deltas = undistort_delta_cases(np.random.default_rng(5))
    out = np.stack([rs.undistort_delta(pts, st, D, 0.1) for D in deltas])
    np.savez_compressed(os.path.join(OUT, "undistort_delta.npz"), points=pts, stamp_us=st, deltas=deltas, scan_period=0.1, out=out,
                        source="slam/common/slam_utils.cpp:163-191 compiled whole from /root/reference (oracle/ref_slam_utils.cpp); pcl::transformPoint = PCL 1.9.1's one-liner")
    pts, cases = undistort_poses_cases(np.random.default_rng(6), n=1500)
    rec = dict(points=pts, n_cases=len(cases))
    for k, c in enumerate(cases):
        rec[f"c{k}_header"], rec[f"c{k}_pose_stamps"], rec[f"c{k}_pose_T"], rec[f"c{k}_stamp_us"] = np.uint64(c["header"]), c["pose_stamps"], c["pose_T"], c["stamp_us"]
        rec[f"c{k}_out"] = rs.undistort_poses(pts, c["stamp_us"], c["header"], c["pose_stamps"], c["pose_T"])
    np.savez_compressed(os.path.join(OUT, "undistort_poses.npz"), **rec,
                        source="slam/common/slam_utils.cpp:193-228 compiled whole from /root/reference (oracle/ref_slam_utils.cpp)")


def pose_estimator():
    """the answers of the reference's hdl_localization::PoseEstimator (oracle/_ref/libref_pose_estimator.so) to the two scripted drives of
    tests/test_pose_estimator_vs_ref.py, in call order"""
    import test_pose_estimator_vs_ref as tp

    if not tp.rp.available():
        raise SystemExit("oracle/_ref/libref_pose_estimator.so missing: run `make -C oracle ref` where /root/reference is mounted")
    out = {}
    for key, fn in (("loop", tp.drive_filter_loop), ("gnss", tp.drive_gnss_only)):  # this order: predict_imu's static dt_smooth carries over
        rec = fn("record")
        out[key + "_flat"] = np.concatenate(rec.log)
        out[key + "_off"] = np.cumsum([0] + [len(v) for v in rec.log])
    np.savez_compressed(os.path.join(OUT, "pose_estimator.npz"), **out,
                        source="hdl_localization/src/pose_estimator.cpp compiled whole from /root/reference (oracle/ref_pose_estimator.cpp), mock matcher")


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "pose_estimator":
        pose_estimator()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fastlio":
        fastlio_drive()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "slam_utils":
        slam_utils()
        return
    if not refmod.available():
        raise SystemExit("oracle/_ref/libref_harness.so missing: run `make -C oracle ref` where /root/reference is mounted")
    from test_oracle_vs_ref import _plane_sets

    # 1. esti_plane: reference code
    rng = np.random.default_rng(42)
    sets = np.stack(_plane_sets(rng, 600))
    ok = np.zeros(len(sets), np.uint8)
    pabcd = np.zeros((len(sets), 4), np.float32)
    for i, p in enumerate(sets):
        o, r = refmod.esti_plane(p)
        ok[i], pabcd[i] = o, r
    np.savez_compressed(os.path.join(OUT, "esti_plane.npz"), points=sets, ok=ok, pabcd=pabcd,
                        source="reference common_lib.h:236-268 via oracle/_ref (scalar Eigen)")

    # 2. iVox kNN: reference code
    scene = synth.Scene(half=30.0, n_boxes=8, seed=5)
    map_pts = scene.sample_surface(8000, seed=6, sigma=0.01)
    q = map_pts[rng.choice(len(map_pts), 400, replace=False)].copy()
    q[:, :3] += rng.normal(0, 0.15, (400, 3)).astype(np.float32)
    q[:10, 2] += 30.0
    res = {}
    for st in (19, 75):
        iv = refmod.IVox(stencil=st)
        iv.add(map_pts[:5000], 0.0)
        iv.add(map_pts[5000:], 1.0)
        nn, cnt = iv.knn(q)
        res[f"nn{st}"] = refmod.canonical(nn, cnt, q)[..., :3]
        res[f"cnt{st}"] = cnt
        res[f"voxels{st}"] = iv.num_voxels
    np.savez_compressed(os.path.join(OUT, "ivox_knn.npz"), map=map_pts, queries=q, **res,
                        source="reference ivox3d.h:139-171,231-256 via oracle/_ref; lists sorted by (d2,x,y,z)")

    # 3. voxel grid (PCL semantics restated; unpinned)
    raw, _ = synth.make_scan(scene, [0.5, -1.0, 1.7], synth.quat_from_rotvec([0, 0, 0.4]), seed=8, n_beams=32, n_az=300)
    raw = raw.copy()
    raw[::97, 1] = np.nan
    ds = oracle.voxel_downsample(raw, 0.5)
    np.savez_compressed(os.path.join(OUT, "voxelgrid.npz"), raw=raw, leaf=np.float32(0.5), ds=ds,
                        source="oracle restatement of PCL 1.9.1 VoxelGrid::applyFilter (PCL not in the reference tree: unpinned)")

    # 4/5. one linearisation and one iterated update
    true_pos, true_q = np.array([0.5, -1.0, 1.7]), synth.quat_from_rotvec([0, 0, 0.4])
    gp, gq = synth.perturb_pose(true_pos, true_q, seed=9, max_t=0.15, max_deg=1.0)
    state = synth.state_from_pose(gp, gq)
    big_map = scene.sample_surface(60000, seed=10, sigma=0.01)
    o = oracle.Lio(stencil=19, capacity=1 << 40, threads=4)
    o.map_add(big_map)
    o.set_state(state)
    o.set_cov(oracle.init_cov())
    o.set_flags(ekf_inited=True, first_scan=False)
    o.set_ds(ds)
    lin = o.linearize(converge=True)
    np.savez_compressed(os.path.join(OUT, "linearize.npz"), map=big_map, ds=ds, state=state, selected=lin["selected"],
                        normvec=lin["normvec"], nn_cnt=lin["nn_cnt"], nn=lin["nn"][..., :3], JtJ=lin["JtJ"], Jtr=lin["Jtr"],
                        sum_abs_res=lin["sum_abs_res"], n_eff=lin["n_eff"],
                        source="oracle h_share_model_geometric (laserMapping.cpp:813-932); its kNN and esti_plane are pinned to the reference")
    o2 = oracle.Lio(stencil=19, capacity=1 << 40, threads=4)
    o2.map_add(big_map)
    o2.set_state(state)
    o2.set_cov(oracle.init_cov())
    o2.set_flags(ekf_inited=True, first_scan=False)
    o2.set_ds(ds)
    logs = o2.update()
    np.savez_compressed(os.path.join(OUT, "update.npz"), state0=state, P0=oracle.init_cov(), state1=o2.get_state(), P1=o2.get_cov(),
                        knn=np.array([l["knn"] for l in logs]), n_eff=np.array([l["n_eff"] for l in logs]),
                        dx=np.stack([l["dx"] for l in logs]), true_pos=true_pos, true_q=true_q,
                        source="oracle update_iterated_dyn_share_modified (esekfom.hpp:1619-1931); IKFoM needs Boost: unpinned")
    # 6. localization matcher (NDT-P2D): one linearisation and one alignment
    import ndt as ondt

    nd_map = scene.sample_surface(40000, seed=12, sigma=0.02)
    nd = ondt.Ndt(1.0, 7)
    nd.set_target(nd_map)
    nd.set_source(ds)
    T_true = np.eye(4)
    T_true[:3, :3] = synth.quat_to_R(true_q)
    T_true[:3, 3] = true_pos
    gp2, gq2 = synth.perturb_pose(true_pos, true_q, seed=13, max_t=0.4, max_deg=2.5)
    T_guess = np.eye(4)
    T_guess[:3, :3] = synth.quat_to_R(gq2)
    T_guess[:3, 3] = gp2
    lin_n = nd.linearize(T_guess)
    T_al, conv, its = nd.align(T_guess)
    np.savez_compressed(os.path.join(OUT, "ndt.npz"), map=nd_map, ds=ds, T_guess=T_guess, T_true=T_true, n_voxels=nd.num_voxels,
                        n_corr=lin_n["n_corr"], H=lin_n["H"], b=lin_n["b"], err=lin_n["err"], T_aligned=T_al, converged=conv, iterations=its,
                        source="oracle/ndt_oracle.cpp (fast_gicp NDTCuda P2D restated; CUDA sources not buildable: unpinned except se3_exp / Eigen pieces)")
    # ---- localisation UKF: a scripted drive through the reference's OWN filter code (oracle/_ref/libref_ukf.so) ----
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_ukf_vs_ref as tu  # the script (IMU steps, IMU-less steps, observations) lives with the test that replays it

    ops = tu.script(0, 60)
    np.savez_compressed(os.path.join(OUT, "ukf.npz"), seed=0, n=60, trace=tu.run_reference(ops).astype(np.float32),
                        source="kkl/alg/unscented_kalman_filter.hpp + hdl_localization/pose_system.hpp compiled from /root/reference (oracle/ref_ukf.cpp)")
    fastlio_drive()
    slam_utils()
    pose_estimator()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
