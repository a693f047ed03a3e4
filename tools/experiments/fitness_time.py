"""lio_ndt_fitness_score / lio_ndt_overlap_score: time per call on a key-frame-sized source against a local-map-sized target"""
import sys, time, numpy as np
sys.path.insert(0, 'lidar-slam-detection_amd/python')
from lsd_amd import lio, synth
sc = synth.Scene(half=80.0, n_boxes=40, seed=5)
rng = np.random.default_rng(1)
def world_scan(x, y, a, seed):
    pos, qq = np.array([x, y, 1.8]), synth.quat_from_rotvec([0, 0, a])
    c = synth.make_scan(sc, pos, qq, seed=seed, max_range=80.0)[0][:, :4].astype(np.float64)
    c[:, :3] = c[:, :3] @ synth.quat_to_R(qq).T + pos
    return c
tgt = np.concatenate([world_scan(x, y, a, s) for s, (x, y, a) in enumerate([(0, 0, 0), (6, 1, 0.3), (-5, 3, -0.4), (2, -7, 1.0)])]).astype(np.float32)
p, q = np.array([1.0, 0.5, 1.8]), synth.quat_from_rotvec([0, 0, 0.1])
raw = synth.make_scan(sc, p, q, seed=77, max_range=80.0)[0][:, :4].astype(np.float32)
T = np.eye(4); T[:3, :3] = synth.quat_to_R(q); T[:3, 3] = p
ndt = lio.Ndt(resolution=1.0, max_points=len(tgt) + 16, max_source_points=len(raw) + 16)
ndt.set_target(tgt)
scan = lio.Scan(max_raw=len(raw) + 16, max_ds=len(raw) + 16)
scan.upload(raw); n_ds = scan.voxel_downsample(0.2)
for far in (0.0, 30.0):  # a source 30 m off: every point walks the rings out to the range
    Tm = T.copy(); Tm[0, 3] += far
    for max_range in (25.0, 4.0):
        ndt.fitness_score(scan, Tm, max_range)
        t0 = time.perf_counter()
        for _ in range(10): s, n_in = ndt.fitness_score(scan, Tm, max_range)
        t1 = time.perf_counter()
        print(f"target {len(tgt)} pts, source {n_ds} pts, offset {far} m, max_range^2 {max_range}: {1e3 * (t1 - t0) / 10:.3f} ms per call, score {s:.6g}, inliers {n_in}")
