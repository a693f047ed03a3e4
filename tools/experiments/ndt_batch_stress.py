"""lio_ndt_align_batch against lio_ndt_align job by job on random set-ups: resolutions, neighbour modes, 1 ... 90 jobs (more than a launch's slots), sources of
3 ... 30 000 points, good and hopeless guesses, one or several targets -- same convergence flags and iteration counts, poses within 1e-9, a second call bit
for bit"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python")]
from lsd_amd import lio, synth


def main(n_cfg=10, seed0=0):
    bad = jobs_n = 0
    for c in range(n_cfg):
        rng = np.random.default_rng(seed0 * 8191 + c)
        scene = synth.Scene(half=60.0, n_boxes=int(rng.choice([5, 25])), seed=int(rng.integers(1, 1000)))
        res, method = float(rng.choice([0.5, 1.0, 2.0])), int(rng.choice([1, 7, 27]))
        target = scene.sample_surface(int(rng.choice([50_000, 400_000])), seed=int(rng.integers(1, 1000)), sigma=0.01)
        tg = [lio.Ndt(resolution=res, search_method=method, max_points=len(target) + 1, max_voxels=400_000, max_source_points=1 << 16)]
        tg[0].set_target(target)
        if rng.random() < 0.4:
            half = np.ascontiguousarray(target[target[:, 0] < 10.0])
            t2 = lio.Ndt(resolution=res, search_method=method, max_points=len(half) + 1, max_voxels=400_000, max_source_points=1 << 16)
            t2.set_target(half)
            tg.append(t2)
        scans, guesses, which = [], [], []
        for k in range(int(rng.integers(1, 91))):
            pos = np.array([rng.uniform(-6, 6), rng.uniform(-6, 6), 1.8])
            q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
            raw, _ = synth.make_scan(scene, pos, q, seed=int(rng.integers(1, 1 << 30)), n_az=int(rng.choice([3, 40, 300, 500])), fov_deg=(-24.8, 2.0))
            sc = lio.Scan(max_raw=1 << 17, max_ds=1 << 16)
            sc.upload(raw)
            sc.voxel_downsample(float(rng.choice([0.2, 0.5])))
            gp, gq = synth.perturb_pose(pos, q, seed=int(rng.integers(1, 1 << 30)), max_t=0.4, max_deg=2.5)
            if rng.random() < 0.05:
                gp = gp + [25.0, -20.0, 0.0]
            G = np.eye(4)
            G[:3, :3], G[:3, 3] = synth.quat_to_R(gq), gp
            scans.append(sc)
            guesses.append(G)
            which.append(int(rng.integers(0, len(tg))))
        targets = [None if w == 0 else tg[w] for w in which]
        batch = tg[0].align_batch(scans, guesses, targets=targets if len(tg) > 1 else None)
        again = tg[0].align_batch(scans, guesses, targets=targets if len(tg) > 1 else None)
        for k, (Tb, cb, itb, evals, rc) in enumerate(batch):
            jobs_n += 1
            Ts, cs, its = tg[which[k]].align(scans[k], guesses[k])
            ok = rc == 0 and (cb, itb) == (cs, its) and float(np.abs(Tb - Ts).max()) < 1e-9 and np.array_equal(again[k][0], Tb) and again[k][1:] == batch[k][1:]
            if not ok:
                bad += 1
                print("MISMATCH cfg", c, "job", k, dict(res=res, method=method, n_jobs=len(scans), targets=len(tg)), "rc", rc, "conv", cb, cs, "it", itb, its, "dT", float(np.abs(Tb - Ts).max()))
    print("configurations", n_cfg, "alignments compared", jobs_n, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
