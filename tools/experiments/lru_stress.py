"""randomised courses / quotas / ages against the oracle's sequential list (the reference's statements): whenever the device reports that it followed the
batch's point-by-point order (lio_map_lru_exact_stats: not_followed unchanged), its map must equal the oracle's after the batch -- voxels, points, dump"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle")]
import oracle
from lsd_amd import lio, synth


def rows(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 4)
    return a[np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))]


def main(n_cfg=60, seed0=0):
    scene = synth.Scene(half=60.0, n_boxes=20, seed=3)
    pts = scene.sample_surface(300_000, seed=11, sigma=0.01)
    pts = pts[(np.abs(pts[:, 1]) < 25) & (pts[:, 2] < 6)]
    bad = followed = skipped = recreated_total = 0
    for c in range(n_cfg):
        rng = np.random.default_rng(seed0 + c)
        cap = int(rng.choice([800, 1500, 2500, 4000, 6000]))
        maxd = float(rng.choice([0.0, 0.5, 3.0, 10.0, 40.0]))
        npts = int(rng.choice([300, 1200, 2500]))
        half = float(rng.choice([3.0, 8.0]))
        step = float(rng.choice([0.5, 4.0]))
        kind = int(rng.integers(0, 3))
        m = lio.Map(resolution=0.5, stencil=19, max_points=600_000, max_voxels=40000)
        m.set_lru(cap, maxd)
        o = oracle.IVox(res=0.5, stencil=19, capacity=cap, max_distance=maxd)
        travel, nf_prev, ok = 0.0, 0, True
        for b in range(24):
            cx = [rng.uniform(-30, 30), (-1) ** b * (3.0 + 0.9 * b), [-25.0, 0.0, 25.0][b % 3]][kind]
            travel += step
            sel = np.flatnonzero(np.abs(pts[:, 0] - cx) < half)
            batch = pts[rng.choice(sel, size=min(npts, len(sel)), replace=False)]
            m.add(batch, travel=travel)
            o.add(batch, travel=travel)
            rec, nf = m.lru_exact_stats()
            if nf != nf_prev:  # this batch was not followed: the maps may differ from here on -- start both afresh from the device's map? no: stop comparing this course
                skipped += 1
                ok = False
                break
            followed += 1
            if m.stats() != (o.num_points, o.num_voxels) or (b % 4 == 3 and not np.array_equal(rows(m.dump()), rows(o.dump()))):
                print("MISMATCH cfg", c, "batch", b, dict(cap=cap, maxd=maxd, npts=npts, half=half, step=step, kind=kind), m.stats(), (o.num_points, o.num_voxels), m.lru_stats(), (rec, nf))
                bad += 1
                ok = False
                break
        if ok and not np.array_equal(rows(m.dump()), rows(o.dump())):
            print("MISMATCH at end cfg", c)
            bad += 1
        recreated_total += m.lru_exact_stats()[0]
    print("configurations", n_cfg, "batches followed and equal", followed, "courses cut short (not followed)", skipped, "mismatches", bad, "voxels re-created", recreated_total)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 60) else 0)
