import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle
from lsd_amd import lio, synth
scene = synth.Scene(half=60.0, n_boxes=20, seed=3)
rng = np.random.default_rng(8)
pts = scene.sample_surface(300_000, seed=11, sigma=0.01)
pts = pts[(np.abs(pts[:, 1]) < 25) & (pts[:, 2] < 6)]
def _batch(scene_pts, cx, rng, n, half=8.0):
    sel = np.flatnonzero(np.abs(scene_pts[:, 0] - cx) < half)
    return scene_pts[rng.choice(sel, size=min(n, len(sel)), replace=False)]
def rows(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 4)
    return a[np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))]
for cap, maxd, course in ((4000, 2.0, [-30.0, 0.0, 30.0] * 6), (5000, 0.5, [float(c) for c in rng.uniform(-30, 30, 30)]), (3500, 1.0, [(-1) ** k * (5.0 + 0.7 * k) for k in range(30)])):
    m = lio.Map(resolution=0.5, stencil=19, max_points=600_000, max_voxels=40000)
    m.set_lru(cap, maxd)
    o = oracle.IVox(res=0.5, stencil=19, capacity=cap, max_distance=maxd)
    travel = 0.0
    for b, cx in enumerate(course):
        travel += 4.0
        batch = _batch(pts, cx, rng, 2500)
        m.add(batch, travel=travel); o.add(batch, travel=travel)
        same = np.array_equal(rows(m.dump()), rows(o.dump()))
        print(cap, b, cx, m.stats(), (o.num_points, o.num_voxels), 'lru', m.lru_stats(), 'exact', m.lru_exact_stats(), 'SAME' if same else 'DIFF')
