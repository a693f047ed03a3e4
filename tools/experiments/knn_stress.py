"""randomised maps and queries against the oracle's iVox (the reference's statements): maps of clusters / planes / sparse noise built by a random number of
inserts (voxels grow and move), optionally under the LRU list, every stencil; queries near the data, far from it, on voxel faces and corners (multiples
of res / 2, +- one ulp), duplicated -- counts and neighbour lists bit for bit (default tie mode: positions in canonical order, the same point sets)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle")]
import oracle
from lsd_amd import lio


def make_points(rng, n):
    kind = int(rng.integers(0, 4))
    span = 10.0 ** rng.uniform(0.3, 1.8)
    if kind == 0:
        p = rng.uniform(-span, span, (n, 3)) * [1, 1, 0.1]
    elif kind == 1:
        k = int(rng.integers(2, 30))
        p = rng.uniform(-span, span, (k, 3))[rng.integers(0, k, n)] + rng.normal(0, 10.0 ** rng.uniform(-2, 0), (n, 3))
    elif kind == 2:
        p = np.stack([rng.uniform(-span, span, n), rng.uniform(-span, span, n), rng.normal(0, 0.02, n)], 1)
        p[: n // 3, 0] = span * 0.3 + rng.normal(0, 0.02, n // 3)  # a wall
    else:
        p = np.round(rng.uniform(-span, span, (n, 3)) * 8) / 8.0  # eighths of a metre: voxel faces and exact ties
    off = rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(0, 3)
    return np.concatenate([p + off, rng.uniform(0, 255, (n, 1))], 1).astype(np.float32)


def main(n_cfg=60, seed0=0):
    bad = checked = 0
    for c in range(n_cfg):
        rng = np.random.default_rng(seed0 * 10007 + c)
        st = int(rng.choice([1, 7, 19, 27, 75]))
        n = int(rng.choice([50, 2000, 20000, 60000]))
        pts = make_points(rng, n)
        use_lru = rng.random() < 0.4
        cap = int(rng.choice([2000, 8000])) if use_lru else 1 << 40
        m = lio.Map(resolution=0.5, stencil=st, max_points=400_000, max_voxels=60000)
        if use_lru:
            m.set_lru(cap, 1.0)
        o = oracle.IVox(res=0.5, stencil=st, capacity=cap, max_distance=1.0 if use_lru else 100.0)
        nb = int(rng.integers(1, 9))
        cuts = sorted(rng.choice(np.arange(1, len(pts)), min(nb - 1, len(pts) - 1), replace=False).tolist()) if nb > 1 and len(pts) > 2 else []
        travel, followed = 0.0, True
        for lo, hi in zip([0] + cuts, cuts + [len(pts)]):
            travel += 2.0
            m.add(pts[lo:hi], travel=travel)
            o.add(pts[lo:hi], travel=travel)
            if use_lru and m.lru_exact_stats()[1]:
                followed = False
                break
        if not followed:
            continue
        assert m.stats() == (o.num_points, o.num_voxels), (c, m.stats(), o.num_points, o.num_voxels)
        base = pts[rng.integers(0, len(pts), 1500), :3]
        q = np.concatenate([base + rng.normal(0, 0.2, base.shape), base[:300] + rng.normal(0, 3.0, (300, 3)), np.round(base[:300] * 4) / 4.0,
                            np.nextafter((np.round(base[:200] * 4) / 4.0).astype(np.float32), np.float32(1e9)), base[:100], base[:100]])
        q = np.concatenate([q, np.zeros((len(q), 1))], 1).astype(np.float32)
        got, cnt = m.knn(q)
        want, wcnt, _ = o.knn(q)
        checked += 1
        ok = np.array_equal(cnt, wcnt) and np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32)) and np.array_equal(np.sort(got[..., 3], 1), np.sort(want[..., 3], 1))
        if not ok:
            bad += 1
            rows = np.flatnonzero((cnt != wcnt) | np.any(got[..., :3].view(np.uint32).reshape(len(q), -1) != want[..., :3].view(np.uint32).reshape(len(q), -1), axis=1))
            print("MISMATCH cfg", c, dict(st=st, n=n, lru=use_lru, cap=cap, batches=nb), "rows", len(rows), rows[:5], "cnt", cnt[rows[:3]], wcnt[rows[:3]])
    print("configurations", n_cfg, "compared", checked, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
