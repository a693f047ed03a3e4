"""randomised clouds against the oracle's VoxelGrid restatement (PCL semantics), bit for bit and in order: sizes 1 ... 260 000, boxes from decimetres to
kilometres (1 to 4 radix passes, the int32 overflow guard), leaves 0.1 ... 2 m, uniform / clustered / beam-like / duplicated / one-voxel clouds, NaN and inf
sprinkled in -- through one scan object (so that the pass prediction of the previous cloud meets the next) and through the batched form"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle")]
import oracle
from lsd_amd import lio


def cloud(rng):
    n = int(rng.choice([1, 2, 5, 63, 64, 65, 2047, 2048, 2049, 4097, int(rng.integers(100, 30000)), int(rng.integers(30000, 260000))]))
    kind = int(rng.integers(0, 6))
    half = np.array([10.0 ** rng.uniform(-1, 3.0), 10.0 ** rng.uniform(-1, 3.0), 10.0 ** rng.uniform(-1, 1.7)])
    centre = rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(0, 3.5)
    if kind == 0:
        p = rng.uniform(-1, 1, (n, 3)) * half
    elif kind == 1:
        p = rng.normal(0, 1, (n, 3)) * half * 0.3
    elif kind == 2:  # a few dense clusters: monster voxels
        k = int(rng.integers(1, 6))
        p = rng.uniform(-1, 1, (k, 3))[rng.integers(0, k, n)] * half + rng.normal(0, 10.0 ** rng.uniform(-3, -0.5), (n, 3))
    elif kind == 3:  # rays from a sensor: dense near, sparse far
        d = rng.uniform(0.5, 1.0, n) ** 3 * half.max()
        a, e = rng.uniform(-np.pi, np.pi, n), rng.uniform(-0.4, 0.1, n)
        p = np.stack([d * np.cos(a) * np.cos(e), d * np.sin(a) * np.cos(e), d * np.sin(e)], 1)
    elif kind == 4:  # duplicates of few points
        p = (rng.uniform(-1, 1, (max(1, n // 50), 3)) * half)[rng.integers(0, max(1, n // 50), n)]
    else:  # everything in one voxel
        p = rng.uniform(0, 0.04, (n, 3))
    p = (p + centre).astype(np.float32)
    out = np.concatenate([p, rng.uniform(0, 255, (n, 1)).astype(np.float32)], 1)
    if rng.random() < 0.3 and n > 3:
        out[rng.integers(0, n, max(1, n // 20)), rng.integers(0, 3)] = np.nan
        out[rng.integers(0, n, max(1, n // 30)), rng.integers(0, 3)] = np.inf if rng.random() < 0.5 else -np.inf
    return np.ascontiguousarray(out, np.float32)


def main(n_cfg=150, seed0=0):
    rng = np.random.default_rng(seed0)
    sc = lio.Scan(max_raw=1 << 18, max_ds=1 << 18)
    bad = 0
    pend = []
    for c in range(n_cfg):
        pts = cloud(rng)
        leaf = float(rng.choice([0.1, 0.2, 0.5, 1.0, 2.0]))
        ref = oracle.voxel_downsample(pts, leaf)
        sc.upload(pts)
        try:
            n = sc.voxel_downsample(leaf)
            got = sc.get_ds()
            ok = n == len(ref) and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        except Exception as ex:  # capacity errors are legitimate only when the oracle's output does not fit either
            ok = len(ref) > (1 << 18)
            if not ok:
                print("EXCEPTION", c, len(pts), leaf, repr(ex)[:200])
        if not ok:
            bad += 1
            print("MISMATCH single", c, "n", len(pts), "leaf", leaf, "n_ref", len(ref))
        if leaf == 0.5 and len(pts) <= (1 << 17):
            pend.append((pts, ref))
    # the batched form (leaf 0.5 inside the batch engine's chain): groups of up to 8 clouds through lio.voxel_downsample_batch if the binding has it
    fn = getattr(lio.Scan, "voxel_downsample_batch", None)
    if fn is not None and pend:
        scans = [lio.Scan(max_raw=1 << 17, max_ds=1 << 17) for _ in range(8)]
        for i in range(0, len(pend), 8):
            grp = pend[i:i + 8]
            for s_, (pts, _) in zip(scans, grp):
                s_.upload(pts)
            ns = fn(scans[:len(grp)], 0.5)
            for s_, n_, (pts, ref) in zip(scans, ns, grp):
                if not (n_ == len(ref) and np.array_equal(s_.get_ds().view(np.uint32), ref.view(np.uint32))):
                    bad += 1
                    print("MISMATCH batched n", len(pts), "n_ref", len(ref))
    print("clouds", n_cfg, "batched", len(pend) if fn is not None else 0, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
