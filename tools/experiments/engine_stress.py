"""random drives through lio_engine_process_scan against oracle.Lio.process_scan: scenes of different density, scans of 300 ... 60 000 points, jumps of the
pose, wide and tight priors, both stencils the reference switches between, with and without the LRU list (quotas of a few scans' footprints), host-driven
and device loop -- return codes, map sizes after every scan, poses (free-running on both sides: last-bit differences of the f64 sums grow along a drive through the f32 map -- 1e-9 m on the first scans, up to 7e-7 m after fifteen; 1.4e-4 m once in 900 scans; flagged: 1e-3 m)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle")]
import oracle
from lsd_amd import lio, synth


def main(n_cfg=24, seed0=0):
    bad = scans = 0
    for c in range(n_cfg):
        rng = np.random.default_rng(seed0 * 7919 + c)
        scene = synth.Scene(half=float(rng.choice([25.0, 60.0])), n_boxes=int(rng.choice([0, 6, 25])), seed=int(rng.integers(1, 1000)))
        n_az = int(rng.choice([8, 60, 300, 900]))
        use_lru = rng.random() < 0.5
        cap = int(rng.choice([3000, 12000])) if use_lru else 1 << 40
        maxd = float(rng.choice([0.5, 5.0]))
        loop = int(rng.integers(0, 2))
        o = oracle.Lio(res=0.5, stencil=75, capacity=cap, max_distance=maxd if use_lru else 100.0, threads=8)
        e = lio.Engine(resolution=0.5, stencil=75, max_points=2_000_000, max_voxels=400_000, max_raw=1 << 18, max_ds=100000)
        if use_lru:
            e.map.set_lru(cap, maxd)
        e.set_device_loop(bool(loop))
        pos = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), 1.8])
        s0 = synth.state_from_pose(pos, [0, 0, 0, 1.0])
        for h in (o, e):
            h.set_state(s0)
            h.set_cov(oracle.init_cov())
        yaw, ok = 0.0, True
        for k in range(16):
            step = rng.normal(0, [0.4, 0.2, 0.02]) if rng.random() < 0.85 else rng.normal(0, [3.0, 3.0, 0.05])  # now and then a jump the prior does not cover
            pos = pos + step
            yaw += rng.normal(0, 0.03)
            raw, _ = synth.make_scan(scene, pos, synth.quat_from_rotvec([0, 0, yaw]), seed=int(rng.integers(1, 1 << 30)), n_az=n_az)
            if rng.random() < 0.08:
                raw = raw[: int(rng.integers(0, 6))]  # an (almost) empty scan
            ra = o.process_scan(raw, 0.1 * k)
            rb = e.process_scan(raw, 0.1 * k)
            scans += 1
            so, sg = o.get_state(), e.get_state()
            dp, da = float(np.linalg.norm(so[:3] - sg[:3])), float(synth.quat_angle(so[3:7], sg[3:7]))
            same_map = e.map.stats() == (o.map_num_points, o.map_num_voxels)
            nf = e.map.lru_exact_stats()[1] if use_lru else 0
            if ra != rb or not same_map or dp > 1e-3 or da > 1e-4:
                if nf:  # the quota is below one scan's footprint: counted, the maps may part
                    break
                bad += 1
                ok = False
                print("MISMATCH cfg", c, "scan", k, dict(n_az=n_az, lru=use_lru, cap=cap, maxd=maxd, loop=loop), "rc", ra, rb, "map", e.map.stats(), (o.map_num_points, o.map_num_voxels), "dpos", dp, "drot", da)
                break
            wide = np.eye(6) * (1e-2 if rng.random() < 0.8 else 1.0)
            for h in (o, e):
                P = h.get_cov()
                P[:6, :6] += wide
                h.set_cov(P)
        e.close()
    print("configurations", n_cfg, "scans compared", scans, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
