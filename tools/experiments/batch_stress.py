"""random job lists through the static batch (lio_batch_process) against the same jobs one by one through an engine on the same map: lists of 1 ... 70
jobs (rounds that are not full, several rounds in flight), clouds of 0 ... 60 000 points (empty and five-point scans, sparse scans that take the dense
branch of the filter on the host), clouds in device memory and in pinned host memory, wide bounding boxes (a sort launched with too few passes is
repeated), tight and wide priors -- return codes, downsampled sizes and pass counts equal, states within 1e-9 (the engine's passes run on the host), a
second call bit for bit"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import scenes
from lsd_amd import lio, synth


def main(n_cfg=8, seed0=0):
    bad = jobs_n = 0
    for c in range(n_cfg):
        rng = np.random.default_rng(seed0 * 9973 + c)
        scene = synth.Scene(half=60.0, n_boxes=int(rng.choice([6, 25])), seed=int(rng.integers(1, 1000)))
        mp = scene.sample_surface(int(rng.choice([20_000, 200_000, 800_000])), seed=int(rng.integers(1, 1000)), sigma=0.01)
        stc = int(rng.choice([19, 27, 7]))
        the_map = lio.Map(resolution=0.5, stencil=stc, max_points=1_000_000, max_voxels=300_000)
        the_map.add(mp)
        bad0 = bad
        info = dict(n_map=len(mp), stencil=stc)
        n_slots, n_groups = int(rng.choice([1, 3, 8, 16])), int(rng.choice([1, 2, 4]))
        if os.environ.get('SLOTS'):
            n_slots = int(os.environ['SLOTS'])
        if os.environ.get('GROUPS'):
            n_groups = int(os.environ['GROUPS'])
        b = lio.Batch(the_map, n_slots=n_slots, n_groups=n_groups, max_raw=1 << 16, max_ds=30000)
        e = lio.Engine(max_raw=1 << 16, max_ds=30000, shared_map=the_map)
        e.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=1e6)  # (in the future: the engine must not switch the shared map to NEARBY18, laserMapping.cpp:1241-1243, under the batch)
        P0 = lio.init_cov()
        jobs, keep = [], []
        for j in range(int(rng.integers(1, 71))):
            pos = np.array([rng.uniform(-30, 30), rng.uniform(-30, 30), 1.8])
            q = synth.quat_from_rotvec([0, 0, rng.uniform(-3, 3)])
            raw, _ = synth.make_scan(scene, pos, q, seed=int(rng.integers(1, 1 << 30)), n_az=int(rng.choice([4, 30, 200, 800])))
            u = rng.random()
            if u < 0.06 and not os.environ.get('NO_TINY'):
                raw = raw[: int(rng.integers(0, 6))]
            elif u < 0.12 and len(raw) > 2 and not os.environ.get('NO_WIDE'):
                raw = raw.copy()
                raw[0, :3] = [300.0, -250.0, 5.0]
                raw[1, :3] = [-280.0, 290.0, -3.0]
            raw = raw[:60000]
            gp, gq = synth.perturb_pose(pos, q, seed=int(rng.integers(1, 1 << 30)), max_t=float(rng.choice([0.05, 0.3])), max_deg=float(rng.choice([0.5, 2.0])))
            st = synth.state_from_pose(gp, gq)
            P = P0 * (1.0 if os.environ.get('NO_WIDEP') else float(rng.choice([1.0, 1.0, 100.0])))
            host = rng.random() < 0.3 and len(raw) > 0 and not os.environ.get('NO_HOST')
            if host:
                pc = lio.PinnedCloud(raw)
                keep.append(pc)
                jobs.append(dict(dptr=pc.ptr, n=pc.n, t=1.0 + 0.1 * j, state=st, cov=P, flags=lio.JOB_HOST_RAW, raw=raw))
            else:
                jobs.append(dict(dptr=scenes.to_device(raw) if len(raw) else 0, n=len(raw), t=1.0 + 0.1 * j, state=st, cov=P, raw=raw))
        plain = [{k: v for k, v in jb.items() if k != "raw"} for jb in jobs]
        rc, res = b.process(plain)
        rc1, res1 = lio.process_batch([e], plain)  # the same jobs one by one through the engine (host-driven passes; nothing enters the shared map)
        if rc != rc1:
            bad += 1
            print("MISMATCH cfg", c, "call return codes", rc, rc1)
        for j, (jb, r, r1) in enumerate(zip(jobs, res, res1)):
            jobs_n += 1
            ok = (r["rc"], r["n_ds"]) == (r1["rc"], r1["n_ds"])
            if ok and r["rc"] == 3:
                ok = (r["n_pass"], r["n_knn_pass"]) == (r1["n_pass"], r1["n_knn_pass"]) and float(np.abs(r["state"] - r1["state"]).max()) < 1e-9
            if not ok:
                bad += 1
                d = float(np.abs(r["state"] - r1["state"]).max()) if r["rc"] == 3 and r1["rc"] == 3 else None
                print("MISMATCH cfg", c, "job", j, dict(slots=n_slots, groups=n_groups, n=len(jb["raw"]), host="flags" in jb), "rc", r["rc"], r1["rc"], "n_ds", r["n_ds"], r1["n_ds"],
                      "passes", (r["n_pass"], r["n_knn_pass"]), (r1["n_pass"], r1["n_knn_pass"]), "max |dstate|", d)
        # a second call and another geometry give the first call's bits
        rc2, res2 = b.process(plain)
        for j, (r, r2) in enumerate(zip(res, res2)):
            if (r["rc"], r["n_ds"]) != (r2["rc"], r2["n_ds"]) or (r["rc"] == 3 and not np.array_equal(r["state"], r2["state"])):
                bad += 1
                print("NOT REPEATABLE cfg", c, "job", j)
        print('cfg', c, info, 'slots', n_slots, 'groups', n_groups, 'jobs', len(jobs), 'bad', bad - bad0)
        b.close() if hasattr(b, "close") else None
        e.close()
    print("configurations", n_cfg, "jobs compared", jobs_n, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
