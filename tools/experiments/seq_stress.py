"""random session schedules through the sequence batch (lio_batch_create_sequences) against per-session engines with the device loop on: sessions that start
in different rounds, pause for rounds, get empty scans, scans of different sizes, with and without the LRU list (quotas of a few scans' footprints) -- the
batch must return the engine's bits: return codes, states, covariances, map contents, eviction and re-creation counts"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import scenes
from lsd_amd import capi, lio, synth


def rows(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 4)
    return a[np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))]


def next_prior(res, add):
    P = res["cov"].copy()
    P[:6, :6] += np.eye(6) * add
    return res["state"].copy(), P


def main(n_cfg=6, seed0=0):
    bad = compared = 0
    for c in range(n_cfg):
        rng = np.random.default_rng(seed0 * 4099 + c)
        scene = synth.Scene(half=60.0, n_boxes=int(rng.choice([8, 25])), seed=int(rng.integers(1, 1000)))
        n_slots, n_groups = int(rng.choice([2, 3, 5])), int(rng.choice([1, 2]))
        n_sess = n_slots * n_groups
        n_scans = 12
        lru = rng.random() < 0.6
        cap, maxd = int(rng.choice([4000, 9000])), float(rng.choice([0.5, 3.0]))
        kw = dict(resolution=0.5, stencil=75, max_points=600_000, max_voxels=40_000 if lru else 100_000, max_raw=1 << 17, max_ds=60000)
        P0 = lio.init_cov()
        plans, sched = [], []
        for s in range(n_sess):
            start = np.array([rng.uniform(-40, 40), rng.uniform(-40, 40), 1.8])
            heading = rng.uniform(0, 2 * np.pi)
            step = rng.uniform(0.2, 1.0) * np.array([np.cos(heading), np.sin(heading), 0.0])
            n_az = int(rng.choice([40, 300, 700]))
            scans = []
            for k in range(n_scans):
                back = k if k < 8 else 14 - k  # the drive turns round: the list's back is revisited
                pos = start + back * step
                q = synth.quat_mul(synth.quat_from_rotvec([0, 0, heading]), synth.quat_from_rotvec([0, 0, 0.02 * k]))
                raw, _ = synth.make_scan(scene, pos, q, seed=int(rng.integers(1, 1 << 30)), n_az=n_az, max_range=40.0)
                n = 0 if rng.random() < 0.08 else len(raw)
                scans.append(dict(dptr=scenes.to_device(raw), n=n, t=3.0 * s + 0.1 * k))
            plans.append((scans, synth.state_from_pose(start, synth.quat_from_rotvec([0, 0, heading]))))
            # the rounds in which the session's scans arrive: a start offset, pauses in between
            r, when = int(rng.integers(0, 5)), []
            for k in range(n_scans):
                when.append(r)
                r += 1 + (int(rng.integers(1, 4)) if rng.random() < 0.15 else 0)
            sched.append(when)
        add = float(rng.choice([1e-2, 1e-1]))
        solo = []
        for s in range(n_sess):
            e = lio.Engine(**kw)
            e.set_device_loop(True)
            if lru:
                e.map.set_lru(cap, maxd)
            st, P = plans[s][1].copy(), P0.copy()
            out = []
            for sc in plans[s][0]:
                e.set_state(st)
                e.set_cov(P)
                rc = e.process_scan_device(sc["dptr"], sc["n"], sc["t"])
                res = dict(rc=rc, state=e.get_state(), cov=e.get_cov())
                out.append(res)
                if rc == 3:
                    st, P = next_prior(res, add)
            e.flush()
            solo.append(dict(out=out, stats=e.map.stats(), dump=rows(e.map.dump()), lru=e.map.lru_stats()[0] if lru else 0, rec=e.map.lru_exact_stats() if lru else (0, 0)))
            e.close()
        b = lio.SequenceBatch(n_slots=n_slots, n_groups=n_groups, **kw)
        if lru:
            for s in range(n_sess):
                b.engine(s).map.set_lru(cap, maxd)
        priors = [(plans[s][1].copy(), P0.copy()) for s in range(n_sess)]
        got = [[] for _ in range(n_sess)]
        nxt = [0] * n_sess
        for r in range(max(w[-1] for w in sched) + 1):
            jobs = []
            for s in range(n_sess):
                k = nxt[s]
                if k < n_scans and sched[s][k] == r:
                    sc = plans[s][0][k]
                    jobs.append(dict(dptr=sc["dptr"], n=sc["n"], t=sc["t"], state=priors[s][0], cov=priors[s][1]))
                    nxt[s] += 1
                else:
                    jobs.append(None)
            rc, res = b.step(jobs)
            if rc != 0:
                print("STEP FAILED cfg", c, "round", r, rc, capi.lib().lio_last_error().decode()[:200])
                bad += 1
                break
            for s in range(n_sess):
                if res[s] is not None:
                    got[s].append(res[s])
                    if res[s]["rc"] == 3:
                        priors[s] = next_prior(res[s], add)
        for s in range(n_sess):
            compared += 1
            ok = len(got[s]) == n_scans and all(a["rc"] == g["rc"] and (a["rc"] != 3 or (np.array_equal(a["state"], g["state"]) and np.array_equal(a["cov"], g["cov"]))) for a, g in zip(solo[s]["out"], got[s]))
            e = b.engine(s)
            ok = ok and e.map.stats() == solo[s]["stats"] and np.array_equal(rows(e.map.dump()), solo[s]["dump"])
            if lru:
                ok = ok and e.map.lru_stats()[0] == solo[s]["lru"] and e.map.lru_exact_stats() == solo[s]["rec"]
            if not ok:
                bad += 1
                print("MISMATCH cfg", c, "session", s, dict(slots=n_slots, groups=n_groups, lru=lru, cap=cap, maxd=maxd), [g["rc"] for g in got[s]], [a["rc"] for a in solo[s]["out"]], e.map.stats(), solo[s]["stats"])
        b.close()
    print("configurations", n_cfg, "sessions compared", compared, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
