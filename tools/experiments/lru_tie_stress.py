"""ties x LRU: lattice points (exact f32 distance ties everywhere, one intensity per point) fed along courses that keep coming back with a quota of a
few batches' footprints -- voxels grow, move, are dropped and re-created inside a batch; the push_back ranks (pool_seq) must travel with the points:
the five neighbours of lattice queries, intensities included, against the oracle's sequential list + its literal std::nth_element selection"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle")]
import oracle
from lsd_amd import lio


def lattice(rng, n, cx, half=6.0, step=0.0625):
    ijk = np.stack([rng.integers(int((cx - half) / step), int((cx + half) / step) + 1, n), rng.integers(-int(half / step), int(half / step) + 1, n),
                    rng.integers(-8, 9, n)], 1)
    return ijk * step


def main(n_cfg=12):
    bad = checked = recreated = 0
    uid = 1.0
    for c in range(n_cfg):
        rng = np.random.default_rng(300 + c)
        cap = int(rng.choice([1500, 3000, 5000]))
        maxd = float(rng.choice([0.0, 2.0]))
        st = int(rng.choice([19, 7, 27]))
        m = lio.Map(resolution=0.5, stencil=st, max_points=600_000, max_voxels=40000)
        if os.environ.get("NO_LRU"):
            cap, maxd = 1 << 40, 100.0
        else:
            m.set_lru(cap, maxd)
        o = oracle.IVox(res=0.5, stencil=st, capacity=cap, max_distance=maxd)
        travel = 0.0
        for b in range(18):
            cx = [(-1) ** b * (2.0 + 0.8 * b), [-14.0, 0.0, 14.0][b % 3]][c % 2]
            travel += 3.0
            xyz = lattice(rng, 4000, cx)
            batch = np.concatenate([xyz, uid + np.arange(len(xyz))[:, None]], 1).astype(np.float32)
            uid += len(xyz)
            m.add(batch, travel=travel)
            o.add(batch, travel=travel)
            if not os.environ.get("NO_LRU") and m.lru_exact_stats()[1]:
                break
            assert m.stats() == (o.num_points, o.num_voxels), (c, b, m.stats(), o.num_points, o.num_voxels)
            if b % 3 == 2:
                q = np.concatenate([lattice(rng, 1500, cx, half=8.0) + 0.03125 * (b % 2), np.zeros((1500, 1))], 1).astype(np.float32)
                for mode in (1, 2):
                    m.set_tie_mode(mode)
                    got, cnt = m.knn(q)
                    got_b, cnt_b = m.knn(q)
                    if not (np.array_equal(got.view(np.uint32), got_b.view(np.uint32)) and np.array_equal(cnt, cnt_b)):
                        print("NOT REPEATABLE cfg", c, "batch", b, "mode", mode, int(np.any(got.view(np.uint32).reshape(len(q), -1) != got_b.view(np.uint32).reshape(len(q), -1), axis=1).sum()))
                    if mode == 1:
                        want, wcnt, _ = o.knn(q)
                    else:
                        want, wcnt = o.knn_as_reference(q)
                    ok = np.array_equal(cnt, wcnt)
                    if ok and mode == 2:
                        ok = np.array_equal(got.view(np.uint32), want.view(np.uint32))
                    elif ok:
                        ok = np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32)) and np.array_equal(np.sort(got[..., 3], 1), np.sort(want[..., 3], 1))
                    checked += 1
                    if not ok:
                        bad += 1
                        print("MISMATCH cfg", c, "batch", b, "tie mode", mode, dict(cap=cap, maxd=maxd, stencil=st))
                        if np.array_equal(cnt, wcnt):
                            rows = np.flatnonzero(np.any(got.view(np.uint32).reshape(len(q), -1) != want.view(np.uint32).reshape(len(q), -1), axis=1))
                            srt = np.flatnonzero(np.any(np.sort(got[..., 3], 1) != np.sort(want[..., 3], 1), axis=1))
                            print("  rows differing bitwise", len(rows), "rows with other point sets", len(srt), "tie stats", m.tie_stats())
                            for r_ in srt[:3]:
                                print("  q", q[r_], "cnt", cnt[r_]); print("   got", got[r_]); print("   want", want[r_])
                                t0 = m.tie_stats()
                                c0 = m.knn_candidates
                                one, _ = m.knn(q[r_:r_ + 1].copy())
                                c1 = m.knn_candidates
                                dump = o.dump()
                                rnd = lambda v: np.sign(v) * np.floor(np.abs(v) + 0.5)  # (half away from zero, as pos2grid)
                                kq = rnd(q[r_, :3].astype(np.float32) * np.float32(2.0))
                                kd_ = rnd(dump[:, :3].astype(np.float32) * np.float32(2.0)) - kq
                                if st == 27:
                                    inst = np.all(np.abs(kd_) <= 1, axis=1)
                                elif st == 19:
                                    inst = np.all(np.abs(kd_) <= 1, axis=1) & (np.abs(kd_).sum(1) <= 2)
                                else:
                                    inst = np.abs(kd_).sum(1) <= 1
                                print("   stencil residents: device", c1 - c0, "oracle", int(inst.sum()))
                                np.savez(os.path.join(ROOT, "gpurun_out", f"tie_case_{c}_{b}.npz"), q=q[r_], dump=m.dump(), st=st, got=got[r_], want=want[r_])
                                t1 = m.tie_stats()
                                m.set_tie_mode(0); can, _ = m.knn(q[r_:r_ + 1].copy()); m.set_tie_mode(2); two, _ = m.knn(q[r_:r_ + 1].copy()); m.set_tie_mode(mode)
                                print("   alone (mode 1) ids", np.sort(one[0, :, 3]), "boundary ties counted for it", t1[0] - t0[0])
                                print("   alone mode 0 ids", np.sort(can[0, :, 3]), "mode 2 ids", np.sort(two[0, :, 3]), "want ids", np.sort(want[r_][:, 3]))
                                # distances of the got / want lists
                                d = lambda a: np.sum((a[:, :3].astype(np.float64) - q[r_, :3]) ** 2, 1)
                                print("   d2 got", d(got[r_]), "d2 want", d(want[r_]))
                        else:
                            print("  counts differ at", np.flatnonzero(cnt != wcnt)[:10])
                m.set_tie_mode(1)
        recreated += 0 if os.environ.get("NO_LRU") else m.lru_exact_stats()[0]
    print("configurations", n_cfg, "query sets compared", checked, "mismatches", bad, "voxels re-created", recreated)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 12) else 0)
