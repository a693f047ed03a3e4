"""Rows N3 / N4 of SURVEY.md section 8f pinned to the reference's own text: the local-map assembly loop body (localization.cpp:305-312,325-372)
and OverlapDetector::filter / calc_fitness_score (overlap_merge.hpp:213-263), cut out where they lie and compiled in oracle/ref_localmap.cpp.
  * the harness reproduces tests/golden/localmap.npz (recorded from it by tools/make_golden_localmap.py) where /root/reference is mounted;
  * a numpy restatement of both stages agrees with the recorded vectors everywhere (CPU, no reference tree needed) -- the restatement the GPU
    tests used to be checked against is now itself checked against the reference's code.
The device side (lio_localmap_*, lio_ndt_overlap_score) is compared with the same vectors in tests/test_ndt_gpu.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import localmap_cases as lc  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "localmap.npz")


@pytest.fixture(scope="module")
def frames():
    return lc.key_frames()


def select_restated(poses, n_pts, pose, key_frame_distance):
    """localization.cpp:328-349 restated: key frames within 30 m (squared distance in f32, strictly below), nearest first, one skipped when it
    is less than key_frame_distance farther than the last one taken, until the concatenation holds >= 200 000 points"""
    q = np.asarray(pose, np.float32)
    e = np.array(poses, np.float32) - q
    d2 = ((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]).astype(np.float32)
    order = [i for i in np.argsort(d2, kind="stable") if d2[i] < np.float32(900.0)]
    take, acc, total = [], np.float32(0.0), 0
    for i in order:
        dist = np.sqrt(d2[i])
        if total and (dist - acc) < key_frame_distance:
            continue
        acc = dist
        take.append(i)
        total += n_pts[i]
        if total >= 200_000:
            break
    return take, (d2[order[0]] if order else None)


def test_reference_excerpt_reproduces_the_recorded_vectors(frames):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import ref_localmap

    if not ref_localmap.available():
        pytest.skip("oracle/_ref/libref_localmap.so not built (no /root/reference here)")
    g = np.load(GOLD)
    fr, poses = frames
    R = ref_localmap.RefLocalMap(resolution=lc.LEAF, key_frame_distance=lc.KEY_FRAME_DISTANCE)
    for w, p in zip(fr, poses):
        R.add_keyframe(w, p)
    for k, (pose, what) in enumerate(lc.LM_POSES):
        assert R.update(pose) == g["codes"][k], what
        m = R.local_map()
        if g["digests"][k][0] < 0:
            assert m is None, what
        else:
            assert np.array_equal(lc.digest(m), g["digests"][k]), what
    c1, c2, T = lc.overlap_case()
    for r, want in zip(g["ranges"], g["fitness"]):
        assert np.array_equal(np.array(ref_localmap.overlap_fitness(c1, c2, T, r)), want)


def test_restated_selection_and_voxelgrid_equal_the_reference_excerpt(frames, oracle_mod):
    g = np.load(GOLD)
    fr, poses = frames
    n_pts = [len(f) for f in fr]
    last = None
    for k, (pose, what) in enumerate(lc.LM_POSES):
        pose = np.array(pose)
        if last is not None and np.linalg.norm(pose - last) <= 10.0:  # localization.cpp:325-327
            assert g["codes"][k] == 0, what
            continue
        take, d2min = select_restated(poses, n_pts, pose, lc.KEY_FRAME_DISTANCE)
        if not take:
            assert g["codes"][k] == 2, what
            continue
        last = pose  # (lastPose moves whenever the radius search found something, also on the "far" branch)
        if d2min >= 400:
            assert g["codes"][k] == 3, what
            continue
        assert g["codes"][k] == 1, what
        want = oracle_mod.voxel_downsample(np.concatenate([fr[i] for i in take]), lc.LEAF)
        assert np.array_equal(lc.digest(want), g["digests"][k]), what
        assert np.array_equal(want[:64].view(np.uint32), g["heads"][k].view(np.uint32))


def test_restated_fitness_score_equals_the_reference_excerpt():
    g = np.load(GOLD)
    c1, c2, T = lc.overlap_case()
    Tf = T.astype(np.float32)

    def flt(p):
        return (np.sqrt(p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) < np.float32(100.0)) & (p[:, 2] > np.float32(0.5))

    target = c1[flt(c1)]
    tp = np.stack([((Tf[r, 0] * c2[:, 0] + Tf[r, 1] * c2[:, 1]) + Tf[r, 2] * c2[:, 2]) + Tf[r, 3] for r in range(3)], 1)
    tk = tp[flt(tp)]
    best = np.full(len(tk), np.inf, np.float32)
    for a in range(0, len(tk), 100):
        d = tk[a:a + 100, None, :] - target[None, :, :3]
        best[a:a + 100] = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).min(1)
    for r, (score, ratio) in zip(g["ranges"], g["fitness"]):
        inl = best <= np.float32(r)
        if inl.sum() == 0:
            assert score > 1e300 and ratio == 0.0
            continue
        # the reference adds the f32 distances into a double one by one, in point order
        s = 0.0
        for v in best[inl]:
            s += float(v)
        assert score == s / inl.sum() and ratio == inl.sum() / len(tk), r
    assert g["fitness_none"][0] > 1e300 and g["fitness_none"][1] == 0.0
