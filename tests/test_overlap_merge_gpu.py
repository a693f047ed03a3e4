"""The map-merge matching flow end to end through the C ABI: OverlapDetector::matching (slam/localization/include/overlap_merge.hpp:151-211)
= per candidate overlap pre-check + coarse NDT + fitness, fine GICP (0.5 m / 0.001) against the best candidate accumulated with its linked
frames, final fitness -- the device flow (lio.OverlapMatcher over lio_ndt_* / lio_gicp_*) next to the same flow composed from the
reference's own matchers (fast_gicp::NDTCuda compiled for gfx950, fast_gicp::FastGICP on the CPU) and brute-force fitness scores."""
import os
import sys

import numpy as np
import pytest
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import gicp_cases  # noqa: E402
import ref_gicp  # noqa: E402
import ref_ndt_cuda  # noqa: E402
from lsd_amd import lio, synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _keyframe(sc, pos, yaw, seed, drift=(0.0, 0.0, 0.0)):
    q = synth.quat_from_rotvec([0, 0, yaw])
    raw, _ = synth.make_scan(sc, np.asarray(pos, float), q, seed=seed, n_az=900, n_beams=32, max_range=60.0)
    pts = gicp_cases._thin(raw[:, :4].astype(np.float32), 0.4)
    odom = gicp_cases._pose(np.asarray(pos, float) + np.asarray(drift), q)
    return pts, odom


def _fitness(cloud1, cloud2, relpose, max_range):
    """calc_fitness_score of overlap_merge.hpp:214-263 (exact nearest neighbours; distances squared, f64 here)"""
    t = lio.overlap_filter(cloud1)
    s = lio.overlap_filter(lio.transform_cloud_f32(cloud2, relpose))
    d, _ = cKDTree(t[:, :3].astype(np.float64)).query(s[:, :3].astype(np.float64))
    d2 = d * d
    inl = d2 <= max_range
    return (d2[inl].mean(), inl.sum() / len(s)) if inl.any() else (np.inf, 0.0)


def _pcl_fitness(target, source, T, max_range):
    s = lio.transform_cloud_f32(source, T)
    d, _ = cKDTree(target[:, :3].astype(np.float64)).query(s[:, :3].astype(np.float64))
    d2 = d * d
    return d2[d2 <= max_range].mean()


@pytest.mark.skipif(not (ref_gicp.available() and ref_ndt_cuda.available()), reason="oracle/_ref harnesses not built")
def test_overlap_matching_flow_vs_reference_matchers():
    sc = synth.Scene(half=50.0, n_boxes=30, seed=12)
    new_pts, new_odom = _keyframe(sc, [2.0, 1.0, 1.8], 0.3, 1)
    cands = [_keyframe(sc, [3.2, 0.4, 1.8], 0.45, 2, drift=(0.4, -0.3, 0.05)),   # overlapping, drifted odometry
             _keyframe(sc, [-30.0, 28.0, 1.8], 2.5, 3)]                            # elsewhere
    linked = [(0, *_keyframe(sc, [4.5, 0.0, 1.8], 0.5, 4, drift=(0.4, -0.3, 0.05)))]
    m = lio.OverlapMatcher(max_points=200_000)
    got = m.matching(new_pts, new_odom, cands, linked)
    assert got is not None and got["best"] == 0

    # the same flow from the reference's matchers
    best_score, best, rel, tried = np.inf, None, None, []
    reg = ref_ndt_cuda.NdtCudaRegistration(resolution=1.0, search_method=7)
    reg.set_target(new_pts)
    for ci, (pts, odom) in enumerate(cands):
        guess = (np.linalg.inv(new_odom) @ odom).astype(np.float32)
        if _fitness(new_pts, pts, guess, 1.0)[1] < 0.2:
            continue
        reg.set_source(pts)
        T, conv, _ = reg.align(guess)
        if not conv:
            continue
        score = _pcl_fitness(new_pts, pts, T, 25.0)
        tried.append(ci)
        if score <= best_score:
            best_score, best, rel = score, ci, T.astype(np.float32)
    assert best == 0
    coarse = [c for c in got["coarse"] if c["candidate"] == 0][0]
    assert [c["candidate"] for c in got["coarse"] if "skipped" not in c] == tried   # the same candidates pass the pre-check and converge
    assert np.abs(coarse["T"][:3, 3] - rel[:3, 3]).max() < 5e-3     # the NDT object of the reference is itself only repeatable to ~1e-3 (test_ndt_vs_ref_cuda)
    assert abs(coarse["score"] - best_score) < 0.02 * best_score
    # overlap_merge.hpp:190-194: the connected frame moved by the f64 `relative` (pcl::transformPointCloud with a Matrix4d: double arithmetic, cast
    # to float) -- restated here independently of the product's helper
    Mrel = np.linalg.inv(cands[0][1]) @ linked[0][2]
    moved = linked[0][1].copy()
    moved[:, :3] = (linked[0][1][:, :3].astype(np.float64) @ Mrel[:3, :3].T + Mrel[:3, 3]).astype(np.float32)
    assert np.abs(moved[:, :3] - lio.transform_cloud_f64(linked[0][1], Mrel)[:, :3]).max() <= 4e-6 * np.abs(moved[:, :3]).max()  # (summation order: an ulp at most)
    accum = np.concatenate([cands[0][0], moved])
    fine = ref_gicp.RefGicp(k=20, max_corr_dist=0.5, transformation_epsilon=0.001, num_threads=4)
    fine.set_target(accum)
    fine.set_source(new_pts)
    Tr, it_r, conv_r = fine.align(np.linalg.inv(rel.astype(np.float64)).astype(np.float32))
    assert conv_r
    score_r, _ = _fitness(accum, new_pts, Tr, 25.0)
    # end to end: both flows end at the same relative pose up to the fine matcher's own stopping tolerance (0.001 m, 0.01 deg)
    assert np.abs(got["relative_pose"][:3, 3] - Tr[:3, 3]).max() < 2e-3
    assert np.abs(got["relative_pose"][:3, :3] - Tr[:3, :3]).max() < 4e-4
    assert abs(got["score"] - score_r) < 0.02 * score_r and got["score"] < 1.5
    truth = np.linalg.inv(gicp_cases._pose([3.2, 0.4, 1.8], synth.quat_from_rotvec([0, 0, 0.45]))) @ gicp_cases._pose([2.0, 1.0, 1.8], synth.quat_from_rotvec([0, 0, 0.3]))
    assert np.abs(got["relative_pose"][:3, 3] - truth[:3, 3]).max() < 0.03
    # the fine matcher alone from the reference's own coarse result: BASELINE.json tolerance
    m.gicp.set_target(accum)
    m.gicp.set_source(new_pts)
    Td, conv_d, it_d = m.gicp.align(np.linalg.inv(rel.astype(np.float64)).astype(np.float32).astype(np.float64), max_corr_dist=0.5, transformation_epsilon=0.001)
    assert conv_d == conv_r and it_d == it_r
    assert np.abs(Td[:3, 3] - Tr[:3, 3]).max() < 1e-4 and np.abs(Td[:3, :3] - Tr[:3, :3]).max() < 1e-5
    m.close()
