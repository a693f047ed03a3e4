"""The N > 1 path: sharding helper and the all-gather + fixed-order sum of per-rank normal equations, run as a real
world_size-2 `gloo` job (CPU here; the same code runs over RCCL with backend "nccl")."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(mode, world, outdir, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + os.getpid() % 2000), os.path.join(HERE, "_dist_worker.py"), mode, outdir]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]


def test_shard_range_partitions_exactly():
    from lsd_amd import dist as ldist

    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            pieces = [ldist.shard_range(n, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
            sizes = [hi - lo for lo, hi in pieces]
            assert max(sizes) - min(sizes) <= 1


def test_allgather_normal_equations_gloo_world2():
    from lsd_amd import dist as ldist

    with tempfile.TemporaryDirectory() as td:
        _run("cpu", 2, td)
        r0, r1 = np.load(os.path.join(td, "rank0.npz")), np.load(os.path.join(td, "rank1.npz"))
    # both ranks hold the identical reduced record, and it is the fixed-order sum of the two local ones, bit for bit
    assert np.array_equal(r0["reduced"], r1["reduced"])
    want = ldist.sequential_sum([r0["local"], r1["local"]])
    assert np.array_equal(r0["reduced"], want)
    assert r0["local"][28] > 100 and r1["local"][28] > 100  # both sub-maps contributed correspondences
    assert not np.array_equal(r0["local"], r1["local"])


@pytest.mark.gpu
def test_joint_registration_two_ranks_one_gpu():
    """full engine path with the reduce hook: two processes (gloo collective, both on cuda:0), each holding one sub-map.
    Expected: identical states on both ranks, equal to a one-process emulation that visits the sub-maps in rank order."""
    sys.path.insert(0, HERE)
    from _dist_worker import make_world
    from lsd_amd import dist as ldist, lio, synth

    with tempfile.TemporaryDirectory() as td:
        _run("gpu", 2, td)
        r0, r1 = np.load(os.path.join(td, "rank0.npz")), np.load(os.path.join(td, "rank1.npz"))
    assert int(r0["rc"]) == 3 and int(r1["rc"]) == 3 and int(r0["calls"]) >= 2
    assert np.array_equal(r0["state"], r1["state"]) and np.array_equal(r0["cov"], r1["cov"])
    # one-process emulation: engine A (sub-map 0) drives the filter; its hook adds what engine B (sub-map 1) sees at
    # the same iterate, in rank order -- exactly what the all-gather + fixed-order sum delivers to every rank
    subs, raw, state, true_pos, true_q = make_world(2)
    A = lio.Engine(stencil=19, max_points=400_000, max_voxels=200_000, max_raw=1 << 17, max_ds=1 << 16)
    B = lio.Engine(stencil=19, max_points=400_000, max_voxels=200_000, max_raw=1 << 17, max_ds=1 << 16)
    for e, sub in ((A, subs[0]), (B, subs[1])):
        e.map_add(sub)
        e.set_static_map(True)
        e.set_flags(ekf_inited=True, first_scan=False, first_lidar_time=-10.0)
    B.scan.upload(raw)
    B.scan.voxel_downsample(0.5)
    B.scan.set_degeneracy_mode(2)
    st = {"knn": 0, "nnT": None, "calls": 0}

    def hook(buf):
        st["calls"] += 1
        if len(buf) == 29:
            k = A.timings()["n_knn_pass"]  # already counts the pass being reduced
            redo = k > st["knn"]
            st["knn"] = k
            lin = lio.linearize(B.map, B.scan, A.get_state(), redo_knn=redo)
            other = ldist.pack_normal_eq(lin["JtJ"], lin["Jtr"], lin["sum_abs_res"], lin["n_eff"])
            buf[:] = ldist.sequential_sum([buf.copy(), other])
            J = np.zeros((6, 6))
            J[np.triu_indices(6)] = buf[:21]
            st["nnT"] = (J + J.T - np.diag(np.diag(J)))[:3, :3]
        else:
            _, V = np.linalg.eigh(st["nnT"])
            buf[:] = ldist.sequential_sum([buf.copy(), np.concatenate(B.scan.degeneracy(V))])

    A.set_state(state)
    A.set_cov(lio.init_cov())
    A.set_reduce_hook(hook)
    assert A.process_scan(raw, 1.0) == 3
    assert st["calls"] == int(r0["calls"])
    assert np.array_equal(A.get_state(), r0["state"]) and np.array_equal(A.get_cov(), r0["cov"])
    # the same natively (lio_engine_set_joint, no Python in the loop): engine A2 drives, engine B2 is the other local sub-map; and once more
    # through a communicator of one rank (lio_comm_*: the gather is a device copy + the rank-order sum kernel) -- bit for bit the same
    for use_comm in (False, True):
        A2 = lio.Engine(stencil=19, max_points=400_000, max_voxels=200_000, max_raw=1 << 17, max_ds=1 << 16)
        B2 = lio.Engine(stencil=19, max_points=400_000, max_voxels=200_000, max_raw=1 << 17, max_ds=1 << 16)
        for e, sub in ((A2, subs[0]), (B2, subs[1])):
            e.map_add(sub)
            e.set_static_map(True)
            e.set_flags(ekf_inited=True, first_scan=False, first_lidar_time=-10.0)
        comm = lio.Comm(rank=0, world=1) if use_comm else None
        A2.set_joint([B2], comm)
        rc, s2, P2 = A2.joint_register(raw, 1.0, state, lio.init_cov())
        assert rc == 3 and np.array_equal(s2, r0["state"]) and np.array_equal(P2, r0["cov"]), use_comm
        if comm is not None:
            n_coll, t_us = comm.stats()
            assert n_coll == int(r0["calls"]) and t_us > 0
    # and the joint solve registers the scan: neither sub-map alone covers it, together they do
    assert np.linalg.norm(r0["state"][:3] - true_pos) < 0.03
    assert synth.quat_angle(r0["state"][3:7], true_q) < 3e-3


@pytest.mark.gpu
def test_native_allgather_device_buffers():
    """lio_allgather_normal_eq with device pointers in and out (a world of one: the gather is a copy, the sum kernel runs)"""
    import ctypes as C

    sys.path.insert(0, HERE)
    import scenes
    from lsd_amd import lio

    comm = lio.Comm(rank=0, world=1)
    local = np.arange(32, dtype=np.float64) * 1.5 - 7.0
    d_local, d_g, d_s = scenes.to_device(local), scenes.to_device(np.zeros(32)), scenes.to_device(np.zeros(32))
    comm.allgather(d_local, d_g, d_s)
    hip_rt = C.CDLL("libamdhip64.so")
    hip_rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip_rt.hipDeviceSynchronize()
    out = np.zeros(32)
    assert hip_rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(d_s), 256, 2) == 0
    assert np.array_equal(out, local)
    g = np.zeros(32)
    assert hip_rt.hipMemcpy(g.ctypes.data_as(C.c_void_p), C.c_void_p(d_g), 256, 2) == 0
    assert np.array_equal(g, local)
