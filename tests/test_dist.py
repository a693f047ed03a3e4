"""The N > 1 path: sharding helper and the all-gather + fixed-order sum of per-rank normal equations, run as a real
world_size-2 `gloo` job (CPU here; the same code runs over RCCL with backend "nccl")."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(mode, world, outdir, timeout=600, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", **extra_env)
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


@pytest.mark.gpu
def test_batched_joint_registration_equals_single_joint_registrations():
    """lio_batch's joint mode (B scans per launch, the sub-maps' sums folded on the device, ONE [B x 32]-double gather per pass, filter loop on
    the device) against B single lio_engine_joint_register calls (host-driven loop, a hand-over per pass) on the same three sub-maps: same
    pass / search counts, states within the device-vs-host filter bar (1e-9; the two loops share eskf_dev.h, the libm calls differ) -- with
    and without a communicator of one rank, and bit-identical between those two and between repeated calls."""
    sys.path.insert(0, HERE)
    import scenes
    from _dist_worker import make_world
    from lsd_amd import lio, synth

    subs, raw0, state0, true_pos, true_q = make_world(3)
    scene = synth.Scene(half=40.0, n_boxes=12, seed=21)
    # a few key-frame scans from other poses, each with its own prior
    scans = [(raw0, state0, true_pos)]
    for k in range(1, 7):
        pos = np.array([0.3 + 1.5 * k, 0.8 - 0.7 * k, 1.7])
        q = synth.quat_from_rotvec([0, 0, 0.2 + 0.3 * k])
        raw, _ = synth.make_scan(scene, pos, q, seed=230 + k, n_az=300)
        gp, gq = synth.perturb_pose(pos, q, seed=240 + k, max_t=0.15, max_deg=1.0)
        scans.append((raw, synth.state_from_pose(gp, gq), pos))
    P0 = lio.init_cov()
    maps = []
    for sub in subs:
        m = lio.Map(resolution=0.5, stencil=19, max_points=400_000, max_voxels=200_000)
        m.add(sub)
        maps.append(m)
    # the single-scan joint path: engines sharing the three maps
    engs = [lio.Engine(max_raw=1 << 17, max_ds=1 << 16, shared_map=m) for m in maps]
    for e in engs:
        e.set_flags(ekf_inited=True, first_scan=False, first_lidar_time=-10.0)
    engs[0].set_joint(engs[1:], None)
    want = []
    for raw, st, _ in scans:
        for e in engs:
            e.scan.reset()
        rc, s, P = engs[0].joint_register(raw, 1.0, st, P0)
        assert rc == 3
        tm = engs[0].timings()
        want.append((s, tm["n_pass"], tm["n_knn_pass"]))
    jobs = [dict(dptr=scenes.to_device(raw), n=len(raw), t=1.0, state=st, cov=P0) for raw, st, _ in scans]
    outs = []
    for comm in (None, lio.Comm(rank=0, world=1)):
        b = lio.Batch(maps[0], n_slots=3, n_groups=2, max_raw=1 << 17, max_ds=1 << 16, sub_maps=maps[1:], comm=comm)
        rc, res = b.process(jobs)
        assert rc == 0
        rc2, res2 = b.process(jobs[::-1])  # other slots, other histories: independent jobs
        assert rc2 == 0
        for r, r2, (s, n_pass, n_knn), (_, _, pos) in zip(res, res2[::-1], want, scans):
            assert r["rc"] == 3 and (r["n_pass"], r["n_knn_pass"]) == (n_pass, n_knn), (r, n_pass, n_knn)
            assert np.abs(r["state"] - s).max() < 1e-9
            assert np.array_equal(r["state"], r2["state"])
            assert np.linalg.norm(r["state"][:3] - pos) < 0.05
        outs.append(res)
        del b
    for a, c in zip(*outs):
        assert np.array_equal(a["state"], c["state"])


@pytest.mark.gpu
def test_batched_joint_registration_two_ranks_one_gpu():
    """the world > 1 logic of the batched joint mode, for real: two processes on cuda:0, two of four sub-maps each, every round's [B x 32]-double
    records exchanged rank-major through lio_batch_set_gather_hook (gloo: RCCL refuses two ranks on one device), summed in rank order inside
    step_batch.  Both ranks must end with bit-identical states, and those must be the one-process result over all four sub-maps up to the
    order of the f64 additions ((m0 + m1) + (m2 + m3) against ((m0 + m1) + m2) + m3)."""
    sys.path.insert(0, HERE)
    from _dist_worker import batch_scans, make_world, run_batch

    with tempfile.TemporaryDirectory() as td:
        _run("gpu_batch", 2, td)
        r0, r1 = np.load(os.path.join(td, "rank0.npz")), np.load(os.path.join(td, "rank1.npz"))
    assert np.array_equal(r0["states"], r1["states"]) and np.array_equal(r0["passes"], r1["passes"])
    assert np.all(r0["rcs"] == 3) and np.all(r1["rcs"] == 3)
    assert int(r0["calls"]) == int(r1["calls"]) >= 2  # one exchange per round and pass
    one = run_batch(make_world(4)[0])
    for k, (r, (_, _, pos)) in enumerate(zip(one, batch_scans())):
        assert r["rc"] == 3 and (r["n_pass"], r["n_knn_pass"]) == tuple(r0["passes"][k])
        assert np.abs(r["state"] - r0["states"][k]).max() < 1e-9, k
        assert not np.array_equal(r["state"], np.zeros(26)) and np.linalg.norm(r["state"][:3] - pos) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_joint_rounds_divide_the_downsample_among_the_ranks(world):
    """With several ranks a joint round's voxel-grid chain runs ONCE per scan, on the rank that owns the slot, and the downsampled clouds travel in one
    all-gather per round (csrc/p2plane.hip pack_ds_batch / unpack_ds_batch) -- until round 6 every rank downsampled every scan, the part of a round that
    did not shrink with the number of GPUs.  The chain is deterministic: states, return codes and pass counts must be BIT-identical to the form in
    which every rank downsamples everything (LIO_JOINT_SPLIT_DS=0), on every rank.  Six jobs through four slots: the second round leaves the last
    ranks' slots idle; world 3 leaves rank 2 with no slot of its own at all (four slots, two per rank)."""
    outs = {}
    for split in ("0", "1"):
        with tempfile.TemporaryDirectory() as td:
            _run("gpu_batch", world, td, LIO_JOINT_SPLIT_DS=split)
            outs[split] = [dict(np.load(os.path.join(td, f"rank{r}.npz"))) for r in range(world)]
    for r in range(world):
        assert np.all(outs["1"][r]["rcs"] == 3)
        for key in ("states", "rcs", "passes"):
            assert np.array_equal(outs["0"][r][key], outs["1"][r][key]), (r, key)
        assert np.array_equal(outs["1"][r]["states"], outs["1"][0]["states"])
    # the divided form makes exactly one more exchange per round (two rounds here), and that exchange carries clouds, not 32-double records
    assert int(outs["1"][0]["calls"]) == int(outs["0"][0]["calls"]) + 2
    assert int(outs["1"][0]["records"]) > 100 * int(outs["0"][0]["records"])
    # the chunk follows the clouds (1.25 x the largest seen, in steps of 1024 points, at least 4096), not max_ds = 65536; nothing was cut
    n_max = int(outs["1"][0]["n_ds"].max())
    for r in range(world):
        assert int(outs["1"][r]["jobs_rerun"]) == 0 and int(outs["0"][r]["chunk_points"]) == 0
        assert int(outs["1"][r]["chunk_points"]) == max(4096, (n_max + n_max // 4 + 1023) // 1024 * 1024)


@pytest.mark.gpu
def test_a_sort_launched_with_too_few_passes_is_repeated_on_every_rank():
    """The batch launches as many radix passes as the last rounds needed.  A scan whose bounding box needs one more is collected as "again" and registered
    alone with all four -- with the downsample divided among the ranks only the slot's OWNER sees the sort, so the radix bits travel in the chunk's
    header: every rank must take the same decision (a rank that did not would wait in the re-run's collectives alone).  Thirteen jobs, the twelfth with
    two returns three hundred metres away: same states on both ranks, bit for bit the states of the form in which every rank downsamples everything."""
    outs = {}
    for split in ("0", "1"):
        with tempfile.TemporaryDirectory() as td:
            _run("gpu_batch", 2, td, LIO_JOINT_SPLIT_DS=split, LIO_TEST_WIDE="1")
            outs[split] = [dict(np.load(os.path.join(td, f"rank{r}.npz"))) for r in range(2)]
    for r in range(2):
        assert len(outs["1"][r]["rcs"]) == 13 and np.all(outs["1"][r]["rcs"] == 3)
        for key in ("states", "rcs", "passes", "n_ds"):
            assert np.array_equal(outs["0"][r][key], outs["1"][r][key]), (r, key)
        assert np.array_equal(outs["1"][r]["states"], outs["1"][0]["states"])
    # the wide scan was registered twice: one more round than the job list asks for (4 rounds of 4 slots + the re-run) -- the divided form makes one
    # exchange of clouds per round on top of what the other form exchanges
    assert int(outs["1"][0]["calls"]) - int(outs["0"][0]["calls"]) == 5


@pytest.mark.gpu
def test_a_cloud_that_does_not_fit_its_chunk_of_the_all_gather_is_registered_again():
    """The slot chunks of the round's all-gather are sized from the clouds seen so far; a denser scan than any before does not fit.  Forced here with
    chunks of 1500 points (LIO_JOINT_DS_CAP): every cloud is cut on its way to the other rank, the flag travels in the chunk's header, BOTH ranks
    discard the round's result for the job and register it again with full-size chunks -- same states, bit for bit, as without the division."""
    outs = {}
    for name, env in (("whole", dict(LIO_JOINT_SPLIT_DS="0")), ("cut", dict(LIO_JOINT_DS_CAP="1500"))):
        with tempfile.TemporaryDirectory() as td:
            _run("gpu_batch", 2, td, **env)
            outs[name] = [dict(np.load(os.path.join(td, f"rank{r}.npz"))) for r in range(2)]
    assert outs["whole"][0]["n_ds"].min() > 1500
    for r in range(2):
        assert int(outs["cut"][r]["jobs_rerun"]) == 6 and np.all(outs["cut"][r]["rcs"] == 3)
        for key in ("states", "rcs", "passes", "n_ds"):
            assert np.array_equal(outs["whole"][r][key], outs["cut"][r][key]), (r, key)


@pytest.mark.gpu
def test_batched_joint_registration_two_gpus_native_rccl_allgather():
    """The same two-rank joint registration with the NATIVE exchange -- one process per GPU, the C ABI's RCCL communicator, ncclAllGather of the round's
    records on the round's stream (csrc/comm.hip, step_batch<JOINT>) -- instead of the gloo hook that stands in for it where two ranks must share
    one device.  Runs wherever two GPUs are visible (the first multi-GPU box exercises RCCL for real); skipped on the one-GPU boxes of this pool,
    which is why `ncclAllGather` with more than one rank had never executed up to round 6."""
    from lsd_amd import capi

    if capi.lib().lio_device_count() < 2:
        pytest.skip("needs two GPUs: ncclAllGather refuses two ranks on one device (covered there by the gloo gather hook, the test above)")
    sys.path.insert(0, HERE)
    from _dist_worker import batch_scans, make_world, run_batch

    with tempfile.TemporaryDirectory() as td:
        _run("gpu_batch_rccl", 2, td)
        r0, r1 = np.load(os.path.join(td, "rank0.npz")), np.load(os.path.join(td, "rank1.npz"))
    assert np.array_equal(r0["states"], r1["states"]) and np.array_equal(r0["passes"], r1["passes"])
    assert np.all(r0["rcs"] == 3) and np.all(r1["rcs"] == 3)
    assert int(r0["calls"]) == int(r1["calls"]) >= 2
    one = run_batch(make_world(4)[0])
    for k, (r, (_, _, pos)) in enumerate(zip(one, batch_scans())):
        assert r["rc"] == 3 and (r["n_pass"], r["n_knn_pass"]) == tuple(r0["passes"][k])
        assert np.abs(r["state"] - r0["states"][k]).max() < 1e-9, k


@pytest.mark.gpu
def test_batched_joint_registration_degenerate_scene_falls_back_to_the_host_path():
    """open ground: the eigenvalue bound of the GLOBAL sum n n^T does not decide, the six degeneracy sums are needed -- they live on several
    sub-maps, so the batched joint mode hands the scan to the host-driven joint path of the slot's engines: the result IS lio_engine_joint_register's"""
    sys.path.insert(0, HERE)
    import scenes
    from lsd_amd import lio

    case = scenes.degenerate_case("open_ground")
    mp = case["map"]
    halves = [np.ascontiguousarray(mp[mp[:, 0] < 1.0]), np.ascontiguousarray(mp[mp[:, 0] >= -1.0])]
    maps = []
    for sub in halves:
        m = lio.Map(resolution=0.5, stencil=19, max_points=400_000, max_voxels=200_000)
        m.add(sub)
        maps.append(m)
    P0 = lio.init_cov()
    engs = [lio.Engine(max_raw=1 << 17, max_ds=1 << 16, shared_map=m) for m in maps]
    for e in engs:
        e.set_flags(ekf_inited=True, first_scan=False, first_lidar_time=-10.0)
    engs[0].set_joint(engs[1:], None)
    rc, s, P = engs[0].joint_register(case["raw"], 1.0, case["guess"], P0)
    assert rc == 3 and engs[0].is_degenerate
    b = lio.Batch(maps[0], n_slots=2, n_groups=1, max_raw=1 << 17, max_ds=1 << 16, sub_maps=maps[1:])
    rcb, res = b.process([dict(dptr=scenes.to_device(case["raw"]), n=len(case["raw"]), t=1.0, state=case["guess"], cov=P0)] * 3)
    assert rcb == 0
    for r in res:
        assert r["rc"] == 3 and np.array_equal(r["state"], s)


@pytest.mark.parametrize("config,world", [("merge", 2), ("merge", 4), ("metric", 2), ("metric", 8), ("merge", 8)])
def test_bench_multi_gpu_launch_dry_run(config, world):
    """`bench.py --gpus N [--config merge] --dry-run` under torch.distributed.run, as the driver launches it, on the CPU: rendezvous, the sharding
    of sub-maps / scans over the ranks, the RCCL unique id made by rank 0 (librccl loaded lazily, no device needed) and shipped to the others"""
    import json

    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(31500 + (os.getpid() + world) % 2000), os.path.join(root, "bench.py"), "--gpus", str(world), "--config", config, "--dry-run"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1  # rank 0 alone prints
    j = json.loads(line[0])
    assert j["dry_run"] and j["n_gpus"] == world and j["rccl_unique_id_exchanged"] is True and len(j["ranks"]) == world
    # the first 8-GPU lease cannot be rehearsed: one process per GPU, rank r of the node on cuda:r, rank 0 alone prints, no rank runs the single-GPU
    # secondary legs (nobody waits at a barrier for a CPU baseline)
    assert [rk["rank"] for rk in j["ranks"]] == list(range(world)) and all(rk["device"] == f"cuda:{rk['local_rank']}" == f"cuda:{rk['rank']}" for rk in j["ranks"])
    assert [rk["prints_the_line"] for rk in j["ranks"]] == [True] + [False] * (world - 1) and not any(rk["runs_secondary_legs"] for rk in j["ranks"])
    if config == "merge":
        owned = sorted(k for rk in j["ranks"] for k in rk["sub_maps"])
        assert owned == list(range(8)) and all(len(rk["sub_maps"]) == 8 // world for rk in j["ranks"])
        assert len({rk["first_guess_digest"] for rk in j["ranks"]}) == 1  # every rank registers the same key frames from the same priors
    else:
        seeds = [tuple(rk["scan_seeds"]) for rk in j["ranks"]]
        assert len(set(seeds)) == world  # weak scaling: every rank its own scans


@pytest.mark.parametrize("config", ["metric", "merge"])
def test_bench_spawns_its_own_ranks(config):
    """`python bench.py --gpus 2` WITHOUT a launcher -- the shape of the driver's BENCH command -- must not time one GPU and say so: it re-executes
    itself under torch.distributed.run with two ranks, and a launcher that started another number of ranks than --gpus is
    refused"""
    import json

    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", config, "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    j = json.loads(line[0])
    assert j["dry_run"] and j["n_gpus"] == 2 and len(j["ranks"]) == 2 and j["rccl_unique_id_exchanged"] is True
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "refusing" in (bad.stderr + bad.stdout)


def test_rccl_probe_runs_in_child_processes_and_never_raises():
    """bench.py's RCCL probe (N > 1 only: the communicator of the C ABI brought up beside the headline) lives in child processes: `--config rcclprobe
    --probe uid` hands out a fresh unique id, and the parent's helper turns every failure of a child -- here: no GPU for the communicator itself --
    into an `error` entry of the line instead of an exception or a hang"""
    import importlib.util
    import json

    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "rcclprobe", "--probe", "uid"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    uid = r.stdout.strip().splitlines()[-1]
    assert len(uid) == 256 and int(uid, 16) >= 0

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class OneRank:  # torch.distributed's broadcast_object_list for a world of one
        @staticmethod
        def broadcast_object_list(box, src=0):
            return None

    out = bench.rccl_probe(OneRank, rank=0, world=1, local_rank=0, iters=2, timeout_s=240.0)
    assert isinstance(out, dict)
    if "error" in out:  # (no GPU in this container: the child reports, the parent carries it on)
        assert isinstance(out["error"], str) and out["error"]
    else:
        assert out["rccl_ranks"] == 1 and out["gathered_in_rank_order"]
    json.dumps(out)
