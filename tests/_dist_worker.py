"""worker for tests/test_dist.py: one rank of a world_size-N gloo job.  argv: mode outdir
mode "cpu": per-rank normal equations from the CPU oracle on this rank's sub-map, reduced with NormalEqAllGather
mode "gpu": a full joint registration with lio.Engine + reduce hook on cuda:0 (collective over gloo)
mode "gpu_batch": the BATCHED joint registration (lio_batch_create_joint) with this rank's share of four sub-maps on cuda:0, the [B x 32]-double
                  records of a round all-gathered over gloo through lio_batch_set_gather_hook (RCCL refuses two ranks on one device)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def make_world(world):
    from lsd_amd import synth

    scene = synth.Scene(half=40.0, n_boxes=12, seed=21)
    full = scene.sample_surface(120_000, seed=22, sigma=0.01)
    # overlapping sub-maps: rank r owns the points whose x falls in its slab (+2 m halo)
    edges = np.linspace(-40, 40, world + 1)
    subs = [full[(full[:, 0] >= edges[r] - 2) & (full[:, 0] < edges[r + 1] + 2)] for r in range(world)]
    true_pos, true_q = np.array([0.3, 0.8, 1.7]), synth.quat_from_rotvec([0, 0, 0.2])
    raw, _ = synth.make_scan(scene, true_pos, true_q, seed=23, n_az=300)
    gp, gq = synth.perturb_pose(true_pos, true_q, seed=24, max_t=0.15, max_deg=1.0)
    return subs, raw, synth.state_from_pose(gp, gq), true_pos, true_q


def batch_scans():
    """key-frame scans with their priors, the same on every rank (what test_dist's batched tests register)"""
    from lsd_amd import synth

    scene = synth.Scene(half=40.0, n_boxes=12, seed=21)
    out = []
    for k in range(6):
        pos = np.array([0.3 + 1.5 * k, 0.8 - 0.7 * k, 1.7])
        q = synth.quat_from_rotvec([0, 0, 0.2 + 0.3 * k])
        raw, _ = synth.make_scan(scene, pos, q, seed=230 + k, n_az=300)
        gp, gq = synth.perturb_pose(pos, q, seed=240 + k, max_t=0.15, max_deg=1.0)
        out.append((raw, synth.state_from_pose(gp, gq), pos))
    return out


def run_batch(sub_maps_pts, gather=None, rank=0, world=1, comm=None, device=0):
    """the scans of batch_scans() jointly against `sub_maps_pts` (this process's sub-maps); returns the result dictionaries"""
    import scenes
    from lsd_amd import lio

    maps = []
    for sub in sub_maps_pts:
        m = lio.Map(resolution=0.5, stencil=19, max_points=400_000, max_voxels=200_000, device=device)
        m.add(sub)
        maps.append(m)
    b = lio.Batch(maps[0], n_slots=4, n_groups=2, max_raw=1 << 17, max_ds=1 << 16, sub_maps=maps[1:], comm=comm)
    if gather is not None:
        b.set_gather_hook(gather, rank, world)
    P0 = lio.init_cov()
    scans_ = batch_scans()
    if os.environ.get("LIO_TEST_WIDE"):
        # three rounds of ordinary scans (the batch learns that three radix passes are enough), then a scan with two returns three hundred metres away: its
        # bounding box needs a fourth pass, the round's sort was launched without it, the job is collected as "again" -- on EVERY rank, because the
        # radix bits of a slot travel with its cloud -- and re-registered alone with four passes
        scans_ = scans_ + scans_[:5]
        raw, st, pos = scans_[2]
        wide = raw.copy()
        wide[0, :3] = [300.0, -250.0, 5.0]
        wide[1, :3] = [-280.0, 290.0, -3.0]
        scans_.append((wide, st, pos))
        scans_.append(scans_[1])
    jobs = [dict(dptr=scenes.to_device(raw), n=len(raw), t=1.0, state=st, cov=P0) for raw, st, _ in scans_]
    rc, res = b.process(jobs)
    assert rc == 0, rc
    run_batch.exchange = b.exchange_stats()
    return res


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    import torch.distributed as dist

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from lsd_amd import dist as ldist

    if mode == "gpu_batch_rccl":
        # the NATIVE exchange: one rank per GPU (cuda:rank), the C ABI's own RCCL communicator (lio_comm_init over a unique id made by rank 0 and
        # handed round over gloo), ncclAllGather of the round's [B x 32] doubles on the round's stream inside lio_batch_process
        import torch

        from lsd_amd import lio

        torch.cuda.set_device(rank)
        box = [lio.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = lio.Comm(rank=rank, world=world, device=rank, uid=box[0])
        subs4 = make_world(4)[0]
        per = 4 // world
        res = run_batch(subs4[rank * per:(rank + 1) * per], None, rank, world, comm=comm, device=rank)
        n_calls, seconds = comm.stats()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), states=np.array([r["state"] for r in res]), rcs=np.array([r["rc"] for r in res]),
                 passes=np.array([[r["n_pass"], r["n_knn_pass"]] for r in res]), calls=n_calls, seconds=seconds)
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "gpu_batch":
        n_sub = 4 if 4 % world == 0 else 2 * world  # (at least two sub-maps per rank: a batch with one map and no communicator is not a joint batch)
        subs4 = make_world(n_sub)[0]
        per = n_sub // world
        gather = ldist.RecordsAllGatherHost()
        res = run_batch(subs4[rank * per:(rank + 1) * per], gather, rank, world)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), states=np.array([r["state"] for r in res]), rcs=np.array([r["rc"] for r in res]),
                 passes=np.array([[r["n_pass"], r["n_knn_pass"]] for r in res]), calls=gather.calls, records=gather.records,
                 chunk_points=run_batch.exchange["chunk_points"], jobs_rerun=run_batch.exchange["jobs_rerun"], n_ds=np.array([r["n_ds"] for r in res]))
        dist.barrier()
        dist.destroy_process_group()
        return
    subs, raw, state, _, _ = make_world(world)
    hook = ldist.NormalEqAllGather()
    if mode == "cpu":
        import oracle

        ds = oracle.voxel_downsample(raw, 0.5)
        o = oracle.Lio(stencil=19, capacity=1 << 40, threads=2)
        o.map_add(subs[rank])
        o.set_state(state)
        o.set_flags(ekf_inited=True, first_scan=False)
        o.set_ds(ds)
        lin = o.linearize(converge=True)
        local = ldist.pack_normal_eq(lin["JtJ"], lin["Jtr"], lin["sum_abs_res"], lin["n_eff"])
        buf = local.copy()
        hook(buf)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), local=local, reduced=buf, lo_hi=np.array(ldist.shard_range(len(ds), rank, world)))
    else:
        from lsd_amd import lio

        e = lio.Engine(stencil=19, max_points=400_000, max_voxels=200_000, max_raw=1 << 17, max_ds=1 << 16, device=0)
        e.map_add(subs[rank])
        e.set_static_map(True)
        e.set_flags(ekf_inited=True, first_scan=False, first_lidar_time=-10.0)
        e.set_state(state)
        e.set_cov(lio.init_cov())
        e.set_reduce_hook(hook)
        rc = e.process_scan(raw, 1.0)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), rc=rc, state=e.get_state(), cov=e.get_cov(), calls=hook.calls)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
