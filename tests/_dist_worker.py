"""worker for tests/test_dist.py: one rank of a world_size-N gloo job.  argv: mode outdir
mode "cpu": per-rank normal equations from the CPU oracle on this rank's sub-map, reduced with NormalEqAllGather
mode "gpu": a full joint registration with lio.Engine + reduce hook on cuda:0 (collective over gloo)"""
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


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    import torch.distributed as dist

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from lsd_amd import dist as ldist

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
