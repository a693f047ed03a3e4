"""bench.py --config merge: BASELINE config 5, joint registration against sub-maps spread over the ranks; and --dry-run."""
from .common import *  # noqa: F401,F403  (argparse, json, os, sys, time, np, ROOT, BENCH_PY, the roofline constants, emit, usable_cpus ...)


def dry_run(args, dist, world, rank, local_rank):
    """the launch path of a multi-GPU run without a GPU: rendezvous, sharding, the exchange of the RCCL unique id through torch.distributed -- one
    JSON line from rank 0 saying what every rank would do"""
    from lsd_amd import lio

    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # what the real run binds and runs: cuda:<LOCAL_RANK> (torch.cuda.set_device(local_rank) in main), the secondary legs / CPU baselines / parity child on
    # a single-GPU run's rank 0 only (with N > 1 ranks no rank waits for them: nothing to time out at a barrier), one stdout line from rank 0
    info = {"rank": rank, "local_rank": local_rank, "device": f"cuda:{local_rank}", "prints_the_line": rank == 0,
            "runs_secondary_legs": bool(rank == 0 and world == 1 and args.secondary and args.config == "metric")}
    if args.config == "merge":
        plan = merge_plan(world, rank, seed=args.seed, n_keyframes=max(args.scan_pool, 16))
        info["sub_maps"] = plan["mine"]
        info["key_frames"] = len(plan["frames"])
        info["first_guess_digest"] = float(np.sum(plan["frames"][0]["guess"]))
        # the voxel-grid chain of a round runs once per scan, on the rank that owns the slot (lio_batch_create_joint over a communicator: csrc/batch.hip)
        slots = min(args.slots, 32)
        per = (slots + world - 1) // world
        info["downsamples_slots"] = [min(slots, rank * per), min(slots, rank * per + per)]
    else:
        info["scan_seeds"] = [args.seed + 100000 * rank, args.seed + 100000 * rank + args.scan_pool - 1]  # first .. last: every rank registers its own scans against its replica
    uid_ok = None
    if world > 1:
        box = [None]
        if rank == 0:
            try:
                box = [lio.Comm.unique_id()]  # librccl is loaded here (dlopen), no device needed for the id
            except Exception as ex:
                box = [repr(ex)]
        dist.broadcast_object_list(box, src=0)
        uid_ok = isinstance(box[0], bytes) and len(box[0]) == 128
        info["uid"] = uid_ok
        gathered = [None] * world
        dist.all_gather_object(gathered, info)
    else:
        gathered = [info]
    if rank == 0:
        print(json.dumps({"dry_run": True, "config": args.config, "n_gpus": world, "ranks": gathered, "rccl_unique_id_exchanged": uid_ok}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def merge_plan(world, rank, n_sub=8, n_keyframes=64, seed=1000):
    """BASELINE config 5 / overlap_merge.hpp:46-48,158-179, CPU only (also the --dry-run of a multi-rank launch): which of the 8 overlapping
    sub-maps this rank holds, the x-slab of each, and the poses / priors of the key-frame scans every rank registers"""
    from lsd_amd import synth

    if n_sub % world:
        raise SystemExit("--config merge: the 8 sub-maps must divide evenly over the GPUs (1, 2, 4 or 8)")
    edges = np.linspace(-100.0, 100.0, n_sub + 1)
    halo = 0.1 * (edges[1] - edges[0])  # 20 % overlap between neighbours
    mine = list(range(rank * n_sub // world, (rank + 1) * n_sub // world))
    rng = np.random.default_rng(seed)
    frames = []
    for k in range(n_keyframes):
        pos = np.array([rng.uniform(-80, 80), rng.uniform(-10, 10), 1.8])
        q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
        gp, gq = synth.perturb_pose(pos, q, seed=seed + 7 * k, max_t=0.3, max_deg=2.0)
        frames.append(dict(pos=pos, q=q, guess=synth.state_from_pose(gp, gq), seed=seed + k))
    return dict(edges=edges, halo=halo, mine=mine, frames=frames)


def bench_merge(args, torch, dist, world, rank, local_rank, dev):
    """BASELINE.json config 5 / SURVEY.md 8d: 8 overlapping sub-maps of 1.25e6 points spread over the N GPUs (8 / N each, one iVox map per
    sub-map); the workload of a map merge (overlap_merge.hpp:158-179: 64 key frames, each an independent alignment): every key-frame scan is
    registered JOINTLY against ALL sub-maps.  Batched and device-resident (lio_batch_create_joint): a round of 32 scans is one blind submission;
    per pass every rank linearises against its own sub-maps, ONE RCCL all-gather of [32 x 32] doubles for the whole round runs on the round's
    stream, every rank runs the same 23-DoF filter pass on the sums taken in rank order.  Scans are resident in HBM on every rank before the clock
    starts (the metric's contract).  Total work is fixed as N grows: strong scaling.  `latency` = one scan at a time through the host-driven
    joint path (lio_engine_joint_register_device: a host-synchronised collective per pass)."""
    from lsd_amd import lio, synth, synth_gpu

    plan = merge_plan(world, rank, seed=args.seed, n_keyframes=max(args.scan_pool, 16))
    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
    # map and key-frame scans are generated on the GPU (torch's generator: the same bits on every rank for the same seed), the map is cut into
    # the sub-maps on the host
    full = synth_gpu.sample_surface(scene, 8_000_000, dev, seed=2, sigma=0.01).cpu().numpy()
    scanner = synth_gpu.StaticScanner(scene, dev, n_az=args.n_az, fov_deg=(-24.8, 2.0), max_range=150.0)
    maps = []
    for k in plan["mine"]:
        sub = full[(full[:, 0] >= plan["edges"][k] - plan["halo"]) & (full[:, 0] < plan["edges"][k + 1] + plan["halo"])]
        m = lio.Map(resolution=0.5, stencil=19, max_points=2_500_000, max_voxels=1_000_000, device=local_rank)
        m.add(np.ascontiguousarray(sub))
        maps.append(m)
    if not (rank == 0 and world == 1 and args.ref_scans > 0):
        del full
    comm = None
    if world > 1:
        box = [lio.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = lio.Comm(rank=rank, world=world, device=local_rank, uid=box[0])
    args.slots, args.groups = min(args.slots, int(os.environ.get("LIO_MERGE_SLOTS", "32"))), min(args.groups, 3)  # every slot carries one scan buffer set PER LOCAL SUB-MAP: 8 x 32 x 3 of them at N = 1
    batch = lio.Batch(maps[0], n_slots=args.slots, n_groups=args.groups, max_raw=1 << 17, max_ds=100000, sub_maps=maps[1:], comm=comm)
    P0 = lio.init_cov()
    scans = []
    for f in plan["frames"]:
        d = scanner.scan(f["pos"], f["q"], seed=f["seed"])
        scans.append(dict(raw=d.cpu().numpy(), d=d, **f))
    torch.cuda.synchronize()
    jobs = [dict(dptr=scans[i % len(scans)]["d"].data_ptr(), n=len(scans[i % len(scans)]["raw"]), t=1.0 + 0.1 * i, state=scans[i % len(scans)]["guess"], cov=P0)
            for i in range(args.steps)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    rc, res = batch.process(jobs[: max(args.warmup, len(scans))] if args.warmup else jobs[: len(scans)])
    if rc != 0 or any(r["rc"] != 3 for r in res):
        raise RuntimeError(f"joint registration failed: {rc} {[r['rc'] for r in res][:8]}")
    err = max(float(np.linalg.norm(r["state"][:3] - scans[i % len(scans)]["pos"])) for i, r in enumerate(res))
    cal = lio.PreparedJobs(jobs)
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    lio.run_prepared(cal, batch=batch)
    torch.cuda.synchronize()
    repeats = max(1, int(np.ceil(1.05 * min(args.min_seconds, 3.0) / max(time.perf_counter() - c0, 1e-6))))
    if dist is not None:
        tr = torch.tensor([float(repeats)], device=dev, dtype=torch.float64)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        repeats = int(tr.item())
    prep = lio.PreparedJobs(jobs * repeats)
    barrier()
    t0 = time.perf_counter()
    rc = lio.run_prepared(prep, batch=batch)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    results = prep.results()
    if rc != 0 or any(r["rc"] != 3 for r in results):
        raise RuntimeError(f"joint registration failed in the timed region: {rc}")
    barrier()
    t_max = t_local
    if dist is not None:
        tt = torch.tensor([t_local], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
    n_timed = len(results)
    pts = sum(len(scans[(i % args.steps) % len(scans)]["raw"]) for i in range(n_timed))
    n_pass = sum(r["n_pass"] for r in results) / n_timed
    # across ranks: every rank must hold the same bits (rank 0 compares a digest of the states)
    digest = float(np.sum([np.sum(r["state"]) for r in results[: args.steps]]))
    same = True
    if dist is not None:
        dg = torch.tensor([digest], device=dev, dtype=torch.float64)
        lo, hi = dg.clone(), dg.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(lo.item() == hi.item())
    # latency leg: one scan at a time through the host-driven joint path of one slot's engines
    e0 = batch.engine(0, 0)
    lat = []
    for i in range(min(16, len(scans))):
        s = scans[i]
        t1 = time.perf_counter()
        rc1, st1, _ = e0.joint_register_device(s["d"].data_ptr(), len(s["raw"]), 1.0, s["guess"], P0)
        lat.append(time.perf_counter() - t1)
        if rc1 != 3:
            raise RuntimeError(f"joint_register_device returned {rc1}")
    coll = comm.stats() if comm is not None else (0, 0.0)
    # ---- roofline of the dominant kernel (knn_batch_kernel, launched once per local sub-map and pass): one round in flight on its own batch object,
    # HIP events around every kernel class; algorithmic bytes as in the metric config, summed over the sub-maps searched.  One GPU only (a second
    # joint batch on the communicator would put its own collectives between the ranks)
    roofline = None
    if world == 1:
        try:
            S = 19
            solo = lio.Batch(maps[0], n_slots=args.slots, n_groups=1, max_raw=1 << 17, max_ds=100000, sub_maps=maps[1:])
            sj = [jobs[i % len(jobs)] for i in range(max(4 * args.slots, len(scans)))]
            solo.process(sj[: args.slots])
            solo.enable_kernel_timing(True)
            solo.kernel_times(reset=True)
            c0s = sum(m.knn_candidates for m in maps)
            rc_s, res_s = solo.process(sj)
            kt = solo.kernel_times(reset=True)
            solo.enable_kernel_timing(False)
            del solo
            n_query = sum(r["n_ds"] * r["n_knn_pass"] for r in res_s) * len(maps)
            cand_pts = sum(m.knn_candidates for m in maps) - c0s
            L = max(int(kt["knn_launches"]), 1)
            us = kt["knn_us"] / L
            b_alg = (n_query * (16 + 16 * S) + 16.0 * cand_pts) / L
            ach = b_alg / (us * 1e-6) / 1e9 if us > 0 else 0.0
            waves = n_query / L / 4.0
            issue_us = waves * KNN_VALU_PER_WAVE_STATIC / (N_SIMD * CLOCK_GHZ * 1e3 / 4.0)
            dev_us = (kt["downsample_us"] + kt["knn_us"] + kt["linearize_us"] + kt["step_us"]) / len(sj)
            roofline = {"bound": "hbm", "limited_by": "latency / VALU issue", "kernel": "knn_batch_kernel<2, false> (one launch per local sub-map and pass, %d scans per launch)" % args.slots,
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                        "frac_basis": "algorithmic bytes (SURVEY 8d; ~36 candidates per query here, so close to what the sweep requests): no counting / PMC pass in this leg",
                        "frac_algorithmic": round(ach / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": int(b_alg), "avg_launch_us": round(us, 2), "launches": L,
                        "candidates_per_query": round(cand_pts / max(n_query, 1), 1),
                        "valu": {"wave_instructions_per_wave": KNN_VALU_PER_WAVE_STATIC, "source": "static ISA count at the metric map's trip counts (an upper bound here: "
                                 "the sub-maps hold fewer candidates per query)", "waves_per_launch": round(waves, 1), "issue_bound_us": round(issue_us, 2),
                                 "frac_of_valu_issue_peak": None},
                        "share_of_device_time": round(kt["knn_us"] / max(kt["downsample_us"] + kt["knn_us"] + kt["linearize_us"] + kt["step_us"], 1e-9), 3),
                        "other_kernels_us": {"downsample_chain_per_round": round(kt["downsample_us"] / max(int(kt["downsample_launches"]), 1), 2),
                                             "linearize_per_launch": round(kt["linearize_us"] / max(int(kt["linearize_launches"]), 1), 2),
                                             "fold_gather_filter_pass_per_launch": round(kt["step_us"] / max(int(kt["step_launches"]), 1), 2),
                                             "device_time_per_scan_one_round_in_flight": round(dev_us, 2)},
                        "timed_region": round(t_max, 4)}
        except Exception as ex:
            roofline = {"error": repr(ex)[-300:]}
    # ---- same-run baseline: the reference has no multi-map registration -- its own scan-to-map code (laserMapping.cpp h_share_model + iVox + esekfom,
    # oracle/_ref/libref_fastlio_release.so, 8 threads) registers a bounded sample of the same key-frame scans against the UNION of the eight sub-maps
    cpu = None
    if rank == 0 and world == 1 and args.ref_scans > 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import ref_fastlio

            if ref_fastlio.available(release=True):
                ref_fastlio.use_release_build()
                R = ref_fastlio.RefFastLio()
                R.set_logging(False)
                R.map_add(full)
                R.set_nearby(18)
                m_ref = min(24, args.ref_scans, len(scans))
                t_ref, p_ref, d_ref = 0.0, 0, 0.0
                for i in range(m_ref):
                    R.reset_cache()
                    c0 = time.perf_counter()
                    rc_r, sr, _ = R.register(scans[i]["raw"], scans[i]["guess"], P0)
                    t_ref += time.perf_counter() - c0
                    p_ref += len(scans[i]["raw"])
                    if rc_r == 3:
                        d_ref = max(d_ref, float(np.linalg.norm(res[i]["state"][:3] - sr[:3])))
                cpu = dict(value=round(p_ref / t_ref, 1), unit="points/s", cores=min(8, usable_cpus()), host_cpus=usable_cpus(), kind="reference",
                           sample=f"{m_ref} of the key-frame scans through the reference's own scan-to-map registration (laserMapping.cpp h_share_model + iVox + esekfom "
                                  f"update, oracle/_ref/libref_fastlio_release.so, MP_PROC_NUM=8) against the union of the eight sub-maps (8e6 points, one iVox): the "
                                  f"reference has no joint multi-map form; its map-merge tools align candidate pairs instead (overlap_merge.hpp:147-211 -- timed as "
                                  f"configs.config4_*.merge_candidates_batched with the reference's matchers beside it), {t_ref:.1f} s",
                           ms_per_scan=round(1e3 * t_ref / m_ref, 2),
                           joint_vs_union_pose_max_dpos_m=d_ref)
        except Exception as ex:
            cpu = {"error": repr(ex)[-300:]}
    if rank == 0:
        out = {"metric": "registered points/sec (multi-map merge: key-frame scans registered jointly against 8 sub-maps spread over the GPUs)",
               "value": round(pts / t_max, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": repeats,
               "timed_scans": n_timed, "timed_seconds": round(t_max, 4),
               "ms_per_step": round(1e3 * t_max / n_timed, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32 per-point geometry / f64 transforms and reductions", "data": "synthetic",
               "config": {"workload": "BASELINE config 5: 8 overlapping sub-maps of ~1.2e6 points (8e6 in total) on %d GPU(s), %d per GPU; 64x%d scans resident in HBM "
                                      "registered jointly, %d per launch (lio_batch_create_joint: per pass one linearisation per local sub-map and ONE all-gather of "
                                      "[%d x 32] doubles per round)" % (world, len(plan["mine"]), args.n_az, args.slots, args.slots),
                          "sub_maps_per_gpu": len(plan["mine"]), "passes_avg": round(n_pass, 2), "key_frames": len(scans)},
               "collective": {"per_round_and_pass": 1 if comm is not None else 0,
                              "backend": "RCCL all-gather on the round's stream (lio_allgather_records), sums in rank order inside the filter-pass kernel" if comm is not None
                              else "none (one GPU: all 8 sub-maps local; the curve over 1/2/4/8 GPUs was NOT measured here -- one-GPU boxes)",
                              "states_identical_on_all_ranks": same,
                              "downsample": dict(batch.exchange_stats(), what="with more than one rank every scan is downsampled on ONE rank (slots dealt in contiguous "
                                                 "shares) and the clouds reach the others in one all-gather per round of slot chunks sized 1.25 x the largest cloud seen "
                                                 "(lio_batch_exchange_stats); zeros on one GPU: nothing to exchange")},
               "latency": {"one_scan_at_a_time_ms": round(1e3 * float(np.median(lat)), 4),
                           "host_synchronised_collectives": coll[0], "collective_avg_us": round(coll[1] / coll[0], 2) if coll[0] else None},
               "roofline": roofline, "cpu_baseline": cpu, "pose_error_vs_truth_m": err}
        emit(out, "merge")
    if dist is not None:
        dist.destroy_process_group()
