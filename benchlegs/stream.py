"""bench.py --config stream: BASELINE config 3, the streaming front end (undistort, register, map_incremental) with the reference beside it."""
from .common import *  # noqa: F401,F403  (argparse, json, os, sys, time, np, ROOT, BENCH_PY, the roofline constants, emit, usable_cpus ...)


def bench_stream(args, torch, local_rank):
    emit(stream_run(args, torch, local_rank), "stream")


def load_bin_dir(path, scan_period=0.1):
    """recorded sweeps for --config stream --bin-dir: sorted *.bin files of x, y, z, intensity f32 records (the KITTI layout; NCLT's velodyne_sync
    converted to it), one file per sweep at 1 / scan_period Hz; per-point stamps are spread uniformly over the sweep in file order (the formats
    carry none).  imu.csv beside them (t_s, gx, gy, gz [rad/s], ax, ay, az [m/s^2]) if there is one, else a level sensor at rest (gravity only:
    the filter then runs on the lidar alone).  Returns (list of (xyzi f32 (n, 4), stamp_us uint32 (n,)), (t, gyr, acc))."""
    import glob

    files = sorted(glob.glob(os.path.join(path, "*.bin")))
    if not files:
        raise SystemExit(f"--bin-dir {path}: no *.bin files")
    sweeps = []
    for f in files:
        p = np.fromfile(f, dtype=np.float32)
        p = p[: len(p) // 4 * 4].reshape(-1, 4)
        st = np.floor(np.arange(len(p), dtype=np.float64) * (scan_period * 1e6 / max(len(p), 1))).astype(np.uint32)
        sweeps.append((np.ascontiguousarray(p), st))
    imu_csv = os.path.join(path, "imu.csv")
    if os.path.exists(imu_csv):
        m = np.loadtxt(imu_csv, delimiter=",", ndmin=2)
        imu = (m[:, 0], m[:, 1:4], m[:, 4:7])
    else:
        t = np.arange(0.0, len(files) * scan_period + 0.3, 0.01)
        imu = (t, np.zeros((len(t), 3)), np.tile([0.0, 0.0, 9.81], (len(t), 1)))
    return sweeps, imu


def stream_side_by_side(args, torch, local_rank, R, get_sweep, imu, m_ref, evict, timed, distinct=False):
    """HIP engines beside the reference `R` on the first m_ref sweeps of a drive (bench.py --config stream): teacher-forced in the default tie mode 1 (the
    reference's neighbour SETS, canonical list order) and in tie mode 2 (its list ORDER too: a parity mode, every query through the reference's selection),
    and one FREE-RUNNING in tie mode 2.  Returns (record, reference seconds, sweeps timed, points timed, sweeps at capacity, seconds at capacity, state at
    m_ref // 2).  Test infrastructure (oracle/) used as the checker / the timed CPU baseline, outside every GPU-timed region."""
    from lsd_amd import capi, lio, synth

    imu_t, imu_g, imu_a = imu

    def side_engine(tie_mode):
        e2 = lio.Engine(resolution=0.5, stencil=75, max_points=2_000_000 + 14_000 * m_ref, max_voxels=(1 << 21), max_raw=1 << 18, max_ds=100000, device=local_rank)
        if not evict:
            e2.map.set_lru((1 << 21) - 100_000, 1e9)
        e2.fastlio_init(scan_period=0.1)
        e2.map.set_tie_mode(tie_mode)
        return e2

    # (engine, state + covariance put back on the reference's after every sweep, map content too)
    sides = {"teacher_forced": (side_engine(1), True, False), "teacher_forced_state_and_map": (side_engine(1), True, True),
             "teacher_forced_state_and_map_tie_mode_2": (side_engine(2), True, True), "teacher_forced_tie_mode_2": (side_engine(2), True, False),
             "free_running_tie_mode_2": (side_engine(2), False, False)}
    tf = {name: dict(dp=[], dr=[], first_bad=None, map_cmp=0, map_same=0, map_dpts=0, map_dvox=0, ev=0, inter=0) for name in sides}
    jj, t_ref, n_ref, pts_ref, t_full, n_full, ref_half = 0, 0.0, 0, 0, 0.0, 0, None
    for k in range(m_ref):
        p, st = get_sweep(k)
        # distinct (the child process against the pinned build): both sides get the sweep in time order with pairwise DISTINCT microsecond stamps (every second / third ray where 120 000 points do not fit
        # 100 000 microseconds): the reference sorts a sweep by time with an unstable std::sort (IMU_Processing.hpp:UndistortPcl), so points with
        # equal stamps would reach its VoxelGrid in an order no other implementation can know -- with distinct stamps that sort has one result, and
        # what is compared is the path, not libstdc++'s introsort (tests/test_fastlio_vs_ref.py::_sweep does the same)
        if distinct:
            o = np.argsort(st, kind="stable")
            p, st = np.ascontiguousarray(p[o]), st[o].astype(np.int64)
            thin = int(np.ceil(len(p) / 90000.0))
            if thin > 1:
                p, st = np.ascontiguousarray(p[::thin]), st[::thin]
            ii = np.arange(len(st))
            st = (np.maximum.accumulate(st - ii) + ii).astype(np.uint32)
        tb = (k * 100000) / 1000000.0  # (the double the reference forms from its integer microsecond header stamp: k * 0.1 differs from it in the last bit for some k, and a point or an IMU sample exactly on a boundary then falls on the other side)
        while jj < len(imu_t) and imu_t[jj] <= tb + 0.12:
            R.imu_enqueue(imu_t[jj], imu_g[jj], imu_a[jj])
            for e2, _, _ in sides.values():
                e2.fastlio_imu_enqueue(imu_t[jj], imu_g[jj], imu_a[jj])
            jj += 1
        c0 = time.perf_counter()
        R.pcl_enqueue(p, st, k * 100000)
        updated = R.main()
        c1 = time.perf_counter()
        s_ref, _, P_ref = R.state()
        ref_map = R.map_dump() if any(fm for _, _, fm in sides.values()) else None
        for name, (e2, forced, forced_map) in sides.items():
            e2.fastlio_pcl_enqueue(p, st, tb)
            rc2 = e2.fastlio_main()
            e2.flush()
            if rc2 == capi.MAIN_UPDATED and updated:
                s2 = e2.get_state()
                rec = tf[name]
                rec["dp"].append(float(np.linalg.norm(s2[0:3] - s_ref[0:3])))
                rec["dr"].append(float(synth.quat_angle(s2[3:7], s_ref[3:7])))
                if rec["first_bad"] is None and (rec["dp"][-1] > 1e-4 or rec["dr"][-1] > 1e-5):
                    rec["first_bad"] = k
                if forced:
                    e2.set_state(s_ref)
                    e2.set_cov(P_ref)
            if forced_map and ref_map is not None and len(ref_map) and e2.map.stats()[1] > 0:
                # both sides began this sweep with the same map content and (to the figures above) the same pose: what map_incremental + AddPoints (+ the
                # LRU list) made of it must agree -- voxel and point counts compared before the engine's map is replaced (the one place they may not:
                # the LRU list's point-by-point order inside a batch)
                # (Only on a drive without evictions: the map put back below carries the reference's points in push_back order, not its LRU list -- order
                # of last touch, distance at creation -- so with the quota in force the engine's next eviction starts from another list.)
                pts2, vox2 = e2.map.stats() if not evict else (len(ref_map), R.map_voxels())
                rec = tf[name]
                rec["map_cmp"] += 0 if evict else 1
                rec["map_same"] += int(pts2 == len(ref_map) and vox2 == R.map_voxels())
                rec["map_dpts"] = max(rec["map_dpts"], abs(int(pts2) - len(ref_map)))
                rec["map_dvox"] = max(rec["map_dvox"], abs(int(vox2) - int(R.map_voxels())))
                ev, inter = e2.map.lru_stats()  # (of this sweep's insert: lio_map_clear below resets the map's counters)
                rec["ev"] += int(ev)
                rec["inter"] += int(inter)
                # the reference's map after this sweep, voxel by voxel in push_back order (IVox::GetAllPoints), in place of the engine's own
                e2.map.clear()
                e2.map.add(ref_map, float(R.info()["travel_distance"]))
        if k + 1 == m_ref // 2:
            ref_half = R.get_state().copy()
        if timed and k >= 20:
            t_ref += c1 - c0
            n_ref += 1
            pts_ref += len(p)
            if R.map_voxels() >= 100000:
                t_full += c1 - c0
                n_full += 1
    per = {}
    for name, (e2, forced, forced_map) in sides.items():
        rec = tf[name]
        if rec["dp"]:
            a_dp, a_dr = np.array(rec["dp"]), np.array(rec["dr"])
            per[name] = {"sweeps": int(len(a_dp)), "max_dpos_m": float(a_dp.max()), "max_drot_rad": float(a_dr.max()), "median_dpos_m": float(np.median(a_dp)),
                         "p99_dpos_m": float(np.percentile(a_dp, 99)), "last_dpos_m": float(a_dp[-1]),
                         "sweeps_beyond_1e_4_m_or_1e_5_rad": int(np.count_nonzero((a_dp > 1e-4) | (a_dr > 1e-5))), "first_sweep_beyond": rec["first_bad"],
                         "map_voxels_end": {"gpu": int(e2.map.stats()[1]), "reference": int(R.map_voxels())}}
            if rec["map_cmp"]:
                per[name]["map_after_every_sweep"] = {"sweeps_compared": rec["map_cmp"], "sweeps_with_the_references_voxel_and_point_counts": rec["map_same"],
                                                      "max_point_count_difference": rec["map_dpts"], "max_voxel_count_difference": rec["map_dvox"],
                                                      "voxels_evicted": rec["ev"], "lru_back_voxels_touched_by_the_evicting_batch_upper_bound": rec["inter"]}
        e2.close()
    return per, t_ref, n_ref, pts_ref, n_full, t_full, ref_half


def stream_tf_pinned(args, torch, local_rank):
    """child process of stream_run (one build of the reference per process): the same drive's first sweeps beside the PINNED build of the reference
    (oracle/_ref/libref_fastlio.so: scalar Eigen, no contraction), whose bits tie mode 2 follows.  One JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_fastlio
    from lsd_amd import synth, synth_gpu

    if not ref_fastlio.available():
        print(json.dumps({"error": "oracle/_ref/libref_fastlio.so is not there"}))
        return
    dev = torch.device("cuda", local_rank)
    scene = synth.Scene(half=500.0, n_boxes=1500, seed=3, box_size=(4.0, 30.0), keep_clear=8.0)
    tr = synth_gpu.Lawnmower(speed=args.speed) if args.grow_to else synth.FigureEight()
    m_ref = args.ref_scans
    sweeper = synth_gpu.Sweeper(scene, tr, dev, fov_deg=(-24.8, 2.0), max_range=100.0, seed=args.seed)
    imu = synth_gpu.imu_stream(tr, 0.0, args.steps * 0.1 + 0.3, rate=100.0, seed=args.seed, gyr_sigma=1e-3, acc_sigma=1e-2)
    R = ref_fastlio.RefFastLio(scan_period=0.1)
    R.set_logging(False)
    per = stream_side_by_side(args, torch, local_rank, R, sweeper.sweep, imu, m_ref, args.lru > 0, False, distinct=True)[0]
    per["build"] = "oracle/_ref/libref_fastlio.so: the reference's translation units with scalar Eigen and no FMA contraction -- the build the path is pinned to"
    per["sweeps_as_fed"] = ("in time order with pairwise distinct microsecond stamps, thinned to <= 90 000 points so that they fit the 100 ms: the reference's unstable "
                            "std::sort by time (UndistortPcl) then has one result -- what is compared is the path, not libstdc++'s introsort on tied stamps")
    print(json.dumps(per))


def stream_run(args, torch, local_rank):
    """BASELINE.json config 3 / SURVEY.md 8d ("NCLT replay", stand-in: NCLT is not available): the streaming FastLIO front half -- IMU
    propagation, motion compensation, downsample, iterated update, map_incremental -- through lio_fastlio_* at 10 Hz, clouds from the host
    (PCIe inside the timed region).  --grow-to N: a lawnmower course at --speed m/s over a 1 km x 1 km scene (new ground all the time) until
    the map holds N points -- no eviction (--lru 0; SURVEY 8d: the reference's 100000-voxel LRU cap would evict, a stated deviation); otherwise
    round 2's figure of eight at 5 m/s for --steps sweeps, with the reference's LRU capacity (--lru 100000) or without (--lru 0).
    Sweeps are generated on the GPU between the timed calls (lsd_amd/synth_gpu.py), or read from --bin-dir."""
    from lsd_amd import capi, lio, synth, synth_gpu

    dev = torch.device("cuda", local_rank)
    n, lru, seed, grow_to = args.steps, args.lru, args.seed, args.grow_to
    t_gen = 0.0
    tr = None
    if args.bin_dir:
        recorded, (imu_t, imu_g, imu_a) = load_bin_dir(args.bin_dir)
        n = min(n, len(recorded))
        course = f"recorded sweeps from {args.bin_dir}"

        def get_sweep(k):
            return recorded[k]
    else:
        scene = synth.Scene(half=500.0, n_boxes=1500, seed=3, box_size=(4.0, 30.0), keep_clear=8.0)
        if grow_to:
            tr = synth_gpu.Lawnmower(speed=args.speed)
            n = min(n, int(tr.duration() / 0.1) - 1)
            course = "lawnmower course (10 rows of %.0f m, %.0f m apart) at %.0f m/s over a 1 km x 1 km scene" % (2 * tr.half_len, tr.spacing, args.speed)
        else:
            tr = synth.FigureEight()
            course = "figure of eight (5 m/s) through a 1 km x 1 km scene"
        sweeper = synth_gpu.Sweeper(scene, tr, dev, fov_deg=(-24.8, 2.0), max_range=100.0, seed=seed)
        imu_t, imu_g, imu_a = synth_gpu.imu_stream(tr, 0.0, n * 0.1 + 0.3, rate=100.0, seed=seed, gyr_sigma=1e-3, acc_sigma=1e-2)

        def get_sweep(k):
            return sweeper.sweep(k)
    evict = lru > 0
    big = bool(grow_to)
    e = lio.Engine(resolution=0.5, stencil=75, max_points=(max(grow_to, 10_000_000) * 13 // 10) if big else 14_000_000,
                   max_voxels=(1 << 21) if evict else ((1 << 23) if big else 6_000_000), max_raw=1 << 18, max_ds=100000, device=local_rank)
    if not evict:
        e.map.set_lru(((1 << 23) if big else 6_000_000) - 100_000, 1e9)  # a capacity the drive never reaches: nothing is evicted
    e.fastlio_init(scan_period=0.1)  # turns on the reference's 100000-voxel / 100 m LRU list unless one was set above
    ii, t_main, t_enq, t_fl, rows, pts = 0, [], [], [], [], 0
    by_size = []  # (map points at the time, main seconds) for the curve "ms per scan against map size"
    map_points = 0
    k_done = 0
    insert_leg = None
    timing_left = -1
    last_states = []
    gpu_state_at = {}
    err_curve = []  # (metres driven, position error against the generating trajectory): odometry drift, no loop closure on this path
    for k in range(n):
        g0 = time.perf_counter()
        p, st = get_sweep(k)
        t_gen += time.perf_counter() - g0
        tb = (k * 100000) / 1000000.0  # (the double the reference forms from its integer microsecond header stamp: k * 0.1 differs from it in the last bit for some k, and a point or an IMU sample exactly on a boundary then falls on the other side)
        while ii < len(imu_t) and imu_t[ii] <= tb + 0.12:
            e.fastlio_imu_enqueue(imu_t[ii], imu_g[ii], imu_a[ii])
            ii += 1
        t0 = time.perf_counter()
        e.fastlio_pcl_enqueue(p, st, tb)
        t1 = time.perf_counter()
        rc = e.fastlio_main()
        t2 = time.perf_counter()
        k_done = k + 1
        if rc < 0:
            raise RuntimeError(f"lio_fastlio_main returned {rc} at scan {k}")
        if tr is not None and k % 250 == 249:
            tk = k_done * 0.1
            sk = e.get_state()
            driven = float(tr._d(tk)) if hasattr(tr, "_d") else None
            dk = sk[0:3] - tr.R(0.0).T @ (tr.pos(tk) - tr.pos(0.0))
            err_curve.append([None if driven is None else round(driven, 1), round(float(np.linalg.norm(dk)), 3), round(float(dk[2]), 3)])
        # the scan's map_incremental was enqueued by fastlio_main, not waited for; the wait (its count, and an overflow, are read here) is TIMED:
        # this loop feeds the next sweep only after the insert is done, like the reference, whose fastlio_main inserts synchronously -- so the part of
        # the insert that fastlio_main's return did not cover belongs to the sweep's cost (it used to fall between the clocks)
        e.flush()
        t3 = time.perf_counter()
        last_states.append((k, e.get_state()))  # (the engine's own poses of the last sweeps: the priors of the kNN leg on the grown map)
        if args.ref_scans > 0 and k + 1 in (min(args.ref_scans, n) // 2, min(args.ref_scans, n)):
            gpu_state_at[k + 1] = last_states[-1][1].copy()  # (the pose where the reference's own drive over the same sweeps is compared, below)
        if len(last_states) > 32:
            last_states.pop(0)
        if timing_left > 0:  # the insert-side roofline leg: per-stage HIP events on (these sweeps are not in the ms/scan figure)
            if rc == capi.MAIN_UPDATED:
                tm = e.timings()
                for key in ("downsample_us", "knn_us", "linearize_us", "insert_us", "undistort_us"):
                    insert_leg[key] += tm[key]
                insert_leg["n_ds"] += tm["n_ds"]
                insert_leg["n_added"] += tm["n_added"]
                insert_leg["scans"] += 1
            timing_left -= 1
            if timing_left == 0:
                break
            continue
        if rc == capi.MAIN_UPDATED and k >= 20:
            t_enq.append(t1 - t0)
            t_main.append(t2 - t1)
            t_fl.append(t3 - t2)
            pts += len(p)
            tm = e.timings()
            rows.append((tm["n_ds"], tm["n_pass"], tm["n_knn_pass"], tm["n_added"]))
            if k % 25 == 0:
                map_points = e.map.stats()[0]
            by_size.append((map_points, t2 - t1))
        if timing_left < 0 and ((grow_to and map_points >= grow_to) or k == n - 101):
            # target reached (or the course is about to end): 100 more sweeps with per-stage events for the insert-side roofline
            e.enable_timing(True)
            insert_leg = dict(downsample_us=0.0, knn_us=0.0, linearize_us=0.0, insert_us=0.0, undistort_us=0.0, n_ds=0, n_added=0, scans=0)
            timing_left = 100
    e.enable_timing(False)
    # ---- the stencil search on THIS map (grown by map_incremental: a few points per voxel, not the 39 of the metric config's pre-built one): the last
    # sweeps once more as independent jobs of a 16-slot batch against the engine's map, HIP events per kernel class, then the counting variant
    knn_grown = None
    if len(last_states) >= 16:
        try:
            e.flush()
            S = 19
            d_sw, jb = [], []
            P0 = lio.init_cov()
            for k, st_k in last_states:  # the prior of a job: the engine's own state after that sweep (the map lives in ITS frame, drift included)
                p, _ = get_sweep(k)
                d = torch.from_numpy(p).to(dev)
                d_sw.append(d)
                jb.append(dict(dptr=d.data_ptr(), n=len(p), t=1.0 + 0.1 * k, state=st_k, cov=P0))
            torch.cuda.synchronize()
            solo = lio.Batch(e.map, n_slots=16, n_groups=1, max_raw=1 << 18, max_ds=100000)
            solo.process(jb[:16])
            solo.enable_kernel_timing(True)
            solo.kernel_times(reset=True)
            c1 = e.map.knn_candidates
            _, res_s = solo.process(jb)
            kt = solo.kernel_times(reset=True)
            n_q = sum(r["n_ds"] * r["n_knn_pass"] for r in res_s)
            cand_pts = e.map.knn_candidates - c1
            solo.enable_kernel_timing(2)
            solo.kernel_times(reset=True)
            t0c = e.map.knn_touched
            solo.process(jb)
            solo.kernel_times(reset=True)
            touched = e.map.knn_touched - t0c
            solo.enable_kernel_timing(False)
            del solo
            L = max(int(kt["knn_launches"]), 1)
            us = kt["knn_us"] / L
            b_alg = (n_q * (16 + 16 * S) + 16.0 * cand_pts) / L
            b_tch = (n_q * (16 + 16 * S) + 16.0 * touched) / L
            knn_grown = {"what": "knn_batch_kernel on the map this drive grew: the last 32 sweeps as independent jobs, 16 per launch, one round in flight",
                         "map_points": int(e.map.stats()[0]), "map_voxels": int(e.map.stats()[1]),
                         "candidates_per_query": round(cand_pts / max(n_q, 1), 1), "touched_per_query": round(touched / max(n_q, 1), 1),
                         "queries": int(n_q), "searches": int(sum(r["n_knn_pass"] for r in res_s)), "registered": int(sum(1 for r in res_s if r["rc"] == 3)),
                         "us_per_scan_and_search": round(kt["knn_us"] / max(sum(r["n_knn_pass"] for r in res_s), 1), 2),
                         "avg_launch_us": round(us, 2), "launches": L,
                         "frac": round(b_alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us > 0 else None,
                         "frac_touched": round(b_tch / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us > 0 else None,
                         "algorithmic_bytes_per_launch": int(b_alg), "touched_bytes_per_launch": int(b_tch), "traffic": None}
        except Exception as ex:
            knn_grown = {"error": repr(ex)[-300:]}
    s = e.get_state()
    err = None
    if tr is not None:
        R0, p0 = tr.R(0.0), tr.pos(0.0)
        err = float(np.linalg.norm(s[0:3] - R0.T @ (tr.pos(k_done * 0.1) - p0)))
    map_points, map_voxels = e.map.stats()
    evicted = e.map.lru_stats()[0]
    recreated, not_followed = e.map.lru_exact_stats()
    rows = np.array(rows, dtype=np.float64)
    tot = float(np.sum(t_main) + np.sum(t_enq) + np.sum(t_fl))  # enqueue + fastlio_main + the wait for its map_incremental
    curve = []
    if by_size:
        bs = np.array(by_size)
        edges = np.arange(0, bs[:, 0].max() + 1e6, 1e6)
        for a_, b_ in zip(edges[:-1], edges[1:]):
            m = (bs[:, 0] >= a_) & (bs[:, 0] < b_)
            if m.sum() >= 5:
                curve.append([round(b_ / 1e6, 1), round(1e3 * float(np.median(bs[m, 1])), 4)])
    roofline = None
    if insert_leg and insert_leg["scans"]:
        ns = insert_leg["scans"]
        b_ins = (16.0 * insert_leg["n_ds"] + 32.0 * insert_leg["n_added"]) / ns      # B_ins = 16 N_ds (read) + (16 + 16) N_add
        us = insert_leg["insert_us"] / ns
        ach = b_ins / (us * 1e-6) / 1e9 if us > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": "map_incremental chain (classify_kernel + classify_scatter_kernel + map_insert_* [+ lru_*])", "achieved": round(ach, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None, "algorithmic_bytes_per_launch": int(b_ins),
                    "avg_launch_us": round(us, 2), "launches": ns, "map_points_at_measurement": int(map_points),
                    "stage_us_per_scan": {k2: round(insert_leg[k2] / ns, 2) for k2 in ("undistort_us", "downsample_us", "knn_us", "linearize_us", "insert_us")},
                    "n_ds_avg": round(insert_leg["n_ds"] / ns, 1), "n_added_avg": round(insert_leg["n_added"] / ns, 1),
                    "note": "stage times from HIP events on the engine's stream (lio_engine_enable_timing) over the 100 sweeps after the timed part; knn_us / "
                            "linearize_us are whole passes (kernels + hand-over); a few thousand points in ~6 launches: launch latency, not bandwidth"}
    # ---- same-run baseline: the reference's OWN FastLIO translation units (oracle/_ref/libref_fastlio_release.so: laserMapping.cpp, IMU_Processing.hpp,
    # iVox, IKFoM with its CMake flags) streaming the first sweeps of the same drive on the host ----
    cpu = None
    if args.ref_scans > 0 and not args.bin_dir:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import ref_fastlio

            if ref_fastlio.available(release=True):
                ref_fastlio.use_release_build()
                R = ref_fastlio.RefFastLio(scan_period=0.1)
                R.set_logging(False)
                m_ref = min(args.ref_scans, k_done)
                per, t_ref, n_ref, pts_ref, n_full, t_full, ref_half = stream_side_by_side(args, torch, local_rank, R, get_sweep, (imu_t, imu_g, imu_a), m_ref, evict, True)
                gpu_same = float(np.mean((np.array(t_main) + np.array(t_enq) + np.array(t_fl))[: max(n_ref, 1)]))
                ref_err = None
                if tr is not None and m_ref > 0:
                    ref_err = float(np.linalg.norm(R.get_state()[0:3] - tr.R(0.0).T @ (tr.pos(m_ref * 0.1) - tr.pos(0.0))))
                cpu = dict(value=round(pts_ref / t_ref, 1), unit="points/s", cores=min(8, usable_cpus()), host_cpus=usable_cpus(), kind="reference",
                           sample=f"sweeps 20..{m_ref - 1} of the same drive through the reference's own fastlio_imu_enqueue / fastlio_pcl_enqueue / fastlio_main "
                                  f"(IMU propagation, undistortion, VoxelGrid [the oracle's restatement], iVox kNN on MP_PROC_NUM=8 threads, esekfom update, "
                                  f"map_incremental with its 100000-voxel LRU), {t_ref:.1f} s; its map is capped at 100000 voxels by its LRU "
                                  f"(sweeps_with_the_reference_map_at_capacity says for how many of these sweeps it was full)",
                           ms_per_scan=round(1e3 * t_ref / max(n_ref, 1), 3), gpu_ms_per_scan_same_sweeps=round(1e3 * gpu_same, 4),
                           sweeps_with_the_reference_map_at_capacity=n_full, ms_per_scan_at_capacity=(round(1e3 * t_full / n_full, 3) if n_full else None),
                           reference_map_voxels_end=int(R.map_voxels()),
                           pose_error_vs_truth_m=ref_err, at_sweep=m_ref)
                # GPU engine against the reference's own FastLIO along the SAME drive: two filters fed the same sweeps part by the amplification of
                # last-bit differences (the drive is long: a trajectory-level figure, not the per-scan tolerance of the static-map legs)
                gv = {"what": "|GPU position - reference position| after the same sweeps of the same drive (both start from the same state; every "
                              "registration feeds the next prior and the map: differences of the last bit amplify along a drive)"}
                sr_end = R.get_state()
                for at, sr in ((m_ref // 2, ref_half), (m_ref, sr_end)):
                    if sr is not None and at in gpu_state_at:
                        gv[f"dpos_m_after_{at}_sweeps"] = float(np.linalg.norm(gpu_state_at[at][0:3] - sr[0:3]))
                        gv[f"drot_rad_after_{at}_sweeps"] = float(synth.quat_angle(gpu_state_at[at][3:7], sr[3:7]))
                cpu["gpu_vs_reference_drive"] = gv
                if per:
                    per["what"] = ("HIP engines fed the same IMU stream and sweeps in step with the reference.  teacher_forced*: the engine is put back on the reference's "
                                   "posterior (state + covariance) after every sweep, so every figure is ONE sweep's difference from the same prior -- IMU propagation, "
                                   "undistortion, downsample, iterated update -- against maps grown by the same inserts (the maps are NOT copied over: a map_incremental "
                                   "decision that flips leaves another point in a young map of one or two points per voxel, which the next sweeps register against; "
                                   "voxel counts compared at the end).  tie_mode_2 = lio_map_set_tie_mode(2): the neighbour lists in the reference's own order "
                                   "(tests/test_fastlio_golden.py holds that mode to 1e-12 m per sweep against the pinned build).  free_running_tie_mode_2: never reset.  "
                                   "Here against the build that is timed (the reference's own flags, vectorised Eigen); `against_the_pinned_build`: the same against "
                                   "the scalar-Eigen build, in a child process")
                    try:
                        import subprocess

                        pr = subprocess.run([sys.executable, BENCH_PY, "--config", "stream", "--tf-pinned", "--steps", str(max(m_ref + 5, 30)), "--ref-scans", str(m_ref),
                                             "--lru", str(args.lru), "--grow-to", str(args.grow_to), "--speed", str(args.speed), "--seed", str(args.seed)],
                                            capture_output=True, text=True, timeout=800)
                        line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
                        per["against_the_pinned_build"] = json.loads(line[-1]) if (pr.returncode == 0 and line) else {"error": (pr.stderr or pr.stdout)[-300:]}
                    except Exception as ex:
                        per["against_the_pinned_build"] = {"error": repr(ex)[-300:]}
                    cpu["gpu_vs_reference_per_sweep"] = per
        except Exception as ex:
            cpu = {"error": repr(ex)[-300:]}
    out = {"metric": "registered points/sec (streaming LIO front half, incremental map)", "value": round(pts / tot, 1), "unit": "points/s", "n_gpus": 1,
           "steps": len(t_main), "warmup": 20, "ms_per_step": round(1e3 * tot / len(t_main), 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 per-point geometry / f64 transforms and reductions", "data": "recorded" if args.bin_dir else "synthetic",
           "config": {"workload": "BASELINE config 3 stand-in: %d sweeps of 64x1875 rays at 10 Hz along a %s, 100 Hz IMU, "
                                  "lio_fastlio_* (IMU propagation + undistortion + downsample + iterated update + map_incremental), clouds from the host" % (len(t_main) + 20, course),
                      "lru_capacity_voxels": lru if evict else None, "grow_to": grow_to or None,
                      "n_ds_avg": round(float(rows[:, 0].mean()), 1), "passes_avg": round(float(rows[:, 1].mean()), 2),
                      "knn_passes_avg": round(float(rows[:, 2].mean()), 2), "points_added_per_scan": round(float(rows[:, 3].mean()), 1),
                      "map_points_end": int(map_points), "map_voxels_end": int(map_voxels), "voxels_evicted": int(evicted),
                      "voxels_dropped_and_recreated_inside_a_batch": int(recreated), "batches_in_which_the_lru_order_was_not_followed": int(not_followed),
                      "main_ms_median": round(1e3 * float(np.median(t_main)), 4), "enqueue_ms_median": round(1e3 * float(np.median(t_enq)), 4),
                      "main_ms_p99": round(1e3 * float(np.percentile(t_main, 99)), 4), "main_ms_median_by_map_size_Mpts": curve,
                      "insert_wait_ms_median": round(1e3 * float(np.median(t_fl)), 4),
                      "ms_per_scan_without_the_insert_wait": round(1e3 * float(np.sum(t_main) + np.sum(t_enq)) / len(t_main), 4),
                      "timing": "ms_per_step = enqueue + lio_fastlio_main + the wait for the map_incremental it enqueued (lio_engine_flush), per sweep: the "
                                "synchronous cost, comparable with the reference's fastlio_main; main_ms_* are lio_fastlio_main alone (state final, insert in flight)",
                      "sweep_generation_s": round(t_gen, 1)},
           "roofline": roofline, "knn_on_this_map": knn_grown, "cpu_baseline": cpu, "pose_error_vs_truth_m": err,
           "drift": {"metres_driven__position_error_m__its_vertical_part_m": err_curve,
                     "note": "pure odometry (no loop closure, no GNSS on this path): drift against the generating trajectory, mostly vertical on this flat "
                             "synthetic ground; the reference's own FastLIO build drifts the same way on the same sweeps (cpu_baseline.pose_error_vs_truth_m "
                             "at its last sweep; profiles/r03_drift_vs_reference.txt follows both for 1200 sweeps)"}}
    e.close()
    return out
