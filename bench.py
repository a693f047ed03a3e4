#!/usr/bin/env python
"""bench.py -- registered points/sec of the LIO scan-matching hot path on MI355X.

Workload (BASELINE.json metric): synthetic 64-beam scans (64 x 1875 = 120 000 points) registered against a
1e7-point static map resident in HBM -- per scan: voxel-grid downsample (leaf 0.5 m) + the iterated ESKF
update (<= 5 passes, each = body->world, [stencil kNN], plane fit, gate, J^T J reduction, host 23-DoF solve).
A "step" is one scan.  Inputs (map + raw scans) are in HBM before the timed region.  Scans are independent,
so with N GPUs every rank registers its own K scans against its own replica of the map (weak scaling, no
data-path collective).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# HIP deals streams to hardware queues round robin; with the default of 4 the rounds in flight of the batched engine share queues with
# idle streams and mostly run back to back (measured: 0.085 -> 0.070 ms per scan with 8).  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lidar-slam-detection_amd", "python"))

import bench_line  # the compact stdout line (bench_line.py, beside this file)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
N_SIMD, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs, peak engine clock
KNN_VALU_PER_WAVE_STATIC = 1630.0  # VALU instructions of one wave (four queries) of knn_batch_kernel at the metric map's trip counts
FULL_LINE = False  # --full-line (the children of the metric config's secondary legs): the whole record on stdout instead of the compact line


def emit(out, tag="metric"):
    """rank 0's output: the whole record to bench_full[_<tag>].json beside this file and to stderr, the compact line (<= bench_line.LIMIT bytes: the
    driver's parser did not take round 4's 30 KB line) as the ONE stdout line"""
    if FULL_LINE:
        print(json.dumps(out))
        return
    name = "bench_full.json" if tag == "metric" else f"bench_full_{tag}.json"
    try:
        with open(os.path.join(ROOT, name), "w") as f:
            json.dump(out, f, indent=1)
        out = dict(out, full_record=name)
    except OSError:
        pass
    print("bench.py full record: " + json.dumps(out), file=sys.stderr)
    sys.stderr.flush()
    print(bench_line.line(out))
    sys.stdout.flush()


_VALU_PEAK = {}


def measured_valu_peak(device=0, waves_per_simd=6, mix=1):
    """the chip's VALU issue rate in wave-instructions per second, MEASURED by tools/valu_peak (every SIMD holding `waves_per_simd` waves of
    independent v_add_u32 / v_min_u32 / DPP work -- the kNN merge's diet); None when the tool's library is not built"""
    key = (device, waves_per_simd, mix)
    if key not in _VALU_PEAK:
        _VALU_PEAK[key] = None
        try:
            import ctypes as C

            L = C.CDLL(os.path.join(ROOT, "tools", "valu_peak", "libvalu_peak.so"))
            L.valu_peak_wave_insts_per_s.restype = C.c_double
            L.valu_peak_wave_insts_per_s.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
            cus, mhz = C.c_int(0), C.c_int(0)
            best = max(L.valu_peak_wave_insts_per_s(device, waves_per_simd, mix, C.byref(cus), C.byref(mhz)) for _ in range(3))
            if best > 0:
                _VALU_PEAK[key] = {"wave_insts_per_s": best, "cus": cus.value, "clock_mhz": mhz.value,
                                   "cycles_per_wave_inst_at_reported_clock": (cus.value * 4 * mhz.value * 1e6 / best) if mhz.value else None,
                                   "mix": {0: "v_add_u32", 1: "v_add_u32 / v_min_u32 / v_add_u32_dpp row_ror", 2: "v_fma_f32", 3: "v_add_f64"}[mix],
                                   "waves_per_simd": waves_per_simd}
        except Exception:
            _VALU_PEAK[key] = None
    return _VALU_PEAK[key]


def usable_cpus():
    """host CPUs this process may use: the affinity mask, capped by the cgroup quota (cpu.max) of the container"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def spawn_ranks(n):
    """re-run this command line as `n` ranks of torch.distributed.run on this node (what the docstring's second form does by hand)"""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--n-az", type=int, default=1875)
    ap.add_argument("--scan-pool", type=int, default=128, help="distinct scans cycled through the timed jobs (SURVEY 8d config 2: distinct seeds, poses all over the map)")
    ap.add_argument("--spread", type=float, default=90.0, help="sensor positions of the scan pool: uniform over [-spread, spread]^2 of the 200 m x 200 m scene")
    ap.add_argument("--cpu-scans", type=int, default=320, help="scans of the same workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--ref-scans", type=int, default=160, help="scans of the same workload timed on the reference's own code, oracle/_ref/libref_fastlio_release.so (0 = skip)")
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--frame-z", type=float, default=0.0, help="--config metric: the map frame's origin sits this far ABOVE the scene's ground (the synthetic scene has its ground at "
                    "z = 0, i.e. the ground plane passes through the map frame's origin -- the one place where esti_plane's n.x + 1 = 0 form is ill-conditioned; a real "
                    "drive's map frame is the first IMU pose, ~1.8 m above the ground: --frame-z 1.8 shifts map, poses and priors accordingly; parity soaks use it)")
    ap.add_argument("--streams", type=int, default=0, help="--engine threads: independent scans in flight per GPU (one engine + HIP stream + host thread "
                                                              "each, all reading the one resident map)")
    ap.add_argument("--engine", choices=["batch", "threads"], default="batch",
                    help="batch: B scans per launch, filter loop on the device, one host thread (lio_batch_*); threads: round 1's one engine + thread per scan")
    ap.add_argument("--slots", type=int, default=128, help="--engine batch: scans per launch (round 5 sweep, tools/experiments/README.md: 64 x 4 0.0182, 96 x 4 0.0176, "
                                                             "128 x 4 0.0176, 160 x 4 0.0175, 256 x 3 0.0177 ms per scan; the kNN kernel's cost per search falls from 7.1 to "
                                                             "6.0 us between 64 and 256 scans per launch)")
    ap.add_argument("--groups", type=int, default=4, help="--engine batch: rounds in flight (one HIP stream each)")
    ap.add_argument("--config", choices=["metric", "merge", "stream", "localize", "sequences", "refparity", "rcclprobe"], default="metric",
                    help="metric: BASELINE.json's headline (independent 120k-pt scans vs a 1e7-pt map); merge: BASELINE config 5, multi-map merge -- 8 sub-maps "
                         "spread over the GPUs, every key-frame scan registered JOINTLY against all of them (RCCL all-gather of the per-rank J^T J / J^T r); "
                         "stream: BASELINE config 3 (streaming front half, incremental map); localize: BASELINE config 4 (NDT scan-to-map vs a 5e7-pt resident map)")
    ap.add_argument("--grow-to", type=int, default=0, help="--config stream: drive the lawnmower course (new ground all the time) until the map holds this many points "
                                                           "(BASELINE config 3: 10000000) or --steps sweeps are done; 0 = the figure of eight of round 2, --steps sweeps")
    ap.add_argument("--speed", type=float, default=20.0, help="--config stream --grow-to: driving speed, m/s")
    ap.add_argument("--bin-dir", default=None, help="--config stream: replay recorded sweeps instead of synthetic ones -- a directory of *.bin files of x,y,z,intensity f32 "
                                                    "records (KITTI / converted NCLT velodyne_sync), 10 Hz, optional imu.csv (t_s,gx,gy,gz,ax,ay,az) beside them")
    ap.add_argument("--dense-points", type=int, default=50_000_000, help="--config localize: points of the prebuilt map resident in HBM")
    ap.add_argument("--vgicp-scans", type=int, default=6, help="--config localize: alignments timed on the reference's CPU fallback matcher (FastVGICP, 4 threads)")
    ap.add_argument("--lru", type=int, default=100000, help="--config stream: iVox capacity in voxels (the reference's 100000, laserMapping.cpp:1063); 0 = never evict")
    ap.add_argument("--prior-t", type=float, default=0.3, help="prior error of a scan, metres (BASELINE: within 0.3 m)")
    ap.add_argument("--prior-deg", type=float, default=2.0, help="prior error of a scan, degrees (BASELINE: within 2 deg)")
    ap.add_argument("--secondary", type=int, default=1, help="N = 1, --config metric: also run BASELINE config 2 (1e6-pt map) and a short config 3 "
                                                             "(streaming, map_incremental + LRU) after the timed region and report them under `configs`")
    ap.add_argument("--dry-run", action="store_true", help="everything up to the first HIP call, on the CPU: arguments, the torch.distributed rendezvous (gloo), the "
                                                           "sharding of the work over the ranks, the RCCL unique id exchange -- a launch check for multi-GPU runs on a box without GPUs")
    ap.add_argument("--parity-scans", type=int, default=32, help="scans of the pool registered by the PINNED build of the reference (scalar Eigen, oracle/_ref/libref_fastlio.so) in a "
                                                                  "child process for cpu_baseline.gpu_vs_reference_pose.pinned_build")
    ap.add_argument("--parity-dir", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--tf-pinned", action="store_true", help=argparse.SUPPRESS)  # --config stream: the child process that runs the side-by-side legs against the pinned build
    ap.add_argument("--probe", default=None, help=argparse.SUPPRESS)  # --config rcclprobe: the child process of rccl_probe (its job as JSON, or "uid")
    ap.add_argument("--full-line", action="store_true", help=argparse.SUPPRESS)  # the whole record on stdout (the metric config's children of the secondary legs)
    ap.add_argument("--upload-scans", type=int, default=-1, help="--config metric: scans of the upload-included leg (the pool from pinned host memory, copy overlapped "
                                                                  "with the rounds in flight); -1 = as many as the timed region, 0 = skip")
    ap.add_argument("--min-seconds", type=float, default=5.0, help="the job list of --steps scans is repeated until the timed region lasts at least this long")
    args = ap.parse_args()
    global FULL_LINE
    FULL_LINE = bool(args.full_line)

    if args.config == "refparity":  # child process of the metric config's parity leg: CPU only
        return ref_parity_leg(args.parity_dir)
    if args.config == "rcclprobe":  # child process of rccl_probe: one rank of the C ABI's communicator, nothing of torch.distributed in it
        return rccl_probe_child(args.probe)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher -- one process per GPU under torch.distributed.run (RCCL / gloo
        # rendezvous on 127.0.0.1), this process only waits for them and hands their exit code back
        return spawn_ranks(args.gpus)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); refusing to report a number for another GPU count")
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if args.dry_run else "nccl", rank=rank, world_size=world)
    if args.dry_run:
        return dry_run(args, dist, world, rank, local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the LIO hot path has no CPU fallback)")
    if torch.cuda.device_count() < world and "LIO_BENCH_SHARE_GPU" not in os.environ:
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s): one process per GPU")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from lsd_amd import lio, synth

    if args.config == "merge":
        return bench_merge(args, torch, dist, world, rank, local_rank, dev)
    if args.config == "stream" and args.tf_pinned:
        return stream_tf_pinned(args, torch, local_rank)
    if args.config == "stream":
        return bench_stream(args, torch, local_rank)
    if args.config == "localize":
        return bench_localize(args, torch, local_rank)
    if args.config == "sequences":
        return bench_sequences(args, torch, local_rank, dev)

    # ---- synthetic workload (SURVEY.md section 8d, config 2 scaled to the metric's 1e7-point map) -------------
    # map and scans are generated ON the GPU (lsd_amd/synth_gpu.py: the same scene and ray model as synth.py, torch's random streams) and copied
    # to the host for the CPU baselines: numpy needs 27 s for the 1e7 surface samples and 0.7 s per scan, i.e. minutes for a pool of 128
    from lsd_amd import synth_gpu

    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
    d_map = synth_gpu.sample_surface(scene, args.map_points, dev, seed=2, sigma=0.01)
    if args.frame_z:
        d_map[:, 2] -= float(np.float32(args.frame_z))
    scanner = synth_gpu.StaticScanner(scene, dev, n_az=args.n_az, fov_deg=(-24.8, 2.0), max_range=150.0)

    def make_pool(n_scans, spread, seed0):
        """n_scans scans with their true poses and priors.  spread: sensor positions uniform over [-spread, spread]^2 (outside the boxes, 1.5 m
        clear), any yaw -- SURVEY 8d config 2: distinct seeds seed0 .. seed0 + n_scans - 1; the prior is within --prior-t / --prior-deg of the truth"""
        rng = np.random.default_rng(seed0 + 7919 * rank)
        pool = []
        for k in range(n_scans):
            while True:
                xy = rng.uniform(-spread, spread, 2)
                if not np.any((scene.lo[:, 0] - 1.5 < xy[0]) & (xy[0] < scene.hi[:, 0] + 1.5) & (scene.lo[:, 1] - 1.5 < xy[1]) & (xy[1] < scene.hi[:, 1] + 1.5)):
                    break
            pos = np.array([xy[0], xy[1], 1.8])
            q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
            d = scanner.scan(pos, q, seed=seed0 + 100000 * rank + k)
            pos = pos - np.array([0.0, 0.0, float(np.float32(args.frame_z))])  # (the scan is body-frame data: only the pose moves with the frame)
            gp, gq = synth.perturb_pose(pos, q, seed=seed0 + 7 * k + rank, max_t=args.prior_t, max_deg=args.prior_deg)
            pool.append(dict(raw=d.cpu().numpy(), d=d, pos=pos, q=q, guess=synth.state_from_pose(gp, gq), seed=seed0 + 100000 * rank + k))
        return pool

    scans = make_pool(args.scan_pool, args.spread, args.seed)           # the timed workload: poses all over the 200 m map
    scans8 = make_pool(8, 4.0, args.seed + 500) if (rank == 0 and args.secondary) else []  # round 3's workload (8 scans within 4 m of one spot), timed beside it
    n_raw = int(np.mean([len(s["raw"]) for s in scans]))

    the_map = lio.Map(resolution=0.5, stencil=19, max_points=max(args.map_points, 1_000_000), max_voxels=max(args.map_points // 4, 1_000_000),
                      device=local_rank)
    # the map goes to HBM once; the raw scans live in torch tensors on the device (inputs resident before timing)
    torch.cuda.synchronize()
    the_map.add_device(d_map.data_ptr(), args.map_points)
    map_pts = d_map.cpu().numpy() if (rank == 0 and world == 1 and (args.cpu_scans > 0 or args.ref_scans > 0)) else None  # the CPU baselines' copy
    del d_map
    d_scans = [s["d"] for s in scans]
    torch.cuda.synchronize()
    n_streams = args.streams
    if n_streams <= 0:  # default: 12 scans in flight per GPU, fewer when the ranks of this node have to share few host CPUs
        n_streams = max(2, min(12, usable_cpus() // max(world, 1) - 1))
    if args.engine == "batch":
        n_streams = 1  # one per-scan engine for the latency / parity legs; the timed region runs on the batched engine
    engines = [lio.Engine(max_raw=1 << 18, max_ds=100000, shared_map=the_map) for _ in range(n_streams)]
    batch = lio.Batch(the_map, n_slots=args.slots, n_groups=args.groups, max_raw=1 << 17, max_ds=100000) if args.engine == "batch" else None

    def run_jobs(jl):
        return batch.process(jl) if batch is not None else lio.process_batch(engines, jl)
    for e in engines:
        e.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)
    eng = engines[0]
    P0 = lio.init_cov()
    map_points, map_voxels = the_map.stats()

    def step(i, e=None):
        e = e or eng
        s = scans[i % len(scans)]
        e.set_state(s["guess"])
        e.set_cov(P0)
        rc = e.process_scan_device(d_scans[i % len(scans)].data_ptr(), len(s["raw"]), 1.0 + 0.1 * i)
        if rc != 3:
            raise RuntimeError(f"process_scan returned {rc}")

    # Setup, untimed: push enough launches through every engine's stream for the HIP runtime to finish growing its
    # per-queue pools -- a one-off ~35 ms stall shows up after roughly 190 scans (~5 000 kernel launches) of a fresh
    # process and never again (tools/experiments/README.md); a service hits it once at start-up.
    prime = [dict(dptr=d_scans[i % len(scans)].data_ptr(), n=len(scans[i % len(scans)]["raw"]), t=1.0 + 0.1 * i, state=scans[i % len(scans)]["guess"],
                  cov=P0) for i in range(40 * n_streams)]
    run_jobs(prime)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i, engines[i % n_streams])
    # pose check outside the timed region: every pooled scan must land on its true pose
    pose_err, ang_err = 0.0, 0.0
    for k in range(len(scans)):
        step(k)
        st = eng.get_state()
        pose_err = max(pose_err, float(np.linalg.norm(st[:3] - scans[k]["pos"])))
        ang_err = max(ang_err, float(synth.quat_angle(st[3:7], scans[k]["q"])))

    # single-stream latency of one scan (reported beside the throughput; not the timed region)
    torch.cuda.synchronize()
    l0 = time.perf_counter()
    for i in range(20):
        step(i)
    latency_ms = 1e3 * (time.perf_counter() - l0) / 20
    if batch is None:
        for e in engines:
            e.scan.enable_kernel_timing(1)  # the dominant kernel only: two event records per kNN launch in the timed region
            e.scan.kernel_times(reset=True)
    acc = dict(n_ds=0, n_pass=0, n_knn=0, pts=0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the timed region is ONE C-ABI call: the K = --steps independent scans (each with its own initial state / covariance), handed to
    # the device in batches by C++ (no Python in the loop).  K scans take a few milliseconds; so that the clock is not measuring
    # start-up effects the same list is repeated R times inside the call (R from an untimed calibration pass), ms_per_step = t / (K R).
    def job_of(i, pool=None):
        s = (pool or scans)[i % len(pool or scans)]
        return dict(dptr=s["d"].data_ptr(), n=len(s["raw"]), t=1.0 + 0.1 * i, state=s["guess"], cov=P0)

    jobs = [job_of(i) for i in range(args.steps)]
    # ... and of the same scans one at a time through a ONE-slot batch: the whole registration (downsample chain, five passes of search / linearise /
    # filter step on the device) as one captured graph and one hipGraphLaunch, against the host-driven per-pass loop above
    latency_graph_ms = None
    if rank == 0 and world == 1 and args.secondary:  # (not in the short forms the profiling passes run: its one-slot launches would enter their per-kernel averages)
        try:
            b1 = lio.Batch(the_map, n_slots=1, n_groups=1, max_raw=1 << 17, max_ds=100000)
            for i in range(4):
                b1.process([job_of(i)])
            l0 = time.perf_counter()
            for i in range(20):
                b1.process([job_of(i)])
            latency_graph_ms = 1e3 * (time.perf_counter() - l0) / 20
            del b1
        except Exception:
            latency_graph_ms = None

    # calibration pass: long enough to fill every round in flight several times (a list shorter than slots x groups runs un-pipelined and
    # over-estimates the time per scan: round 4's first line had a 1.4 s region for --min-seconds 5)
    n_cal = max(8 * args.steps, 6 * args.slots * args.groups if batch is not None else 8 * n_streams)
    cal = lio.PreparedJobs([job_of(i) for i in range(n_cal)])
    lio.run_prepared(cal, engines=engines, batch=batch)  # (once untimed: graphs instantiated, pools grown)
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    lio.run_prepared(cal, engines=engines, batch=batch)
    torch.cuda.synchronize()
    t_cal = max(time.perf_counter() - c0, 1e-6) * args.steps / n_cal  # seconds per --steps scans
    repeats = max(1, int(np.ceil(1.1 * args.min_seconds / t_cal)))
    if dist is not None:  # the same R on every rank
        tr = torch.tensor([float(repeats)], device=dev, dtype=torch.float64)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        repeats = int(tr.item())
    # K x R jobs walking through the WHOLE pool (job i = scan i mod pool): with --steps 20 the list used to be the first 20 scans repeated R times
    timed_jobs = [job_of(i) for i in range(args.steps * repeats)]
    prep = lio.PreparedJobs(timed_jobs)  # marshalled into the C ABI's job array BEFORE the clock starts: the timed region is the one C call
    cand0 = the_map.knn_candidates
    barrier()
    t0 = time.perf_counter()
    rc = lio.run_prepared(prep, engines=engines, batch=batch)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    results = prep.results()
    if rc != 0 or any(r["rc"] != 3 for r in results):
        raise RuntimeError(f"process_batch failed: {rc} {[r['rc'] for r in results][:8]}")
    for i, r in enumerate(results):
        acc["n_ds"] += r["n_ds"]
        acc["n_pass"] += r["n_pass"]
        acc["n_knn"] += r["n_knn_pass"]
        acc["pts"] += len(scans[i % len(scans)]["raw"])
    n_timed = len(timed_jobs)
    barrier()
    t_max = t_local
    total_pts = acc["pts"]
    if dist is not None:
        tt = torch.tensor([t_local], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
        tp = torch.tensor([float(acc["pts"])], device=dev, dtype=torch.float64)
        dist.all_reduce(tp, op=dist.ReduceOp.SUM)
        total_pts = float(tp.item())

    cand = the_map.knn_candidates - cand0
    S = 19
    # ---- the same workload with the upload INCLUDED (t_scan upload excluded and included, both reported): the pool's clouds in
    # page-locked HOST memory, every job LIO_JOB_HOST_RAW -- the library copies a round's clouds to HBM on the round's stream, beside the kernels of the
    # other rounds in flight (the reference's path starts with this copy: slam/src/py_utils.cpp:149-181, slam_wrapper.cpp:64-84).  Same C call, same jobs.
    upload = None
    if batch is not None and args.upload_scans != 0:
        try:
            pinned = [lio.PinnedCloud(sc["raw"]) for sc in scans]

            def job_host(i):
                sc, pc = scans[i % len(scans)], pinned[i % len(scans)]
                return dict(dptr=pc.ptr, n=pc.n, t=1.0 + 0.1 * i, state=sc["guess"], cov=P0, flags=lio.JOB_HOST_RAW)

            n_up = args.upload_scans if args.upload_scans > 0 else int(min(n_timed, max(4 * args.slots * args.groups, np.ceil(2.5 / max(t_local / n_timed, 3.5e-5)))))
            lio.run_prepared(lio.PreparedJobs([job_host(i) for i in range(2 * args.slots * args.groups)]), batch=batch)  # (raw rings allocated, untimed)
            prep_up = lio.PreparedJobs([job_host(i) for i in range(n_up)])
            barrier()
            u0 = time.perf_counter()
            rc_up = lio.run_prepared(prep_up, batch=batch)
            torch.cuda.synchronize()
            t_up = time.perf_counter() - u0
            res_up = prep_up.results()
            same_up = rc_up == 0 and all(r["rc"] == 3 and np.array_equal(r["state"], results[i % len(scans)]["state"]) for i, r in enumerate(res_up))
            bytes_up = float(sum(16 * len(scans[i % len(scans)]["raw"]) for i in range(n_up)))
            if dist is not None:
                tu = torch.tensor([t_up], device=dev, dtype=torch.float64)
                dist.all_reduce(tu, op=dist.ReduceOp.MAX)
                t_up = float(tu.item())
            upload = {"ms_per_step": round(1e3 * t_up / n_up, 4), "value": round(world * bytes_up / 16.0 / t_up, 1), "unit": "points/s", "timed_scans": n_up,
                      "timed_seconds": round(t_up, 4), "host_bytes_per_scan": int(bytes_up / n_up), "pcie_GBps": round(bytes_up / t_up / 1e9, 2),
                      "parity_ok": bool(same_up),
                      "what": "the timed region's jobs with the clouds in pinned host memory (lio_pinned_alloc) and LIO_JOB_HOST_RAW: hipMemcpyAsync of a round's "
                              "clouds on the round's stream, overlapped with the other rounds in flight; results bit-identical to the resident run's (parity_ok)"}
            del prep_up, pinned
        except Exception as ex:  # the headline must not depend on this leg
            upload = {"error": repr(ex)[-300:]}
    n_ds_avg = acc["n_ds"] / n_timed
    # ---- roofline of the dominant kernel (stencil kNN) ------------------------------------------------------------------------------
    # algorithmic bytes [SURVEY.md 8d]: B_knn = N_ds * (16 query + 16 * S slot probes) + 16 * (points resident in the probed voxels), per
    # scan and neighbour-search pass; the kernel's time from HIP events recorded on the streams it is launched on.  Timed OUTSIDE the
    # timed region (event records cost host time) with the device to the kernel itself -- one round in flight -- which is what
    # rocprofv3's per-kernel duration of the same command measures as well.
    others = {}

    def solo_leg(map_, sj):
        """one round in flight on its own batch object, HIP events around every kernel class (lio_batch_enable_kernel_timing), then the same jobs
        through the kNN kernel's counting variant: the figures of the dominant kernel's roofline for the jobs `sj` against `map_`"""
        solo = lio.Batch(map_, n_slots=args.slots, n_groups=1, max_raw=1 << 17, max_ds=100000)
        solo.process(sj[: 2 * args.slots])  # warm
        solo.enable_kernel_timing(True)
        solo.kernel_times(reset=True)
        c1 = map_.knn_candidates
        rc_s, res_s = solo.process(sj)
        kt = solo.kernel_times(reset=True)
        solo.enable_kernel_timing(False)
        n_search = sum(r["n_ds"] * r["n_knn_pass"] for r in res_s)  # queries, all searches
        searches = max(sum(r["n_knn_pass"] for r in res_s), 1)
        cand_pts = map_.knn_candidates - c1
        launches = max(int(kt["knn_launches"]), 1)
        rounds = max(int(kt["downsample_launches"]), 1)
        leg = {"us": kt["knn_us"] / launches, "launches": launches, "bytes": (n_search * (16 + 16 * S) + 16.0 * cand_pts) / launches,
               "queries_per_launch": n_search / launches, "candidates_per_query": cand_pts / max(n_search, 1),
               "others": {"downsample_chain_per_round": round(kt["downsample_us"] / rounds, 2),
                          "linearize_per_launch": round(kt["linearize_us"] / max(int(kt["linearize_launches"]), 1), 2),
                          "filter_pass_per_launch": round(kt["step_us"] / max(int(kt["step_launches"]), 1), 2),
                          "knn_per_scan_and_search": round(kt["knn_us"] / searches, 2),
                          "device_time_per_scan_one_round_in_flight": round((kt["downsample_us"] + kt["knn_us"] + kt["linearize_us"] + kt["step_us"]) / len(sj), 2)}}
        # the same jobs once more through the COUNTING variant of the kernel: the candidate points the pruned sweep really loads ("touched")
        solo.enable_kernel_timing(2)
        solo.kernel_times(reset=True)
        t0c, u0c = map_.knn_touched, map_.knn_unique
        solo.process(sj)
        kt2 = solo.kernel_times(reset=True)
        solo.enable_kernel_timing(False)
        leg["touched_bytes"] = (n_search * (16 + 16 * S) + 16.0 * (map_.knn_touched - t0c)) / max(int(kt2["knn_launches"]), 1)
        # UNIQUE bytes of a launch: every query's own point and stencil slots once, every DISTINCT candidate point the launch loads once (the
        # counting variant's bitmap over the pool, cleared before each launch) -- what the launch has to move at least once whatever its caches do
        leg["unique_bytes"] = (n_search * (16 + 16 * S) + 16.0 * (map_.knn_unique - u0c)) / max(int(kt2["knn_launches"]), 1)
        leg["unique_points_per_launch"] = (map_.knn_unique - u0c) / max(int(kt2["knn_launches"]), 1)
        del solo
        return leg

    def knn_roofline(leg, traffic_file):
        """the roofline object of the batched kNN kernel from a solo_leg.  `frac` is a UTILISATION: bytes that reach the memory side (PMC FETCH_SIZE x 2 +
        WRITE_SIZE per launch, profiles/<traffic_file>, collected in their own rocprofv3 --pmc passes of this workload) over the kernel's live HIP-event
        time over the 8 TB/s peak; without a PMC file of this workload, the bytes the kernel's loads REQUEST (its counting variant on the same jobs), an
        upper bound of what reaches HBM.  The reference algorithm's bytes (every point of the 19 stencil voxels of every query) are credit
        for work the exactly pruned sweep does not do: they stand beside it as frac_algorithmic and may exceed 1.  frac_valu = the kernel's VALU
        wave-instructions per launch over the chip's MEASURED issue rate (tools/valu_peak) over the same time."""
        us = leg["us"]
        per_s = 1.0 / (us * 1e-6) if us > 0 else 0.0
        alg = leg["bytes"] * per_s / 1e9
        touched = leg["touched_bytes"] * per_s / 1e9
        unique = leg.get("unique_bytes", 0.0) * per_s / 1e9
        traffic, valu_per_wave, valu_src = None, KNN_VALU_PER_WAVE_STATIC, "static ISA count (llvm-objdump of knn.o, loop body at the average trip counts)"
        valu_per_launch = None
        tpath = os.path.join(ROOT, "profiles", traffic_file)
        if os.path.exists(tpath):  # HBM bytes per launch + VALU instructions per wave from the PMC passes (tools/pmc_traffic.py, its own rocprofv3 --pmc runs)
            try:
                tj = json.load(open(tpath))
                if tj.get("slots_per_launch", args.slots) == args.slots and tj.get("scan_pool", args.scan_pool) == args.scan_pool:
                    traffic = tj.get("hbm_bytes_per_launch")
                    if tj.get("valu_wave_instructions_per_launch"):
                        valu_per_launch = float(tj["valu_wave_instructions_per_launch"])
                        valu_src = "PMC: SQ_INSTS_VALU per launch of this kernel on the same workload (profiles/%s, its own rocprofv3 --pmc pass)" % traffic_file
            except Exception:
                traffic = None
        waves = leg["queries_per_launch"] / 4.0  # sixteen lanes per query: four queries per wave-pass (a launched wave takes several in turn)
        measured = valu_src.startswith("PMC")
        if valu_per_launch is None:
            valu_per_launch = waves * valu_per_wave
        vp = measured_valu_peak(local_rank)
        peak_rate = vp["wave_insts_per_s"] if vp else None
        # (the static count is the metric map's: ~300 candidates per query; a sparser map sweeps fewer voxels per query, so without a PMC count of THIS
        # workload the figure is an upper bound and no fraction is formed from it)
        valu_ok = us > 0 and peak_rate and (measured or abs(leg["candidates_per_query"] - 300.0) < 60.0)
        frac_valu = round(valu_per_launch / peak_rate / (us * 1e-6), 4) if valu_ok else None
        mem = traffic * per_s / 1e9 if traffic else touched
        return dict(bound="hbm", limited_by="latency / VALU issue (no MFMA on this path): frac_valu beside the byte fractions",
                    kernel="knn_batch_kernel<2, false> (16 lanes per query, %d scans per launch)" % args.slots,
                    achieved=round(mem, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(mem / HBM_PEAK_GBS, 4),
                    frac_basis=("pmc: FETCH_SIZE x 2 + WRITE_SIZE per launch (profiles/%s) over the live HIP-event time" % traffic_file) if traffic
                               else "bytes the kernel's loads request (counting variant, same jobs): no PMC pass of this workload",
                    traffic=traffic,
                    achieved_algorithmic=round(alg, 1), frac_algorithmic=round(alg / HBM_PEAK_GBS, 4),
                    touched_bytes_per_launch=int(leg["touched_bytes"]), frac_touched=round(touched / HBM_PEAK_GBS, 4),
                    unique_bytes_per_launch=int(leg.get("unique_bytes", 0)), frac_unique=round(unique / HBM_PEAK_GBS, 4),
                    unique_candidate_points_per_launch=int(leg.get("unique_points_per_launch", 0)),
                    frac_hbm_traffic=(round(traffic * per_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
                    frac_valu=frac_valu, valu_peak_wave_insts_per_s=(round(peak_rate, 0) if peak_rate else None),
                    valu_wave_insts_per_launch=round(valu_per_launch, 0),
                    valu={"wave_instructions_per_launch": round(valu_per_launch, 0), "per_four_queries": round(valu_per_launch / max(waves, 1.0), 1),
                          "source": valu_src, "four_query_units_per_launch": round(waves, 1), "peak_measured": vp,
                          "issue_bound_us": (round(1e6 * valu_per_launch / peak_rate, 2) if peak_rate else None),
                          "frac_of_valu_issue_peak": frac_valu},
                    algorithmic_bytes_per_launch=int(leg["bytes"]), avg_launch_us=round(us, 2), launches=leg["launches"],
                    candidates_per_query=round(leg["candidates_per_query"], 1),
                    note="frac = memory-side bytes (PMC) or requested bytes over the kernel's time over 8 TB/s: a utilisation.  frac_unique = the bytes a launch "
                         "must move at least once (its queries, their stencil slots, every DISTINCT candidate point it loads: measured by the counting "
                         "variant's bitmap) over the same time: the floor of the traffic -- frac / frac_unique says how often a byte is re-fetched.  "
                         "frac_algorithmic = the reference "
                         "algorithm's bytes (every point of the 19 stencil voxels of every query) over the same time: credit for bytes the pruned sweep does "
                         "not read, may exceed 1.  frac_touched = the bytes the exactly pruned sweep asks for.  frac_valu = VALU wave-instructions per launch "
                         "over the measured issue rate (tools/valu_peak/valu_peak.hip, run in this process)")

    if batch is not None:
        leg = solo_leg(the_map, [job_of(i) for i in range(max(args.slots * 8, len(scans)))])
        iso_us, iso_bytes, iso_launches, others, touched_bytes = leg["us"], leg["bytes"], leg["launches"], leg["others"], leg["touched_bytes"]
        timed_region = None
    else:
        kt = dict(knn_us=0.0, linearize_us=0.0, finalize_us=0.0, knn_launches=0, linearize_launches=0, finalize_launches=0)
        for e in engines:
            k1 = e.scan.kernel_times(reset=True)
            for k in kt:
                kt[k] += k1[k]
            e.scan.enable_kernel_timing(0)
        eng.scan.enable_kernel_timing(1)
        eng.scan.kernel_times(reset=True)
        cand1 = the_map.knn_candidates
        for i in range(32):
            step(i)
        k1 = eng.scan.kernel_times(reset=True)
        iso_launches = max(int(k1["knn_launches"]), 1)
        iso_us = k1["knn_us"] / iso_launches
        iso_bytes = n_ds_avg * (16 + 16 * S) + 16.0 * (the_map.knn_candidates - cand1) / iso_launches
        eng.scan.enable_kernel_timing(2)
        eng.scan.kernel_times(reset=True)
        for i in range(16):
            step(i)
        k2 = eng.scan.kernel_times(reset=True)
        eng.scan.enable_kernel_timing(0)
        others = {"linearize+report": round(k2["linearize_us"] / max(k2["linearize_launches"], 1), 2)}
        launches = max(int(kt["knn_launches"]), 1)
        knn_bytes = n_ds_avg * (16 + 16 * S) + 16.0 * cand / launches
        knn_us = kt["knn_us"] / launches
        shared = knn_bytes / (knn_us * 1e-6) / 1e9 if knn_us > 0 else 0.0
        timed_region = {"avg_launch_us": round(knn_us, 2), "launches": launches, "achieved": round(shared, 1), "frac": round(shared / HBM_PEAK_GBS, 4),
                        "streams": n_streams}
        kernel_name = "knn_kernel<2, 0> (16 lanes per query)"
    achieved = iso_bytes / (iso_us * 1e-6) / 1e9 if iso_us > 0 else 0.0
    # ---- the whole scan against the roofline, as SURVEY.md 8d defines it: B_scan = B_ds + n_knn B_knn + n_pass B_lin (+ B_ins, none against a
    # static map) over the scan's wall time in the timed region ----
    n_pass_avg, n_knn_avg = acc["n_pass"] / n_timed, acc["n_knn"] / n_timed
    b_ds = 16.0 * n_raw + 16.0 * n_ds_avg
    b_knn = n_ds_avg * (16 + 16 * S) + 16.0 * cand / max(acc["n_knn"], 1)
    b_lin = n_ds_avg * (16 + 5 * 16) + 16.0 * n_ds_avg + 8 * 32 * np.ceil(n_ds_avg / 64)
    b_scan = b_ds + n_knn_avg * b_knn + n_pass_avg * b_lin
    # the device's own copy rate (report the fraction of the nominal AND of a measured peak): 1 GiB device-to-device copies,
    # read + write counted, torch events on torch's stream (nothing of the hot path is in flight here)
    copy_peak = None
    try:
        nbytes = 1 << 30
        src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        for _ in range(3):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_peak = round(2.0 * nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del src, dst
    except Exception:
        copy_peak = None
    t_scan = t_max / n_timed
    if batch is not None:
        roofline = knn_roofline(leg, "knn_batch_traffic.json")
    else:
        roofline = dict(bound="hbm", kernel=kernel_name, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=None, algorithmic_bytes_per_launch=int(iso_bytes), avg_launch_us=round(iso_us, 2), launches=iso_launches, per_stream=timed_region)
    roofline.update(measured_copy_peak=copy_peak, frac_of_measured_copy_peak=(round(roofline["achieved"] / copy_peak, 4) if copy_peak else None),
                    timed_region=round(t_max, 4), timed_region_s=round(t_max, 4), other_kernels_us=others,
                    whole_scan={"algorithmic_bytes_per_scan": int(b_scan), "seconds_per_scan": t_scan, "credit_GBps": round(b_scan / t_scan / 1e9, 1),
                                "credit_over_peak": round(b_scan / t_scan / 1e9 / HBM_PEAK_GBS, 4),
                                "what": "SURVEY 8d's bytes of the REFERENCE algorithm per scan (every point of the 19 stencil voxels of every query counted) over "
                                        "the scan's wall time: credit for work the exactly pruned, cache-shared sweep does not do -- NOT a bandwidth utilisation (the "
                                        "kernels' own are roofline.frac and configs.*.roofline)",
                                "terms": {"B_ds": int(b_ds), "B_knn": int(b_knn), "n_knn": round(n_knn_avg, 2), "B_lin": int(b_lin), "n_pass": round(n_pass_avg, 2),
                                          "B_ins": 0, "note": "B_ins = 0: the metric's map is static (BASELINE config 2 / the headline: independent scans against a "
                                                              "fixed map, no map_incremental); the insert is timed in configs.config3_* and configs.sequence_batch"}})

    # ---- CPU baseline: the oracle restatement of the same path on a bounded sample of the same workload ---------
    cpu = None
    batch_vs_oracle = None
    if rank == 0 and world == 1 and args.cpu_scans > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle  # test infrastructure; used here only as the timed CPU baseline / checker

        threads = min(8, usable_cpus())  # the reference parallelises the kNN loop over MP_PROC_NUM = 8 threads; fewer if the box has fewer
        o = oracle.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=threads)
        o.map_add(map_pts)
        o.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)
        t_cpu, pts_cpu, worst_dp, worst_da = 0.0, 0, 0.0, 0.0
        batch_dp, batch_da, batch_ds, batch_checked, pass_mismatch = 0.0, 0.0, 0.0, 0, []
        for i in range(args.cpu_scans):
            s = scans[i % len(scans)]
            parity = i < len(scans)
            if parity:  # equal histories: the neighbour cache (Nearest_Points) persists across scans on both sides
                o.reset_cache()
                eng.scan.reset()
            o.set_state(s["guess"])
            o.set_cov(P0)
            c0 = time.perf_counter()
            ds = oracle.voxel_downsample(s["raw"], 0.5)
            o.set_ds(ds)
            lo_passes = o.update()
            t_cpu += time.perf_counter() - c0
            pts_cpu += len(s["raw"])
            if parity:  # full-size parity: GPU pose vs oracle pose on the same scan
                step(i)
                sg, so = eng.get_state(), o.get_state()
                worst_dp = max(worst_dp, float(np.linalg.norm(sg[:3] - so[:3])))
                worst_da = max(worst_da, float(synth.quat_angle(sg[3:7], so[3:7])))
                if batch is not None and i < len(results):
                    # ... and the TIMED path itself: the state the batched engine returned for this scan inside the timed region (job i of the timed
                    # list = scan i of the pool; jobs are independent scans, so every later repeat of it must carry the same bits -- checked below)
                    rb = results[i]
                    if (rb["n_pass"], rb["n_knn_pass"]) != (len(lo_passes), sum(p["knn"] for p in lo_passes)):
                        pass_mismatch.append([i, rb["n_pass"], rb["n_knn_pass"], len(lo_passes), sum(p["knn"] for p in lo_passes)])
                    batch_dp = max(batch_dp, float(np.linalg.norm(rb["state"][:3] - so[:3])))
                    batch_da = max(batch_da, float(synth.quat_angle(rb["state"][3:7], so[3:7])))
                    batch_ds = max(batch_ds, float(np.abs(rb["state"] - so).max()))
                    batch_checked += 1
        batch_vs_oracle = None
        if batch is not None:
            same = all(np.array_equal(results[i]["state"], results[i % len(scans)]["state"]) for i in range(len(results)))
            batch_vs_oracle = {"max_dpos_m": batch_dp, "max_drot_rad": batch_da, "max_dstate": batch_ds, "scans_checked": batch_checked,
                               "all_timed_results_bit_identical_to_the_checked_ones": bool(same), "timed_results": len(results),
                               "pass_structure_mismatches": pass_mismatch, "parity_ok": bool(same and batch_ds <= 1e-9 and not pass_mismatch),
                               "note": f"state_out of the timed lio_batch_process call itself ({args.slots} slots x {args.groups} rounds in flight, one hipGraphLaunch per round) against the "
                                       "oracle's registration of the same scan; same pass / search counts required"}
            if not batch_vs_oracle["parity_ok"]:  # reported in the line (parity_ok: false) and on stderr; the measurement itself stands
                print(f"bench.py: PARITY FAILURE -- batched engine differs from the oracle on the timed jobs: {batch_vs_oracle}", file=sys.stderr)
        port = dict(value=round(pts_cpu / t_cpu, 1), unit="points/s", cores=threads, kind="port",
                    sample=f"{args.cpu_scans} scans of the same workload (oracle/lio_oracle.cpp: VoxelGrid + iVox kNN on {threads} OpenMP threads + "
                           f"esti_plane + iterated ESKF, rest single-threaded as in the reference), {t_cpu:.1f} s",
                    ms_per_scan=round(1e3 * t_cpu / args.cpu_scans, 2),
                    gpu_vs_oracle_pose={"max_dpos_m": worst_dp, "max_drot_rad": worst_da}, batch_vs_oracle_pose=batch_vs_oracle)
        cpu = port
        # ---- the reference's OWN code on the same workload: laserMapping.cpp / iVox / IKFoM compiled from /root/reference with the
        # flags of its CMakeLists.txt (oracle/ref_fastlio.cpp, prebuilt into oracle/_ref by build(); travels to the GPU box) -----
        import ref_fastlio  # oracle/ref_fastlio.py

        if args.ref_scans > 0 and ref_fastlio.available(release=True):
            del o
            ref_fastlio.use_release_build()
            R = ref_fastlio.RefFastLio()
            R.set_logging(False)
            R.map_add(map_pts)
            R.set_nearby(18)
            t_ref, pts_ref, ref_dp, ref_da = 0.0, 0, 0.0, 0.0
            per_scan = []  # (|dpos|, |drot|, GPU state) against the reference's own code, scan by scan
            for i in range(args.ref_scans):
                s = scans[i % len(scans)]
                parity = i < len(scans)
                if parity:
                    R.reset_cache()
                    eng.scan.reset()
                c0 = time.perf_counter()
                rc_ref, sr, _ = R.register(s["raw"], s["guess"], P0)
                t_ref += time.perf_counter() - c0
                pts_ref += len(s["raw"])
                if rc_ref != 3:
                    raise RuntimeError(f"reference registration returned {rc_ref}")
                if parity:  # GPU pose vs the reference's pose (neighbour order and dense-algebra rounding differ: tolerance, not bits)
                    step(i)
                    sg = eng.get_state()
                    ref_dp = max(ref_dp, float(np.linalg.norm(sg[:3] - sr[:3])))
                    ref_da = max(ref_da, float(synth.quat_angle(sg[3:7], sr[3:7])))
                    per_scan.append((float(np.linalg.norm(sg[:3] - sr[:3])), float(synth.quat_angle(sg[3:7], sr[3:7])), sg.copy(), np.array(sr, dtype=np.float64).copy()))
            # `R` is the reference's code built with ITS flags (-O3 -DNDEBUG: Eigen vectorised, the compiler free to contract) -- the build that is
            # timed.  The build the path is PINNED to is the other one (oracle/_ref/libref_fastlio.so: scalar Eigen, no contraction -- DESIGN.md
            # section 4: Eigen's operation order depends on the build, and esti_plane's 5 x 3 QR is ill-conditioned for planes through the map
            # frame's origin, which this scene's ground z = 0 is).  One build per process (both define the reference's file-scope globals): the
            # pinned build registers the same scans in a child process, once as it is (neighbours 1..4 in std::nth_element's order) and once with
            # every search's lists sorted into the oracle's canonical order (ref_fl_set_canonical).
            gvr = {"build": "the reference's own flags (-O3 -DNDEBUG, vectorised Eigen): the build that is timed", "max_dpos_m": ref_dp, "max_drot_rad": ref_da,
                   "scans": len(per_scan)}
            if per_scan:
                dps, das = np.array([p[0] for p in per_scan]), np.array([p[1] for p in per_scan])
                w = int(np.argmax(dps))
                gvr.update(median_dpos_m=float(np.median(dps)), p90_dpos_m=float(np.percentile(dps, 90)),
                           scans_beyond_1e_4_m_or_1e_5_rad=int(np.count_nonzero((dps > 1e-4) | (das > 1e-5))),
                           worst_scan={"index": w, "seed": scans[w]["seed"], "dpos_m": float(dps[w]), "drot_rad": float(das[w]),
                                       "pose_error_vs_truth_m": float(np.linalg.norm(per_scan[w][2][:3] - scans[w]["pos"]))})
                try:
                    import subprocess
                    import tempfile

                    del R
                    m_par = min(len(per_scan), args.parity_scans)
                    # the same scans once more with the neighbour lists in the reference's own ORDER (lio_map_set_tie_mode 2: every query through the
                    # reference's selection, a checker ~100 x the search's cost): what the pinned build is compared with AS IT IS, nothing sorted on either side
                    mode2 = {}
                    try:
                        the_map.set_tie_mode(2)
                        for i in range(m_par):
                            eng.scan.reset()
                            step(i)
                            mode2[f"gpu2_{i}"] = eng.get_state().copy()
                    finally:
                        the_map.set_tie_mode(1)
                    with tempfile.TemporaryDirectory(prefix="lio_bench_parity_") as td:
                        np.save(os.path.join(td, "map.npy"), map_pts)
                        np.savez(os.path.join(td, "scans.npz"), P0=P0, n=m_par, **{f"raw{i}": scans[i]["raw"] for i in range(m_par)},
                                 **{f"guess{i}": scans[i]["guess"] for i in range(m_par)}, **{f"gpu{i}": per_scan[i][2] for i in range(m_par)},
                                 **{f"rel{i}": per_scan[i][3] for i in range(m_par)}, **mode2)
                        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "refparity", "--parity-dir", td], capture_output=True, text=True,
                                            timeout=600)
                    line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
                    if pr.returncode != 0 or not line:
                        raise RuntimeError((pr.stderr or pr.stdout)[-300:])
                    gvr["pinned_build"] = json.loads(line[-1])
                except Exception as ex:
                    gvr["pinned_build"] = {"error": repr(ex)[-300:]}
            cpu = dict(value=round(pts_ref / t_ref, 1), unit="points/s", cores=min(8, usable_cpus()), host_cpus=usable_cpus(), kind="reference",
                       sample=f"{args.ref_scans} scans of the same workload through the reference's own laserMapping.cpp h_share_model + iVox + esekfom "
                              f"update_iterated_dyn_share_modified (oracle/_ref/libref_fastlio_release.so: -O3 -DNDEBUG, MP_EN with MP_PROC_NUM=8 as its "
                              f"CMakeLists.txt sets on x86_64; pcl::VoxelGrid replaced by the oracle's restatement), {t_ref:.1f} s",
                       ms_per_scan=round(1e3 * t_ref / args.ref_scans, 2),
                       gpu_vs_reference_pose=gvr, port=port)

    # ---- secondary configurations (BASELINE.json configs 2 and 3), outside the timed region, reported under `configs` ----
    configs = None
    if rank == 0 and world == 1 and args.secondary and batch is not None:
        configs = {}
        def timed_leg(b_, jl, seconds):
            """jl through the batch b_ once to warm, then repeated for about `seconds`: (ms per scan, points/s, results of the first pass)"""
            rc0, r0 = b_.process(jl)
            if rc0 != 0 or any(r["rc"] != 3 for r in r0):
                raise RuntimeError(f"secondary leg failed: {rc0}")
            pw = lio.PreparedJobs(jl)
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            lio.run_prepared(pw, batch=b_)
            torch.cuda.synchronize()
            reps = max(1, int(np.ceil(seconds / max(time.perf_counter() - w0, 1e-6))))
            pj = lio.PreparedJobs(jl * reps)
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            lio.run_prepared(pj, batch=b_)
            torch.cuda.synchronize()
            dt = time.perf_counter() - w0
            return 1e3 * dt / pj.n, sum(j["n"] for j in jl) * reps / dt, r0, pj.n

        try:
            # round 3's workload beside the headline: 8 scans within 4 m of one spot of the SAME map, same engine object
            j8 = [job_of(i, scans8) for i in range(args.slots * args.groups * 2)]
            ms8, pps8, r8, n8 = timed_leg(batch, j8, 1.5)
            leg8 = solo_leg(the_map, [job_of(i, scans8) for i in range(args.slots * 8)])
            configs["pool8_one_spot"] = {"workload": "round 3's timed workload: 8 distinct scans within +-4 m of the map's centre (everything L2 / Infinity-Cache resident), "
                                                     "beside the headline's pool of %d scans spread over +-%.0f m" % (len(scans), args.spread),
                                         "ms_per_scan": round(ms8, 4), "points_per_s": round(pps8, 1), "scans_timed": n8,
                                         "n_ds_avg": round(float(np.mean([r["n_ds"] for r in r8])), 1), "passes_avg": round(float(np.mean([r["n_pass"] for r in r8])), 2),
                                         "roofline": {k: v for k, v in knn_roofline(leg8, "knn_batch_traffic_pool8.json").items() if k != "note"}}
        except Exception as ex:  # the headline must not depend on the secondary legs
            configs["pool8_one_spot"] = {"error": repr(ex)[-400:]}
        try:
            # config 2: the same 64 x 1875 scans against a 1e6-point map, through the same batched engine
            d2 = synth_gpu.sample_surface(scene, 1_000_000, dev, seed=2, sigma=0.01)
            map2 = lio.Map(resolution=0.5, stencil=19, max_points=1_000_000, max_voxels=1_000_000, device=local_rank)
            torch.cuda.synchronize()
            map2.add_device(d2.data_ptr(), 1_000_000)
            map2_pts = d2.cpu().numpy()
            del d2
            b2 = lio.Batch(map2, n_slots=args.slots, n_groups=args.groups, max_raw=1 << 17, max_ds=100000)
            j2 = [job_of(i) for i in range(max(args.slots * args.groups * 2, len(scans)))]
            ms2, pps2, r2, n2 = timed_leg(b2, j2, 2.0)
            pe2 = max(float(np.linalg.norm(r2[i]["state"][:3] - scans[i % len(scans)]["pos"])) for i in range(len(j2)))
            del b2
            leg2 = solo_leg(map2, [job_of(i) for i in range(max(args.slots * 8, len(scans)))])
            c2 = {"workload": "64x%d scans (the headline's pool of %d) vs 1000000-pt static map, batched engine" % (args.n_az, len(scans)), "ms_per_scan": round(ms2, 4),
                  "points_per_s": round(pps2, 1), "scans_timed": n2,
                  "n_ds_avg": round(float(np.mean([r["n_ds"] for r in r2])), 1), "passes_avg": round(float(np.mean([r["n_pass"] for r in r2])), 2),
                  "pose_error_vs_truth_m": pe2, "roofline": knn_roofline(leg2, "knn_batch_traffic_config2.json"), "cpu_baseline": None}
            c2["roofline"]["other_kernels_us"] = leg2["others"]
            del map2
            if args.ref_scans > 0:  # same-run baseline: the reference's own code on a bounded sample of the same scans against the same 1e6 points
                import ref_fastlio

                if ref_fastlio.available(release=True):
                    ref_fastlio.use_release_build()
                    R2 = ref_fastlio.RefFastLio()
                    R2.set_logging(False)
                    R2.map_add(map2_pts)
                    R2.set_nearby(18)
                    m2 = min(40, args.ref_scans)
                    t_r2, p_r2, e_r2, dps2, das2, rel2 = 0.0, 0, 0.0, [], [], []
                    for i in range(m2):
                        sc2 = scans[i % len(scans)]
                        R2.reset_cache()
                        c0 = time.perf_counter()
                        rc_r2, sr2, _ = R2.register(sc2["raw"], sc2["guess"], P0)
                        t_r2 += time.perf_counter() - c0
                        p_r2 += len(sc2["raw"])
                        rel2.append(np.array(sr2, dtype=np.float64).copy())
                        if rc_r2 == 3:
                            dps2.append(float(np.linalg.norm(r2[i]["state"][:3] - sr2[:3])))
                            das2.append(float(synth.quat_angle(r2[i]["state"][3:7], sr2[3:7])))
                    e_r2 = max(dps2) if dps2 else 0.0
                    gvr2 = {"build": "the reference's own flags (-O3 -DNDEBUG, vectorised Eigen)", "scans": len(dps2), "max_dpos_m": e_r2,
                            "max_drot_rad": max(das2) if das2 else 0.0, "median_dpos_m": float(np.median(dps2)) if dps2 else None,
                            "scans_beyond_1e_4_m_or_1e_5_rad": int(np.count_nonzero((np.array(dps2) > 1e-4) | (np.array(das2) > 1e-5)))}
                    c2["cpu_baseline"] = dict(value=round(p_r2 / t_r2, 1), unit="points/s", cores=min(8, usable_cpus()), host_cpus=usable_cpus(), kind="reference",
                                              sample=f"{m2} scans of the same workload through the reference's own laserMapping.cpp h_share_model + iVox + esekfom update "
                                                     f"(oracle/_ref/libref_fastlio_release.so, MP_PROC_NUM=8) against the same 1e6 map points, {t_r2:.1f} s",
                                              ms_per_scan=round(1e3 * t_r2 / m2, 2), gpu_vs_reference_pose_max_dpos_m=e_r2, gpu_vs_reference_pose=gvr2)
                    del R2
                    try:  # ... and the build the path is PINNED to, in a child process (one build per process), with the release build's poses beside the GPU's
                        import subprocess
                        import tempfile

                        with tempfile.TemporaryDirectory(prefix="lio_bench_parity2_") as td:
                            np.save(os.path.join(td, "map.npy"), map2_pts)
                            np.savez(os.path.join(td, "scans.npz"), P0=P0, n=m2, **{f"raw{i}": scans[i % len(scans)]["raw"] for i in range(m2)},
                                     **{f"guess{i}": scans[i % len(scans)]["guess"] for i in range(m2)}, **{f"gpu{i}": r2[i]["state"] for i in range(m2)},
                                     **{f"rel{i}": rel2[i] for i in range(m2)})
                            pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "refparity", "--parity-dir", td], capture_output=True, text=True,
                                                timeout=600)
                        line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
                        if pr.returncode != 0 or not line:
                            raise RuntimeError((pr.stderr or pr.stdout)[-300:])
                        gvr2["pinned_build"] = json.loads(line[-1])
                    except Exception as ex:
                        gvr2["pinned_build"] = {"error": repr(ex)[-300:]}
            configs["config2_1e6_map"] = c2
        except Exception as ex:  # the headline must not depend on the secondary legs
            configs["config2_1e6_map"] = {"error": repr(ex)[-400:]}
        # configs 3 and 4 run as their own processes (their own maps: 1e7 points grown by map_incremental, a 5e7-point NDT target; this
        # process idles meanwhile, its few GB of HBM do not matter on a 288 GB part); each prints the JSON line
        # `bench.py --config stream|localize` prints, embedded here
        import subprocess

        for key, extra in (("config3_stream_to_1e7_points", ["--config", "stream", "--grow-to", "10000000", "--steps", "6000", "--lru", "0"]),
                           ("config3_stream_lru_1e5_300_sweeps", ["--config", "stream", "--steps", "300", "--lru", "100000", "--ref-scans", "300"]),
                           ("config4_localize_5e7_map", ["--config", "localize", "--steps", "200", "--scan-pool", "32"]),
                           ("config5_merge_8_submaps_1_gpu", ["--config", "merge", "--steps", "256", "--warmup", "64", "--scan-pool", "64", "--min-seconds", "2"]),
                           ("sequence_batch", ["--config", "sequences", "--steps", "24", "--slots", "128", "--groups", "2"])):
            try:
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--full-line", "--seed", str(args.seed)] + extra, capture_output=True, text=True, timeout=900)
                line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
                if pr.returncode != 0 or not line:
                    raise RuntimeError((pr.stderr or pr.stdout)[-400:])
                j = json.loads(line[-1])
                configs[key] = {"ms_per_scan": j["ms_per_step"], "points_per_s": j["value"], **j["config"],
                                "roofline": j.get("roofline"), "cpu_baseline": j.get("cpu_baseline"), "pose_error_vs_truth_m": j.get("pose_error_vs_truth_m")}
                for extra_key in ("drift", "collective", "latency", "knn_on_this_map", "parity", "one_session_at_a_time", "device_us_per_round", "pose_error_vs_truth"):
                    if j.get(extra_key):
                        configs[key][extra_key] = j[extra_key]
            except Exception as ex:  # the headline must not depend on the secondary legs
                configs[key] = {"error": repr(ex)[-500:]}

    # N > 1: the metric's ranks are replicas (no data-path collective); the native communicator of the C ABI (lio_comm_*: RCCL over xGMI, what
    # config 5's joint registration runs on) is brought up once OUTSIDE the timed region and its small-message all-gather timed, so that a
    # multi-GPU run leaves a measured collective latency and the communicator's own rank count in the line
    collective = None
    if dist is not None:
        try:
            collective = rccl_probe(dist, rank, world, local_rank)
        except Exception as ex:  # the probe must never cost the run its line
            collective = {"error": repr(ex)[-300:]}
    if rank == 0:
        value = total_pts / t_max
        out = {
            "metric": "registered points/sec (120k-pt scan vs 1e7-pt map, full iterate-to-converge)",
            "value": round(value, 1), "unit": "points/s", "n_gpus": world, "rccl_ranks": (collective or {}).get("rccl_ranks", 1 if world == 1 else None),
            "collective": collective, "steps": args.steps, "warmup": args.warmup,
            "repeats": repeats, "timed_scans": n_timed, "timed_seconds": round(t_max, 4),
            "ms_per_step": round(1e3 * t_max / n_timed, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 per-point geometry / f64 transforms and reductions", "data": "synthetic",
            "config": {"workload": f"64x{args.n_az} synthetic scan (~{n_raw} pts) vs {map_points}-pt static map ({map_voxels} voxels of 0.5 m), "
                                   "voxel downsample + iterated ESKF update to convergence (map_incremental excluded: the map is static), one scan per step, "
                                   f"scans sharded across GPUs; {len(scans)} distinct scans per GPU, sensor positions uniform over +-{args.spread:.0f} m of the 200 m scene",
                       "scan_pool": len(scans), "scan_seeds": [scans[0]["seed"], scans[-1]["seed"]], "spread_m": args.spread, "frame_z_m": args.frame_z,
                       "n_raw": n_raw, "n_ds_avg": round(n_ds_avg, 1), "passes_avg": round(n_pass_avg, 2),
                       "knn_passes_avg": round(n_knn_avg, 2), "stencil": 19,
                       "knn_candidates_per_query": round(cand / max(n_ds_avg * acc["n_knn"], 1), 1),
                       "map_bytes_hbm": the_map.nbytes,
                       "engine": ("batched: %d scans per launch, %d rounds in flight, filter loop on the device, 1 host thread" % (args.slots, args.groups))
                                 if batch is not None else ("%d engines, one host thread + stream each" % n_streams),
                       "single_stream_latency_ms_per_scan": round(latency_ms, 4),
                       "single_scan_one_graph_latency_ms": (round(latency_graph_ms, 4) if latency_graph_ms else None)},
            "pose_error_vs_truth": {"max_dpos_m": pose_err, "max_drot_rad": ang_err,
                                    "note": "the reference's algorithm itself: at most four ESKF iterations from a prior 0.3 m / 2 deg off; the GPU pose "
                                            "equals the oracle's and the reference's own (cpu_baseline.gpu_vs_*_pose)"},
            "batch_vs_oracle_pose": batch_vs_oracle, "upload_included": upload, "roofline": roofline, "cpu_baseline": cpu, "configs": configs,
        }
        emit(out, "metric")
    if dist is not None:
        if collective and "did not come up" in str(collective.get("error", "")):  # a worker thread is stuck inside ncclCommInitRank: leave without the teardown
            sys.stdout.flush()
            os._exit(0)
        dist.destroy_process_group()


def ref_parity_leg(td):
    """the scans of <td>/scans.npz registered by the PINNED build of the reference's own code (oracle/_ref/libref_fastlio.so) against <td>/map.npy: the
    poses the GPU path returned for them against the reference's, with its neighbour lists as std::nth_element leaves them and in canonical order.
    One JSON line.  Test infrastructure (oracle/) used as the checker, on the host, outside every timed region."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_fastlio
    from lsd_amd import synth

    if not ref_fastlio.available():
        print(json.dumps({"error": "oracle/_ref/libref_fastlio.so is not there"}))
        return
    d = np.load(os.path.join(td, "scans.npz"))
    R = ref_fastlio.RefFastLio()
    R.set_logging(False)
    R.map_add(np.load(os.path.join(td, "map.npy")))
    R.set_nearby(18)
    P0, n = d["P0"], int(d["n"])
    out = {}
    out_dp_canonical = np.zeros(0)
    for mode in ("neighbour_lists_as_nth_element_leaves_them", "neighbour_lists_in_canonical_order"):
        R.set_canonical(mode.endswith("canonical_order"))
        dp, da, bp, ba, dp2, da2 = [], [], [], [], [], []
        for i in range(n):
            R.reset_cache()
            rc, sr, _ = R.register(d[f"raw{i}"], d[f"guess{i}"], P0)
            if rc != 3:
                continue
            g = d[f"gpu{i}"]
            dp.append(float(np.linalg.norm(g[:3] - sr[:3])))
            da.append(float(synth.quat_angle(g[3:7], sr[3:7])))
            if f"gpu2_{i}" in d.files and not mode.endswith("canonical_order"):  # the HIP path with its lists in the reference's order against the untouched reference
                g2 = d[f"gpu2_{i}"]
                dp2.append(float(np.linalg.norm(g2[:3] - sr[:3])))
                da2.append(float(synth.quat_angle(g2[3:7], sr[3:7])))
            if f"rel{i}" in d.files and not mode.endswith("canonical_order"):  # the reference against ITSELF: its release build's pose of this scan against this (pinned) build's
                rl = d[f"rel{i}"]
                bp.append(float(np.linalg.norm(rl[:3] - sr[:3])))
                ba.append(float(synth.quat_angle(rl[3:7], sr[3:7])))
        dp, da = np.array(dp), np.array(da)
        if bp:
            bp, ba = np.array(bp), np.array(ba)
            out["the_references_release_build_against_its_pinned_build"] = {
                "what": "the SAME reference sources built twice (its own CMake flags with vectorised Eigen / scalar Eigen without contraction), the same scans, priors and "
                        "map, both untouched: what the reference moves by when only its build changes -- the resolution at which 'the reference's pose' is defined",
                "scans": int(len(bp)), "max_dpos_m": float(bp.max()), "max_drot_rad": float(ba.max()), "median_dpos_m": float(np.median(bp)),
                "scans_beyond_1e_4_m_or_1e_5_rad": int(np.count_nonzero((bp > 1e-4) | (ba > 1e-5)))}
        if dp2:
            dp2, da2 = np.array(dp2), np.array(da2)
            out["hip_in_tie_mode_2_against_the_pinned_build_as_it_is"] = {
                "what": "lio_map_set_tie_mode(map, 2): the neighbour lists in the reference's own order; the pinned build untouched (nothing sorted on either side)",
                "scans": int(len(dp2)), "max_dpos_m": float(dp2.max()), "max_drot_rad": float(da2.max()), "median_dpos_m": float(np.median(dp2)),
                "scans_beyond_1e_4_m_or_1e_5_rad": int(np.count_nonzero((dp2 > 1e-4) | (da2 > 1e-5)))}
        if mode.endswith("canonical_order"):
            out_dp_canonical = dp
        out[mode] = {"scans": int(len(dp)), "max_dpos_m": float(dp.max()), "max_drot_rad": float(da.max()), "median_dpos_m": float(np.median(dp)),
                     "scans_beyond_1e_4_m_or_1e_5_rad": int(np.count_nonzero((dp > 1e-4) | (da > 1e-5)))}
    R.set_canonical(False)
    out["build"] = "oracle/_ref/libref_fastlio.so: the reference's translation units with scalar Eigen and no FMA contraction -- the build the path is pinned to"
    # What is left in canonical order: queries whose FIFTH-nearest candidate ties with the sixth in f32 squared distance.  The reference keeps whichever
    # std::nth_element leaves (ivox3d_node.hpp:107-127, ivox3d.h:159-164: implementation-defined), oracle and kernels break the tie by (d2, x, y, z):
    # another neighbour SET, which no ordering of the lists repairs.  Shown on the scan that differs most: the first search of the update, oracle
    # (= the GPU path, bit for bit) against the reference, query by query.
    try:
        import oracle

        worst = int(np.argmax(out_dp_canonical)) if len(out_dp_canonical) else -1
        if worst >= 0 and out_dp_canonical[worst] > 1e-9:
            raw, g = d[f"raw{worst}"], d[f"guess{worst}"]
            o = oracle.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=8)
            o.map_add(np.load(os.path.join(td, "map.npy")))
            o.set_flags(ekf_inited=True, first_scan=False, travel=0.0, first_lidar_time=-10.0)
            o.set_state(g)
            o.set_cov(P0)
            o.set_ds(oracle.voxel_downsample(raw, 0.5))
            lo = o.linearize(True)
            wpts = o.get_ds_world()
            R.reset_cache()
            R.register(raw, g, P0)
            R.reset_cache()
            hr = R.h_share(g, converge=2)
            ties = []
            for q in np.nonzero(np.abs(lo["nn"] - hr["nn"]).reshape(len(wpts), -1).max(1) > 0)[0]:
                a = {tuple(r) for r in lo["nn"][q][: lo["nn_cnt"][q], :3].tolist()}
                b = {tuple(r) for r in hr["nn"][q][: hr["nn_cnt"][q], :3].tolist()}
                w = wpts[q].astype(np.float32)

                def d2(pt):
                    e = np.asarray(pt, np.float32) - w[:3]
                    return float(np.float32(e[0] * e[0]) + np.float32(np.float32(e[1] * e[1]) + np.float32(e[2] * e[2])))

                ties.append({"query": int(q), "only_in_oracle_d2": [d2(x) for x in a - b], "only_in_reference_d2": [d2(x) for x in b - a]})
            out["what_is_left_in_canonical_order"] = {
                "scan": worst, "dpos_m": float(out_dp_canonical[worst]), "queries_with_another_neighbour_set_in_the_first_search": len(ties), "their_members": ties[:8],
                "note": "equal f32 squared distances on both sides = a tie at the fifth-nearest boundary, resolved by std::nth_element in the reference "
                        "(implementation-defined) and by the total order (d2, x, y, z) in the oracle and the kernels"}
    except Exception as ex:
        out["what_is_left_in_canonical_order"] = {"error": repr(ex)[-300:]}
    print(json.dumps(out))


def rccl_probe(dist, rank, world, local_rank, n_records=64, iters=200, timeout_s=75.0):
    """bring up lio_comm (ncclCommInitRank through the C ABI) on all ranks and time lio_allgather_records of [n_records x 32] doubles per rank -- the
    per-round, per-pass collective of the batched joint registration (config 5).  Every rank does it in a CHILD process (`--config rcclprobe`: its
    own HIP context on the rank's GPU, nothing of torch.distributed inside): the process that holds the headline never loads a second RCCL beside
    torch's, and a communicator that crashes or hangs -- this path has never met more than one GPU -- costs the run a minute, not its line.  The
    unique id comes from a child of rank 0 and travels through torch.distributed."""
    import subprocess

    me = os.path.abspath(__file__)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    box = [None]
    if rank == 0:
        try:
            r = subprocess.run([sys.executable, me, "--config", "rcclprobe", "--probe", "uid"], capture_output=True, text=True, timeout=60, env=env)
            line = (r.stdout.strip().splitlines() or [""])[-1]
            box = [line if len(line) == 256 else None]
        except Exception:
            box = [None]
    dist.broadcast_object_list(box, src=0)
    if not box[0]:
        return {"error": "no RCCL unique id (librccl could not be loaded by liblio_hip.so?)"}
    job = json.dumps(dict(rank=rank, world=world, device=local_rank, uid=box[0], n_records=n_records, iters=iters))
    try:
        r = subprocess.run([sys.executable, me, "--config", "rcclprobe", "--probe", job], capture_output=True, text=True, timeout=timeout_s, env=env)
        line = (r.stdout.strip().splitlines() or [""])[-1]
        out = json.loads(line) if line.startswith("{") else {"error": ("rc %d: " % r.returncode) + (r.stderr or "")[-300:]}
    except subprocess.TimeoutExpired:
        out = {"error": f"the communicator did not come up within {timeout_s:.0f} s"}
    except Exception as ex:
        out = {"error": repr(ex)[-300:]}
    return out


def rccl_probe_child(spec):
    """--config rcclprobe: "uid" prints a fresh ncclUniqueId as hex; otherwise one rank of the communicator (see rccl_probe), one JSON line"""
    import torch  # (first: liblio_hip.so's lazily loaded librccl then resolves to the copy torch ships and has loaded -- the build this image's RCCL tests ran with)

    from lsd_amd import capi, lio

    if spec == "uid":
        print(lio.Comm.unique_id().hex())
        return
    a = json.loads(spec)
    out = {}
    try:
        torch.cuda.set_device(a["device"])
        dev = torch.device("cuda", a["device"])
        rank, world, n_records, iters = a["rank"], a["world"], a["n_records"], a["iters"]
        comm = lio.Comm(rank=rank, world=world, device=a["device"], uid=bytes.fromhex(a["uid"]))
        lib = capi.lib()
        loc = torch.full((n_records * 32,), float(rank), dtype=torch.float64, device=dev)
        gat = torch.zeros((world * n_records * 32,), dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(20):
            if lib.lio_allgather_records(comm.h, loc.data_ptr(), gat.data_ptr(), n_records, st) != 0:
                raise RuntimeError(lib.lio_last_error().decode())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.lio_allgather_records(comm.h, loc.data_ptr(), gat.data_ptr(), n_records, st)
        e1.record()
        torch.cuda.synchronize()
        heads = gat.view(world, -1)[:, 0].cpu().numpy()
        out.update(rccl_ranks=int(lib.lio_comm_world(comm.h)), backend="RCCL all-gather through lio_allgather_records (librccl loaded by liblio_hip.so), in a child process per rank",
                   bytes_per_rank=n_records * 256, avg_us=round(e0.elapsed_time(e1) * 1e3 / iters, 2), iterations=iters,
                   gathered_in_rank_order=bool(np.array_equal(heads, np.arange(world, dtype=np.float64))))
        comm.close()
    except Exception as ex:
        out["error"] = repr(ex)[-300:]
    print(json.dumps(out))


def bench_sequences(args, torch, local_rank, dev):
    """Throughput WITH map_incremental (SURVEY 8d's B_ins): --slots x --groups independent SLAM sessions, each a short drive
    through the 200 m scene with ITS OWN map grown by map_incremental, registered and inserted round by round through lio_batch_sequences_step
    (one blind submission per group and round: downsample chain, 5 x {kNN against the slot's own map, linearisation, filter pass}, classify +
    AddPoints for all slots).  Clouds resident in HBM; the prior of scan k is the posterior of scan k - 1 moved by the known step (no IMU in
    this leg).  Beside it: the same drives one session at a time through lio_engine_process_scan_device (the single-scan path, host-driven
    loop -- config 3's path with resident clouds), and for four sessions the bit-for-bit comparison with the per-session engine."""
    from lsd_amd import lio, synth, synth_gpu

    B, G = args.slots, args.groups
    n_sess = B * G
    K = max(8, args.steps)
    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
    scanner = synth_gpu.StaticScanner(scene, dev, n_az=args.n_az, fov_deg=(-24.8, 2.0), max_range=150.0)
    rng = np.random.default_rng(args.seed + 31)
    step_len = 1.0  # 10 m/s at 10 Hz
    t_gen = time.perf_counter()
    plans = []
    def clear_of_boxes(xy):
        return not np.any((scene.lo[:, 0] - 1.5 < xy[0]) & (xy[0] < scene.hi[:, 0] + 1.5) & (scene.lo[:, 1] - 1.5 < xy[1]) & (xy[1] < scene.hi[:, 1] + 1.5))

    for s in range(n_sess):
        while True:  # a straight drive that stays 1.5 m clear of every box over all K sweeps (and inside the scene)
            xy = rng.uniform(-70, 70, 2)
            yaw = rng.uniform(-np.pi, np.pi)
            step = step_len * np.array([np.cos(yaw), np.sin(yaw), 0.0])
            end = xy + (K - 1) * step[:2]
            if np.all(np.abs(end) < 90.0) and all(clear_of_boxes(xy + k * step[:2]) for k in range(K)):
                break
        q = synth.quat_from_rotvec([0, 0, yaw])
        scans = []
        for k in range(K):
            pos = np.array([xy[0], xy[1], 1.8]) + k * step
            scans.append(dict(d=scanner.scan(pos, q, seed=args.seed + 1000 * s + k), pos=pos, t=0.1 * k))
        plans.append(dict(scans=scans, s0=synth.state_from_pose(scans[0]["pos"], q), step=step))
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    n_raw = int(np.mean([len(sc["d"]) for p_ in plans for sc in p_["scans"]]))
    P0 = lio.init_cov()
    kw = dict(resolution=0.5, stencil=19, max_points=1_500_000, max_voxels=300_000, max_raw=1 << 17, max_ds=100000, device=local_rank)

    def next_prior(state, cov, step):
        st = np.array(state, dtype=np.float64).copy()
        st[:3] += step
        P = np.array(cov, dtype=np.float64).reshape(23, 23).copy()
        P[:6, :6] += np.eye(6) * 1e-3
        return st, P

    # ---- the sessions as slots of the sequence batch ----
    sb = lio.SequenceBatch(n_slots=B, n_groups=G, **{k: v for k, v in kw.items()})
    priors = [(p_["s0"].copy(), P0.copy()) for p_ in plans]
    states = [[] for _ in range(n_sess)]
    t_round, n_reg = [], []
    rcs_all = []
    for k in range(K):
        jobs = [dict(dptr=plans[s]["scans"][k]["d"].data_ptr(), n=len(plans[s]["scans"][k]["d"]), t=plans[s]["scans"][k]["t"], state=priors[s][0], cov=priors[s][1])
                for s in range(n_sess)]
        sb.load(jobs)
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        rc = sb.run()
        dt = time.perf_counter() - w0
        if rc != 0:
            raise RuntimeError("lio_batch_sequences_step: %d %s" % (rc, lio.capi.lib().lio_last_error().decode()))
        reg = 0
        for s in range(n_sess):
            a = sb.arr[s]
            rcs_all.append(a.rc)
            states[s].append((a.rc, sb.states_out[s].copy(), sb.covs_out[s].copy(), a.n_ds, a.n_pass, a.n_knn_pass))
            if a.rc == 3:
                priors[s] = next_prior(sb.states_out[s], sb.covs_out[s], plans[s]["step"])
                reg += 1
            else:  # nothing registered (time origin, map seed): the sensor moved on all the same
                priors[s] = (priors[s][0] + np.r_[plans[s]["step"], np.zeros(23)], priors[s][1])
        t_round.append(dt)
        n_reg.append(reg)
    full = [i for i in range(K) if n_reg[i] == n_sess]  # rounds in which every session registered + inserted a scan (from the third on)
    if len(full) < 4:
        raise RuntimeError("sequence batch: only %d full rounds" % len(full))
    timed = full[2:]  # two more rounds for the predicted radix passes / first touches to settle
    ms_per_sweep = 1e3 * sum(t_round[i] for i in timed) / (len(timed) * n_sess)
    nds = float(np.mean([states[s][i][3] for s in range(n_sess) for i in timed]))
    npass = float(np.mean([states[s][i][4] for s in range(n_sess) for i in timed]))
    nknn = float(np.mean([states[s][i][5] for s in range(n_sess) for i in timed]))
    pes = [float(np.linalg.norm(states[s][K - 1][1][:3] - plans[s]["scans"][K - 1]["pos"])) for s in range(n_sess)]
    pe = max(pes)
    map_pts = [sb.engine(s).map.stats() for s in range(n_sess)]
    added = [(map_pts[s][0]) for s in range(n_sess)]
    # device time per round by class (HIP events on the groups' streams), from two more rounds of the same sessions standing still at their last pose
    dev_us = None
    try:
        sb.enable_kernel_timing(True)
        for _ in range(2):
            jobs = [dict(dptr=plans[s]["scans"][K - 1]["d"].data_ptr(), n=len(plans[s]["scans"][K - 1]["d"]), t=0.1 * K, state=priors[s][0] - np.r_[plans[s]["step"], np.zeros(23)],
                         cov=priors[s][1]) for s in range(n_sess)]
            sb.load(jobs)
            if sb.run() != 0:
                raise RuntimeError("timed round failed")
        kt = sb.kernel_times()
        sb.enable_kernel_timing(False)
        dev_us = {"downsample_chain": round(kt["downsample_us"] / max(kt["downsample_launches"], 1), 1),
                  "knn_per_launch": round(kt["knn_us"] / max(kt["knn_launches"], 1), 1), "knn_launches_per_round": kt["knn_launches"] / max(kt["downsample_launches"], 1),
                  "linearize_per_launch": round(kt["linearize_us"] / max(kt["linearize_launches"], 1), 1),
                  "filter_pass_per_launch": round(kt["step_us"] / max(kt["step_launches"], 1), 1),
                  "map_incremental": round(kt["insert_us"] / max(kt["insert_launches"], 1), 1), "slots_per_round": B,
                  "note": "HIP events on the round's stream (timed rounds run as plain launches; untimed ones as one graph per group), one round per group in flight"}
    except Exception as ex:
        dev_us = {"error": repr(ex)[-200:]}

    # ---- one session at a time through its own engine: timing (host-driven loop, the default) and, with the device loop, the bits ----
    def solo(s, device_loop, k_max):
        e = lio.Engine(**kw)
        e.set_device_loop(device_loop)
        st, P = plans[s]["s0"].copy(), P0.copy()
        out, ts = [], []
        for k in range(k_max):
            sc = plans[s]["scans"][k]
            e.set_state(st)
            e.set_cov(P)
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            rc = e.process_scan_device(sc["d"].data_ptr(), len(sc["d"]), sc["t"])
            e.flush()
            ts.append(time.perf_counter() - w0)
            out.append((rc, e.get_state(), e.get_cov().reshape(-1)))
            if rc == 3:
                st, P = next_prior(out[-1][1], out[-1][2], plans[s]["step"])
            else:
                st = st + np.r_[plans[s]["step"], np.zeros(23)]
        stats = e.map.stats()
        e.close()
        return out, ts, stats

    n_check = min(4, n_sess)
    identical, worst = True, 0.0
    for s in range(n_check):
        out, _, stats = solo(s, True, K)
        for k in range(K):
            rc_b, st_b, cov_b = states[s][k][0], states[s][k][1], states[s][k][2]
            if out[k][0] != rc_b:
                identical = False
            if rc_b == 3:
                worst = max(worst, float(np.abs(out[k][1] - st_b).max()))
                if not (np.array_equal(out[k][1], st_b) and np.array_equal(out[k][2], cov_b)):
                    identical = False
        if tuple(stats) != tuple(map_pts[s]):
            identical = False
    out1, ts1, _ = solo(0, False, K)
    solo_ms = 1e3 * float(np.mean([ts1[i] for i in timed]))
    # same-run CPU baseline: session 0's sweeps through the oracle's restatement of fastlio_main after IMU processing (VoxelGrid, iVox kNN on 8
    # threads, esekfom update, map_incremental) -- the engine-level port that tests/test_lru_gpu.py holds the engines against -- with the same priors
    cpu = None
    if args.cpu_scans > 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle  # test infrastructure; used here only as the timed CPU baseline / checker

            threads = min(8, usable_cpus())
            o = oracle.Lio(res=0.5, stencil=19, capacity=1 << 40, threads=threads)
            st, P = plans[0]["s0"].copy(), P0.copy()
            t_o, pts_o, worst = 0.0, 0, 0.0
            for k in range(K):
                raw = plans[0]["scans"][k]["d"].cpu().numpy()
                o.set_state(st)
                o.set_cov(P)
                c0 = time.perf_counter()
                rc_o = o.process_scan(raw, plans[0]["scans"][k]["t"])
                dt_o = time.perf_counter() - c0
                if rc_o != states[0][k][0]:
                    worst = float("inf")
                if rc_o == 3:
                    so = o.get_state()
                    worst = max(worst, float(np.linalg.norm(so[:3] - states[0][k][1][:3])))
                    st, P = next_prior(so, o.get_cov(), plans[0]["step"])
                    if k in timed:
                        t_o += dt_o
                        pts_o += len(raw)
                else:
                    st = st + np.r_[plans[0]["step"], np.zeros(23)]
            cpu = dict(value=round(pts_o / t_o, 1), unit="points/s", cores=threads, host_cpus=usable_cpus(), kind="port",
                       sample=f"session 0's {len(timed)} timed sweeps through the oracle's engine-level restatement (oracle.Lio.process_scan: VoxelGrid, iVox kNN on "
                              f"{threads} threads, esekfom update, map_incremental into its own iVox), same priors rule, {t_o:.1f} s; the reference's own code on "
                              f"streaming sweeps is configs.config3_*.cpu_baseline",
                       ms_per_sweep=round(1e3 * t_o / max(len(timed), 1), 2), gpu_vs_oracle_pose_max_dpos_m=worst)
        except Exception as ex:
            cpu = {"error": repr(ex)[-300:]}
    # SURVEY 8d per sweep, map insert included: B_ds + n_knn B_knn + n_pass B_lin + B_ins
    cand = None
    add_per_sweep = float(np.mean([(map_pts[s][0]) for s in range(n_sess)])) / max(K - 1, 1)
    b_ds = 16.0 * n_raw + 16.0 * nds
    b_lin = 116.0 * nds
    b_ins = 16.0 * nds + 32.0 * add_per_sweep
    out = {
        "metric": "registered + inserted points/sec (B independent SLAM sessions, each with its own map)", "value": round(n_raw / (ms_per_sweep * 1e-3), 1), "unit": "points/s",
        "n_gpus": 1, "steps": K, "warmup": 0, "ms_per_step": round(ms_per_sweep, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 per-point geometry / f64 transforms and reductions", "data": "synthetic",
        "config": {"workload": f"{n_sess} SLAM sessions ({G} groups x {B} slots), each {K} sweeps of 64x{args.n_az} rays (~{n_raw} pts) 1 m apart through the 200 m scene, own map per "
                               f"session grown by map_incremental; lio_batch_sequences_step: downsample + iterated update + map_incremental of all sessions per round in one "
                               f"submission per group, clouds resident in HBM, priors = previous posterior + the known step (no IMU)",
                   "sessions": n_sess, "rounds_timed": len(timed), "n_raw": n_raw, "n_ds_avg": round(nds, 1), "passes_avg": round(npass, 2), "knn_passes_avg": round(nknn, 2),
                   "points_added_per_sweep": round(add_per_sweep, 1), "map_points_end_avg": round(float(np.mean([m_[0] for m_ in map_pts])), 1),
                   "map_voxels_end_avg": round(float(np.mean([m_[1] for m_ in map_pts])), 1), "scan_generation_s": round(t_gen, 1),
                   "return_codes": {str(c): int(rcs_all.count(c)) for c in sorted(set(rcs_all))}},
        "pose_error_vs_truth_m": pe,
        "pose_error_vs_truth": {"after_sweeps": K, "metres_driven": round(step_len * (K - 1), 1), "median_m": float(np.median(pes)), "p90_m": float(np.percentile(pes, 90)), "max_m": pe,
                                "note": "lidar-only odometry over the drive (no IMU in this leg, tight priors): drift, not a registration failure -- the per-session "
                                        "engines give the same bits (parity)"},
        "roofline": {"bound": "hbm", "kernel": "whole sweep incl. map_incremental (SURVEY 8d: B_ds + n_knn B_knn + n_pass B_lin + B_ins; B_knn from the maps these sessions grow is "
                                               "not counted here -- see configs.config3_*.knn_on_this_map for the kernel on such a map)",
                     "achieved": round((b_ds + npass * b_lin + b_ins) / (ms_per_sweep * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round((b_ds + npass * b_lin + b_ins) / (ms_per_sweep * 1e-3) / 8e12, 4), "traffic": None,
                     "terms": {"B_ds": int(b_ds), "B_lin": int(b_lin), "n_pass": round(npass, 2), "B_ins": int(b_ins), "B_knn": "not counted"}},
        "cpu_baseline": cpu,
        "device_us_per_round": dev_us,
        "one_session_at_a_time": {"ms_per_sweep": round(solo_ms, 4), "what": "session 0's sweeps through lio_engine_process_scan_device + flush on its own engine (host-driven loop, "
                                                                             "resident clouds): the single-scan path incl. map_incremental", "speedup_of_the_batch": round(solo_ms / ms_per_sweep, 2)},
        "parity": {"sessions_checked": n_check, "sweeps_each": K, "bit_identical_to_the_per_session_engine": bool(identical), "max_abs_state_difference": worst,
                   "what": "state, covariance and return code of every sweep, map point / voxel counts at the end, against the same scans pushed one by one through "
                           "lio_engine_process_scan_device on an engine with the device loop on"},
    }
    emit(out, "sequences")


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
    tf = {name: dict(dp=[], dr=[], first_bad=None) for name in sides}
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

                        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "stream", "--tf-pinned", "--steps", str(max(m_ref + 5, 30)), "--ref-scans", str(m_ref),
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


def bench_localize(args, torch, local_rank):
    """BASELINE.json config 4 / SURVEY.md 8d: the localisation mode's matcher -- per scan VoxelGrid(leaf 0.2) + NDT-P2D (resolution 1.0, DIRECT7,
    registrations.cpp:105-118) Levenberg-Marquardt alignment from a guess within 0.5 m / 3 deg -- against (a) the prebuilt map RESIDENT in HBM
    (--dense-points, 5e7 = 800 MB of XYZI) and (b) the reference's semantic, a <= 200 000-point local map (localization.cpp:305-308); --steps scans
    each.  The headline value is (a).  Roofline leg: ndt_cost_kernel (correspondences + cost + H + b of one evaluation), HIP events on its stream.
    Baselines in the same run: the reference's own CUDA kernels + LM loop built for gfx950 (oracle/_ref/libref_ndt_cuda.so) on this GPU, and its CPU
    fallback matcher FastVGICP on 4 host threads (oracle/_ref/libref_gicp.so)."""
    from lsd_amd import lio, synth, synth_gpu

    dev = torch.device("cuda", local_rank)
    scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
    scanner = synth_gpu.StaticScanner(scene, dev, n_az=args.n_az, fov_deg=(-24.8, 2.0), max_range=150.0)

    def make_pool(n_scans, spread, seed0):
        """scans at poses uniform over [-spread, spread]^2 (outside the boxes), generated on the device; one LM guess within 0.5 m / 3 deg per step"""
        rng = np.random.default_rng(seed0)
        pl = []
        for k in range(n_scans):
            while True:
                xy = rng.uniform(-spread, spread, 2)
                if not np.any((scene.lo[:, 0] - 1.5 < xy[0]) & (xy[0] < scene.hi[:, 0] + 1.5) & (scene.lo[:, 1] - 1.5 < xy[1]) & (xy[1] < scene.hi[:, 1] + 1.5)):
                    break
            pos = np.array([xy[0], xy[1], 1.8])
            q = synth.quat_from_rotvec([0, 0, rng.uniform(-np.pi, np.pi)])
            d = scanner.scan(pos, q, seed=seed0 + 50 + k)
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = synth.quat_to_R(q), pos
            pl.append(dict(raw=d.cpu().numpy(), d=d, pos=pos, q=q, T=T))
        gs = []
        for i in range(args.steps):
            sc = pl[i % len(pl)]
            gp, gq = synth.perturb_pose(sc["pos"], sc["q"], seed=seed0 + 1000 + i, max_t=0.5, max_deg=3.0)
            G = np.eye(4)
            G[:3, :3], G[:3, 3] = synth.quat_to_R(gq), gp
            gs.append(G)
        return pl, gs

    # the resident 5e7-point map is matched from poses all over the scene; the 200 k-point local map (24 key frames along a line through the middle)
    # from poses inside it -- the reference's localisation never leaves its local map
    pools = {"resident": make_pool(args.scan_pool, args.spread, args.seed + 7), "local_200k": make_pool(8, 4.0, args.seed + 7)}
    pools["resident_one_spot"] = pools["local_200k"]  # round 3's workload against the resident map, beside the spread pool
    pool, guesses = pools["local_200k"]
    n_raw = int(np.mean([len(s["raw"]) for s in pool]))
    leaf = 0.2
    s = lio.Scan(max_raw=1 << 18, max_ds=200000)
    torch.cuda.synchronize()
    g0 = time.perf_counter()
    dense = synth_gpu.sample_surface(scene, args.dense_points, dev, seed=2, sigma=0.01)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - g0
    # the reference's semantic (localization.cpp:303-373): the local map = the clouds of the key frames within 30 m of the pose, nearest first,
    # thinned by key_frame_distance, concatenated until >= 200 000 points, VoxelGrid(resolution) -- assembled on the device by lio_localmap_*
    # from 24 key frames (scans taken every 2 m along a line through the scene's middle, downsampled to 0.2 m, in the map frame)
    lm = lio.LocalMap(max_total_points=4_000_000, max_local_points=200_000, max_keyframe_points=200_000, device=local_rank)
    for kf in range(24):
        kpos = np.array([-23.0 + 2.0 * kf, 0.7 * np.sin(0.4 * kf), 1.8])
        kq = synth.quat_from_rotvec([0, 0, 0.05 * kf])
        kraw = scanner.scan(kpos, kq, seed=args.seed + 900 + kf).cpu().numpy()
        s.upload(kraw)
        s.voxel_downsample(leaf)
        kds = s.get_ds()
        kw = kds.copy()
        kw[:, :3] = (kds[:, :3].astype(np.float64) @ synth.quat_to_R(kq).T + kpos).astype(np.float32)
        lm.add_keyframe(kw, kpos)
    n_local = lio.Ndt(resolution=1.0, search_method=7, max_points=400_000, max_voxels=200_000, max_source_points=200000, device=local_rank)
    code, nk_used, n_local_pts = lm.update(n_local, [0.0, 0.0, 1.8], leaf=leaf)
    if code != 1:
        raise RuntimeError(f"local map assembly returned {code}")
    near = torch.from_numpy(lm.download()).to(dev)
    n_local.close()
    cases = {}
    ref_inputs = {}
    scans_b = []  # the scan buffer sets of the batched leg (made on first use)
    for name, cloud in (("resident", dense), ("resident_one_spot", dense), ("local_200k", near)):
        pool, guesses = pools[name]
        npts = int(cloud.shape[0])
        n = lio.Ndt(resolution=1.0, search_method=7, max_points=npts, max_voxels=max(npts // 4, 200_000), max_source_points=200000, device=local_rank)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n.set_target_device(cloud.data_ptr(), npts)
        nvox = n.num_voxels
        t_build = time.perf_counter() - t0
        for w in range(min(8, args.steps)):  # warm
            sc = pool[w % len(pool)]
            s.set_device(sc["d"].data_ptr(), len(sc["raw"]))
            s.voxel_downsample(leaf)
            n.align(s, guesses[w])
        errs, angs, its, nds, conv = [], [], [], [], 0
        not_conv = []  # (job, LM iterations, |pose - truth|) of alignments that ended at max_iterations
        poses_single = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            sc = pool[i % len(pool)]
            s.set_device(sc["d"].data_ptr(), len(sc["raw"]))
            nds.append(s.voxel_downsample(leaf))
            Ta, cv, it = n.align(s, guesses[i])
            poses_single.append(Ta)
            its.append(it + 1)
            conv += bool(cv)
            if not cv and len(not_conv) < 8 and (i % len(pool), ) not in [(q[0] % len(pool), ) for q in not_conv]:
                not_conv.append((i, it + 1, float(np.linalg.norm(Ta[:3, 3] - sc["T"][:3, 3]))))
            errs.append(float(np.linalg.norm(Ta[:3, 3] - sc["T"][:3, 3])))
            angs.append(float(np.arccos(np.clip((np.trace(Ta[:3, :3].T @ sc["T"][:3, :3]) - 1) / 2, -1, 1))))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # throughput form of the same workload (the scans are independent: each has its own guess): VoxelGrid of 32 scans with one set of
        # launches (lio_scan_voxel_downsample_batch) + their alignments in one lio_ndt_align_batch call, the LM loop on the device
        batched = None
        try:
            if not scans_b:
                scans_b.extend(lio.Scan(max_raw=1 << 18, max_ds=200000, device=local_rank) for _ in range(64))
            sb = scans_b

            def run_batched(NB, split=None):
                out = []
                for base in range(0, args.steps, NB):
                    idx = list(range(base, min(base + NB, args.steps)))
                    c0 = time.perf_counter()
                    for j, i in enumerate(idx):
                        sb[j].set_device(pool[i % len(pool)]["d"].data_ptr(), len(pool[i % len(pool)]["raw"]))
                    lio.Scan.voxel_downsample_batch(sb[:len(idx)], leaf)
                    c1 = time.perf_counter()
                    out += n.align_batch(sb[:len(idx)], [guesses[i] for i in idx])
                    if split is not None:
                        split[0] += c1 - c0
                        split[1] += time.perf_counter() - c1
                return out

            batched = {"what": "the same scans and guesses, NB at a time: lio_scan_voxel_downsample_batch + lio_ndt_align_batch (independent scans, as in the "
                               "metric config; a live localisation loop is sequential and takes the per-scan figure)"}
            for NB in (32, 64):
                run_batched(NB)  # warm (slot buffers of the matcher)
                torch.cuda.synchronize()
                split = [0.0, 0.0]
                tb0 = time.perf_counter()
                res_b = run_batched(NB, split)
                torch.cuda.synchronize()
                dtb = time.perf_counter() - tb0
                dmax = max(float(np.abs(rb[0] - ps).max()) for rb, ps in zip(res_b, poses_single))
                batched[f"{NB}_scans_per_call"] = {"ms_per_scan": round(1e3 * dtb / args.steps, 4), "points_per_s": round(n_raw * args.steps / dtb, 1),
                                                   "voxelgrid_ms_per_scan": round(1e3 * split[0] / args.steps, 4), "align_ms_per_scan": round(1e3 * split[1] / args.steps, 4),
                                                   "converged": int(sum(int(rb[1]) for rb in res_b)), "max_abs_difference_from_the_single_scan_results": dmax,
                                                   "evaluations": int(sum(int(rb[3]) for rb in res_b)), "align_seconds": split[1]}
        except Exception as ex:
            batched = {"error": repr(ex)[-300:]}
        # roofline leg: the same alignments once more with HIP events around every ndt_cost_kernel launch
        n.enable_kernel_timing(True)
        n.kernel_times(reset=True)
        for i in range(min(args.steps, 64)):
            sc = pool[i % len(pool)]
            s.set_device(sc["d"].data_ptr(), len(sc["raw"]))
            s.voxel_downsample(leaf)
            n.align(s, guesses[i])
        kt = n.kernel_times(reset=True)
        n.enable_kernel_timing(False)
        L = max(int(kt["launches"]), 1)
        # B_corr = N_ds' (16 + 7 x 16) per correspondence update, B_der = N_pairs (8 + 16 + 52) per evaluation
        b_alg = (kt["source_points"] / L) * 16.0 + (kt["update_launches"] / L) * (kt["source_points"] / L) * 7 * 16.0 + (kt["pairs"] / L) * 76.0
        us = kt["cost_us"] / L
        ach = b_alg / (us * 1e-6) / 1e9 if us > 0 else 0.0
        if batched and "error" not in batched:
            # the batched cost kernel against the same per-evaluation bytes (SURVEY 8d's B_corr + B_der): evaluations x bytes over the align part
            for key in ("32_scans_per_call", "64_scans_per_call"):
                bj = batched[key]
                ach_b = bj.pop("evaluations") * b_alg / max(bj.pop("align_seconds"), 1e-9) / 1e9
                bj["roofline"] = {"bound": "hbm", "kernel": "ndt_cost_batch<DIRECT7> + ndt_lm_step_batch (whole align part, host checks included)", "achieved": round(ach_b, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_b / HBM_PEAK_GBS, 5), "traffic": None}
        cases[name] = {"target_points": npts, "target_voxels": nvox, "target_build_ms": round(1e3 * t_build, 2), "ms_per_scan": round(1e3 * dt / args.steps, 4),
                       "points_per_s": round(n_raw * args.steps / dt, 1), "n_ds_avg": round(float(np.mean(nds)), 1), "lm_iterations_avg": round(float(np.mean(its)), 2),
                       "converged": conv, "batched": batched, "pos_err_m_median": float(np.median(errs)), "pos_err_m_max": float(np.max(errs)), "rot_err_rad_median": float(np.median(angs)),
                       "roofline": {"bound": "hbm", "kernel": "ndt_cost_kernel<DIRECT7> (1 lane per source point: 7 voxel probes + P2D cost [+ H, b], f64 block reduce)",
                                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                                    "algorithmic_bytes_per_launch": int(b_alg), "avg_launch_us": round(us, 2), "launches": L,
                                    "evaluations_per_alignment": round(L / min(args.steps, 64), 2), "pairs_per_launch": round(kt["pairs"] / L, 1)}}
        if name == "local_200k":
            ref_inputs["target"] = cloud.cpu().numpy()
        if not_conv and args.ref_scans > 0:
            # whose failures are the alignments that end at max_iterations -- the reference's own NDT_CUDA (its kernels compiled for
            # gfx950, oracle/_ref/libref_ndt_cuda.so) on the SAME target cloud, the same downsampled scans and the same guesses
            try:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle as orc_nc
                import ref_ndt_cuda as refn_nc

                if refn_nc.available():
                    reg_nc = refn_nc.NdtCudaRegistration(1.0, 7)
                    reg_nc.set_target(cloud.cpu().numpy())
                    rows_nc = []
                    for (i_nc, it_nc, e_nc) in not_conv:
                        sc_nc = pool[i_nc % len(pool)]
                        reg_nc.set_source(orc_nc.voxel_downsample(sc_nc["raw"], leaf))
                        Tr_nc, cv_nc, itr_nc = reg_nc.align(guesses[i_nc])
                        rows_nc.append({"job": int(i_nc), "ours": {"lm_iterations": int(it_nc), "pos_err_m": round(e_nc, 4)},
                                        "reference": {"converged": bool(cv_nc), "lm_iterations": int(itr_nc + 1),
                                                      "pos_err_m": round(float(np.linalg.norm(Tr_nc[:3, 3] - sc_nc["T"][:3, 3])), 4)}})
                    reg_nc.close()
                    cases[name]["not_converged"] = {"alignments": rows_nc, "reference_converged": int(sum(r["reference"]["converged"] for r in rows_nc)),
                                                    "checked": len(rows_nc),
                                                    "what": "alignments of this case that ended at max_iterations (distinct scans, at most 8), and the reference's own "
                                                            "fast_gicp::NDTCuda on the same target cloud, downsampled scan and guess"}
            except Exception as ex:
                cases[name]["not_converged"] = {"error": repr(ex)[-300:]}
        n.close()
    del dense
    # ---- the map-merge shape (overlap_merge.hpp:46-48,158-179): 64 new key frames x <= 3 candidate frames, every pair an independent alignment of the
    # new frame (source) against the candidate (target) -- one lio_ndt_align_batch call against the per-alignment loop ------------------------------
    merge = None
    try:
        n_t = 3
        tgts = []
        for k in range(n_t):  # three candidate frames: key-frame clouds (downsampled scans in the map frame) from the local map's neighbourhood
            kpos = np.array([-6.0 + 6.0 * k, 1.0 - k, 1.8])
            kq = synth.quat_from_rotvec([0, 0, 0.3 * k])
            kraw = scanner.scan(kpos, kq, seed=args.seed + 950 + k).cpu().numpy()
            s.upload(kraw)
            s.voxel_downsample(leaf)
            kds = s.get_ds()
            kw = kds.copy()
            kw[:, :3] = (kds[:, :3].astype(np.float64) @ synth.quat_to_R(kq).T + kpos).astype(np.float32)
            t = lio.Ndt(resolution=1.0, search_method=7, max_points=len(kw) + 16, max_voxels=200_000, max_source_points=200000, device=local_rank)
            t.set_target(kw)
            tgts.append(t)
        srcs = []
        for w in range(len(pool)):
            sc = lio.Scan(max_raw=1 << 18, max_ds=200000)
            sc.set_device(pool[w]["d"].data_ptr(), len(pool[w]["raw"]))
            sc.voxel_downsample(leaf)
            srcs.append(sc)
        jobs_s, jobs_g, jobs_t = [], [], []
        for kf in range(64):
            for c in range(n_t):
                jobs_s.append(srcs[kf % len(srcs)])
                gi = (kf % len(pool)) + len(pool) * (((kf // len(pool)) * n_t + c) % max(len(guesses) // len(pool), 1))  # a guess made for this source scan
                jobs_g.append(guesses[gi % len(guesses)])
                jobs_t.append(tgts[c])
        prep = tgts[0].prepare_batch(jobs_s, jobs_g, jobs_t)
        tgts[0].run_batch(prep)  # warm (allocates the slot buffers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rcb = tgts[0].run_batch(prep)
        t_batch = time.perf_counter() - t0
        conv_b = sum(int(a.converged) for a in prep[0])
        t0 = time.perf_counter()
        conv_s, dmax = 0, 0.0
        for i in range(len(jobs_s)):
            Ta, cv, it = jobs_t[i].align(jobs_s[i], jobs_g[i])
            conv_s += int(cv)
            dmax = max(dmax, float(np.abs(Ta - np.array(prep[0][i].out).reshape(4, 4)).max()))
        t_loop = time.perf_counter() - t0
        merge = {"alignments": len(jobs_s), "targets": n_t, "batched_call_ms": round(1e3 * t_batch, 3), "per_alignment_loop_ms": round(1e3 * t_loop, 3),
                 "ms_per_alignment_batched": round(1e3 * t_batch / len(jobs_s), 4), "ms_per_alignment_loop": round(1e3 * t_loop / len(jobs_s), 4),
                 "converged_batched": conv_b, "converged_loop": conv_s, "max_abs_difference_of_the_results": dmax, "rc": int(rcb),
                 "what": "overlap_merge.hpp:158-179's workload: 64 key frames x 3 candidate frames = 192 independent NDT alignments (source already downsampled); "
                         "lio_ndt_align_batch (64 slots per launch, LM loop on the device) vs 192 lio_ndt_align calls"}
        for t in tgts:
            t.close()
    except Exception as ex:
        merge = {"error": repr(ex)[-300:]}
    # ---- baselines on the reference's semantic (local map), same scans, same guesses -------------------------------------------------------
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    base = {}
    m_ref = min(args.steps, args.ref_scans if args.ref_scans > 0 else 0, 64)
    ds_host = [orc.voxel_downsample(pool[w]["raw"], leaf) for w in range(len(pool))] if m_ref or args.vgicp_scans else []
    try:
        import ref_ndt_cuda as refn

        if m_ref and refn.available():
            reg = refn.NdtCudaRegistration(1.0, 7)
            reg.set_target(ref_inputs["target"])
            reg.set_source(ds_host[0])
            reg.align(guesses[0])
            t_ref, it_ref, e_ref, d_ref_t, d_ref_r, own_t, own_r = 0.0, [], [], [], [], [], []

            def rot_angle(A, B):  # from the skew part: arccos of the trace loses everything below 4e-4 rad on the reference's f32 matrices
                Rm = A[:3, :3] @ B[:3, :3].T
                return float(np.arcsin(min(1.0, 0.5 * np.linalg.norm([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]]))))

            R_REP = 8  # the reference against ITSELF, every scan: the same alignment R_REP times, each on a rebuilt voxel map (its voxel means are f32 atomics)
            env_t, env_r, near_t, near_r, inside = [], [], [], [], 0
            for i in range(m_ref):
                c0 = time.perf_counter()
                reg.set_source(ds_host[i % len(pool)])
                Tr, cv, it = reg.align(guesses[i])
                t_ref += time.perf_counter() - c0
                it_ref.append(it + 1)
                e_ref.append(float(np.linalg.norm(Tr[:3, 3] - pool[i % len(pool)]["T"][:3, 3])))
                if i < len(poses_single):  # (poses_single: the last case of the loop above = the same local-200k target, scans and guesses)
                    d_ref_t.append(float(np.linalg.norm(Tr[:3, 3] - poses_single[i][:3, 3])))
                    d_ref_r.append(rot_angle(Tr, poses_single[i]))
                    runs = [Tr]
                    for _ in range(R_REP - 1):
                        reg.set_target(ref_inputs["target"])
                        reg.set_source(ds_host[i % len(pool)])
                        runs.append(reg.align(guesses[i])[0])
                    et = max(float(np.linalg.norm(a[:3, 3] - b[:3, 3])) for a in runs for b in runs)
                    er = max(rot_angle(a, b) for a in runs for b in runs)
                    nt = min(float(np.linalg.norm(a[:3, 3] - poses_single[i][:3, 3])) for a in runs)
                    nr = min(rot_angle(a, poses_single[i]) for a in runs)
                    env_t.append(et); env_r.append(er); near_t.append(nt); near_r.append(nr)
                    own_t.append(et); own_r.append(er)
                    inside += int((nt <= max(et, 1e-4)) and (nr <= max(er, 1e-5)))
            reg.close()
            if d_ref_t:
                base["gpu_vs_reference_pose"] = {"scans": len(d_ref_t), "max_dpos_m": float(np.max(d_ref_t)), "max_drot_rad": float(np.max(d_ref_r)),
                                                 "median_dpos_m": float(np.median(d_ref_t)),
                                                 "scans_beyond_1e_4_m_or_1e_5_rad": int(np.count_nonzero((np.array(d_ref_t) > 1e-4) | (np.array(d_ref_r) > 1e-5))),
                                                 "reference_run_to_run": {"alignments": len(own_t) * R_REP, "max_dpos_m": float(np.max(own_t)) if own_t else None,
                                                                          "max_drot_rad": float(np.max(own_r)) if own_r else None,
                                                                          "median_dpos_m": float(np.median(own_t)) if own_t else None},
                                                 "per_scan_envelope": {
                                                     "what": "every scan aligned %d times by the reference, each on a rebuilt voxel map: envelope = the largest distance between two of its "
                                                             "own results for that scan; HIP is INSIDE when its distance to the nearest of them is no larger (floor: the north_star "
                                                             "tolerance)" % R_REP,
                                                     "scans": len(env_t), "hip_inside_the_references_own_envelope": inside,
                                                     "hip_to_nearest_reference_run_m": {"median": float(np.median(near_t)), "max": float(np.max(near_t))},
                                                     "reference_envelope_m": {"median": float(np.median(env_t)), "max": float(np.max(env_t))},
                                                     "hip_to_nearest_reference_run_rad": {"median": float(np.median(near_r)), "max": float(np.max(near_r))},
                                                     "reference_envelope_rad": {"median": float(np.median(env_r)), "max": float(np.max(env_r))},
                                                     "inside_in_translation": int(np.count_nonzero(np.array(near_t) <= np.maximum(np.array(env_t), 1e-4))),
                                                     "inside_in_rotation": int(np.count_nonzero(np.array(near_r) <= np.maximum(np.array(env_r), 1e-5))),
                                                     "hip_over_envelope_max_ratio": float(np.max(np.array(near_t) / np.maximum(np.array(env_t), 1e-4)))},
                                                 "note": "HIP NDT against the reference's own fast_gicp::NDTCuda (compiled for gfx950) on the same local-200k target, scans and "
                                                         "guesses.  Both stop when the LM step falls below LsqRegistration's termination thresholds, i.e. anywhere within that "
                                                         "distance of the optimum, and the reference accumulates H / b / cost with f32 atomics in thread order: "
                                                         "reference_run_to_run is the SAME alignment repeated by the reference on a rebuilt voxel map.  The north_star tolerance "
                                                         "(1e-4 m / 1e-5 rad) is FastLIO's pose; tests/test_ndt_vs_ref_cuda.py holds the matcher to max(1e-4 m, 3 x that spread)"}
            base["reference_ndt_cuda_on_this_gpu"] = {"ms_per_scan": round(1e3 * t_ref / m_ref, 3), "scans": m_ref, "lm_iterations_avg": round(float(np.mean(it_ref)), 2),
                                                      "pos_err_m_median": float(np.median(e_ref)),
                                                      "what": "fast_gicp::NDTCuda<PointXYZI, PointXYZI> (registrations.cpp:105-118) with the reference's own CUDA / Thrust kernels compiled "
                                                              "for gfx950 (oracle/_ref/libref_ndt_cuda.so), setInputSource + align on the local-200k target; the VoxelGrid before it "
                                                              "(CPU in the reference) is NOT in this time"}
    except Exception as ex:
        base["reference_ndt_cuda_on_this_gpu"] = {"error": repr(ex)[-300:]}
    cpu = None
    try:
        import ref_gicp

        if args.vgicp_scans > 0 and ref_gicp.available():
            threads = min(4, usable_cpus())
            vg = ref_gicp.RefVgicp(k=20, resolution=1.0, search_method=1, transformation_epsilon=0.1, rotation_epsilon=0.1, max_iterations=64, num_threads=threads)
            c0 = time.perf_counter()
            vg.set_target(ref_inputs["target"])
            t_tgt = time.perf_counter() - c0
            t_v, e_v = 0.0, []
            for i in range(args.vgicp_scans):
                c0 = time.perf_counter()
                ds = orc.voxel_downsample(pool[i % len(pool)]["raw"], leaf)
                vg.set_source(ds)
                out = vg.align(guesses[i])
                t_v += time.perf_counter() - c0
                Tv = out[0] if isinstance(out, tuple) else out["T"]
                e_v.append(float(np.linalg.norm(np.asarray(Tv)[:3, 3] - pool[i % len(pool)]["T"][:3, 3])))
            vg.close()
            cpu = dict(value=round(n_raw * args.vgicp_scans / t_v, 1), unit="points/s", cores=threads, host_cpus=usable_cpus(), kind="reference",
                       sample=f"{args.vgicp_scans} of the same alignments through the reference's matcher for machines without CUDA -- fast_gicp::FastVGICP as "
                              f"select_registration_method(\"FAST_VGICP\") configures it (registrations.cpp:56-66; oracle/_ref/libref_gicp.so, {threads} OpenMP threads, an exact "
                              f"grid k-NN in place of PCL's kd-tree) -- VoxelGrid(0.2) [the oracle's restatement] + setInputSource (20-NN covariances) + align on the local-200k "
                              f"target, {t_v:.1f} s (+ {t_tgt:.1f} s setInputTarget once)",
                       ms_per_scan=round(1e3 * t_v / args.vgicp_scans, 2), pos_err_m_median=float(np.median(e_v)), other=base)
    except Exception as ex:
        cpu = {"error": repr(ex)[-300:], "other": base}
    if cpu is None:
        cpu = {"other": base} if base else None
    if cpu is not None and base.get("gpu_vs_reference_pose"):
        cpu["gpu_vs_reference_pose"] = base["gpu_vs_reference_pose"]  # (beside the baseline's own figures: what the compact line reports per leg)
    head = cases["resident"]
    out = {"metric": "registered points/sec (localisation: VoxelGrid 0.2 + NDT-P2D LM alignment vs a prebuilt map resident in HBM)", "value": head["points_per_s"],
           "unit": "points/s", "n_gpus": 1, "steps": args.steps, "warmup": 8, "ms_per_step": head["ms_per_scan"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32 per-point arithmetic / f64 reductions and LM", "data": "synthetic",
           "config": {"workload": "BASELINE config 4: %d scans of 64x%d rays (~%d pts), leaf-0.2 VoxelGrid + NDT-P2D (res 1.0, DIRECT7) LM alignment from a guess within "
                                  "0.5 m / 3 deg vs a %d-pt map resident in HBM (map generated on the GPU in %.1f s)" % (args.steps, args.n_az, n_raw, args.dense_points, t_gen),
                      "n_raw": n_raw, "leaf": leaf, "scan_pool": len(pools["resident"][0]), "spread_m": args.spread,
                      "resident_map": {k: v for k, v in head.items() if k != "roofline"},
                      "resident_map_one_spot_pool": cases["resident_one_spot"],
                      "local_200k_map": cases["local_200k"], "local_map_key_frames_used": nk_used, "merge_candidates_batched": merge},
           "roofline": head["roofline"], "cpu_baseline": cpu, "pose_error_vs_truth_m": head["pos_err_m_max"]}
    emit(out, "localize")


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


if __name__ == "__main__":
    main()
