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
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from benchlegs.common import *  # noqa: E402,F401,F403  (sets GPU_MAX_HW_QUEUES before the HIP runtime starts; argparse, os, sys ...)
from benchlegs.common import set_full_line, spawn_ranks  # noqa: E402
from benchlegs.localize import bench_localize  # noqa: E402
from benchlegs.merge import bench_merge, dry_run  # noqa: E402
from benchlegs.metric import bench_metric  # noqa: E402
from benchlegs.parity import ref_parity_leg  # noqa: E402
from benchlegs.rccl import rccl_probe, rccl_probe_child  # noqa: E402,F401
from benchlegs.sequences import bench_sequences  # noqa: E402
from benchlegs.stream import bench_stream, load_bin_dir, stream_tf_pinned  # noqa: E402,F401


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
    set_full_line(args.full_line)

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

    return bench_metric(args, torch, dist, world, rank, local_rank, dev)


if __name__ == "__main__":
    main()
