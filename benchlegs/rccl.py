"""The RCCL probe: the C ABI's communicator brought up in child processes (bench.py --config rcclprobe is one of them)."""
from .common import *  # noqa: F401,F403  (argparse, json, os, sys, time, np, ROOT, BENCH_PY, the roofline constants, emit, usable_cpus ...)


def rccl_probe(dist, rank, world, local_rank, n_records=64, iters=200, timeout_s=75.0):
    """bring up lio_comm (ncclCommInitRank through the C ABI) on all ranks and time lio_allgather_records of [n_records x 32] doubles per rank -- the
    per-round, per-pass collective of the batched joint registration (config 5).  Every rank does it in a CHILD process (`--config rcclprobe`: its
    own HIP context on the rank's GPU, nothing of torch.distributed inside): the process that holds the headline never loads a second RCCL beside
    torch's, and a communicator that crashes or hangs -- this path has never met more than one GPU -- costs the run a minute, not its line.  The
    unique id comes from a child of rank 0 and travels through torch.distributed."""
    import subprocess

    me = BENCH_PY
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
