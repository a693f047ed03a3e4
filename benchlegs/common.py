"""What every leg of bench.py shares: the environment the HIP runtime must find, the repository paths, the roofline constants, the output
function (full record to a file and stderr, the compact line to stdout), the measured VALU peak, the launcher of the ranks."""
import argparse
import json
import os
import sys
import time

import numpy as np

# HIP deals streams to hardware queues round robin; with the default of 4 the rounds in flight of the batched engine share queues with
# idle streams and mostly run back to back (measured: 0.085 -> 0.070 ms per scan with 8).  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # the repository root (bench.py, bench_line.py)
BENCH_PY = os.path.join(ROOT, "bench.py")  # what a leg's child processes run
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lidar-slam-detection_amd", "python"))

import bench_line  # the compact stdout line (bench_line.py, beside this file)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
N_SIMD, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs, peak engine clock
KNN_VALU_PER_WAVE_STATIC = 1630.0  # VALU instructions of one wave (four queries) of knn_batch_kernel at the metric map's trip counts
_STATE = {"full_line": False}  # --full-line (the children of the metric config's secondary legs): the whole record on stdout instead of the compact line




def set_full_line(on):
    """--full-line: the whole record on stdout instead of the compact line (the children of the metric config's secondary legs)"""
    _STATE["full_line"] = bool(on)


def emit(out, tag="metric"):
    """rank 0's output: the whole record to bench_full[_<tag>].json beside this file and to stderr, the compact line (<= bench_line.LIMIT bytes: the
    driver's parser did not take round 4's 30 KB line) as the ONE stdout line"""
    if _STATE["full_line"]:
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
           BENCH_PY] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)
