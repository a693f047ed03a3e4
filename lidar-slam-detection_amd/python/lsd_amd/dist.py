"""Multi-GPU use of the LIO core: one process per GPU, `torch.distributed` ("nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).

Two shapes of parallelism:
  * independent scans / alignments: `shard_range` splits the work list, every rank keeps a replica of the map,
    no collective on the data path (this is what bench.py --gpus N measures);
  * joint registration against sub-maps that live on different GPUs (BASELINE.json config 5, multi-map merge):
    every rank linearises the same scan against ITS sub-map, `NormalEqAllGather` all-gathers the per-rank
    29-double records (J^T J upper triangle, J^T r, sum|r|, N_eff) and sums them in fixed rank order, and every rank
    runs the same 23-DoF update on the same numbers -- bitwise identical states on all ranks, no broadcast needed.
    The payload is 232 bytes per rank per pass: latency-bound, not xGMI-bandwidth-bound.
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous [lo, hi) of n work items for `rank` of `world` (sizes differ by at most one)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class NormalEqAllGather:
    """reduce hook for lio.Engine.set_reduce_hook: all-gather + fixed-order sum, in place"""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        backend = dist.get_backend(group)
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
        self.calls = 0
        self.bytes = 0

    def __call__(self, buf):
        torch = self.torch
        n = len(buf)
        local = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64)).to(self.device)
        flat = torch.empty(self.world * n, dtype=torch.float64, device=self.device)  # rank-major: [r * n + i]
        self.dist.all_gather_into_tensor(flat, local, group=self.group)
        gathered = flat.view(self.world, n)
        acc = gathered[0].clone()
        for r in range(1, self.world):  # fixed rank order: every rank forms the identical sum
            acc += gathered[r]
        buf[:] = acc.cpu().numpy()
        self.calls += 1
        self.bytes += 8 * n * self.world


class RecordsAllGatherHost:
    """gather hook for lio.Batch.set_gather_hook over a torch.distributed group with HOST tensors (gloo): the round's [n x 32] doubles leave the
    device, are all-gathered rank-major and come back -- the transport of the tests where RCCL cannot run (two ranks on one GPU); a real
    multi-GPU job hands lio_batch_create_joint a lio_comm instead (RCCL on the round's stream, no host in the loop)"""

    def __init__(self, group=None):
        import ctypes

        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        self.calls = 0
        self.records = 0  # 32-double records this rank has sent

    def __call__(self, d_local, d_gathered, n, stream):
        torch = self.torch
        if self.hip.hipStreamSynchronize(stream) != 0:
            return -1
        local = torch.empty(n * 32, dtype=torch.float64)
        if self.hip.hipMemcpy(local.data_ptr(), d_local, n * 32 * 8, 2) != 0:  # device -> host
            return -1
        flat = torch.empty(self.world * n * 32, dtype=torch.float64)  # rank-major: [rank][record][32]
        self.dist.all_gather_into_tensor(flat, local, group=self.group)
        if self.hip.hipMemcpy(d_gathered, flat.data_ptr(), self.world * n * 32 * 8, 1) != 0:  # host -> device
            return -1
        self.calls += 1
        self.records += n
        return 0


def sequential_sum(records):
    """what NormalEqAllGather computes, for a list of per-rank records held in one process (tests, 1-GPU emulation)"""
    acc = np.array(records[0], dtype=np.float64, copy=True)
    for r in records[1:]:
        acc += np.asarray(r, dtype=np.float64)
    return acc


def pack_normal_eq(JtJ, Jtr, sum_abs_res, n_eff):
    """the 29-double record of include/lio_hip.h (lio_reduce_fn, first call)"""
    J = np.asarray(JtJ, np.float64).reshape(6, 6)
    buf = np.zeros(29)
    t = 0
    for a in range(6):
        for c in range(a, 6):
            buf[t] = J[a, c]
            t += 1
    buf[21:27] = Jtr
    buf[27] = sum_abs_res
    buf[28] = n_eff
    return buf
