"""Does RCCL accept two ranks on ONE device?  (The boxes of this pool have one GPU; if it does, the world-2 RCCL path can run for real.)
Launched under torch.distributed.run with two processes; gloo carries the unique id; each rank then gathers a 32-double record."""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "lidar-slam-detection_amd", "python"))
from lsd_amd import lio
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
uid = [lio.Comm.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
try:
    comm = lio.Comm(rank=rank, world=world, device=0, uid=uid[0])
except Exception as ex:
    print(f"rank {rank}: lio_comm_init failed: {ex}", flush=True)
    sys.exit(0)
loc = torch.full((32,), float(rank + 1), dtype=torch.float64, device="cuda:0")
gat = torch.zeros((world * 32,), dtype=torch.float64, device="cuda:0")
summ = torch.zeros((32,), dtype=torch.float64, device="cuda:0")
comm.allgather(loc.data_ptr(), gat.data_ptr(), summ.data_ptr(), None)
torch.cuda.synchronize()
print(f"rank {rank}: gathered {gat.view(world, 32)[:, 0].tolist()} sum {summ[0].item()}", flush=True)
