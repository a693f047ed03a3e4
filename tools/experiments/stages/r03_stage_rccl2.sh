#!/bin/bash
export NCCL_DEBUG=WARN
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 tools/experiments/rccl_two_ranks_one_gpu.py 2>&1 | grep -v amdgpu.ids | tail -25
echo "rc=$?"
