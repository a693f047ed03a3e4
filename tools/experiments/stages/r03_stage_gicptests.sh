#!/bin/bash
# the GICP family after a kernel change: its parity tests, the flows built on it, then the timing table
python -m pytest tests/test_gicp_gpu.py tests/test_overlap_merge_gpu.py tests/test_outer_boundary.py -m gpu -q 2>&1 | tail -15
python tools/experiments/gicp_time.py > gpurun_out/gicp_timing.txt 2>&1
cat gpurun_out/gicp_timing.txt
