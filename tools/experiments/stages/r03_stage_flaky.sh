#!/bin/bash
for i in 1 2 3; do
  python -m pytest tests/test_localization_boundary.py tests/test_dist.py tests/test_overlap_merge_gpu.py tests/test_ndt_vs_ref_cuda.py -m gpu -q 2>&1 | tail -2
done
