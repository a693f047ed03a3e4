#!/bin/bash
python tools/experiments/fitness_time.py 2>&1 | tail -8
python -m pytest tests/test_ndt_gpu.py tests/test_overlap_merge_gpu.py tests/test_localization_boundary.py -m gpu -q 2>&1 | tail -8
