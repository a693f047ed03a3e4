#!/bin/bash
python -m pytest tests/test_ndt_gpu.py -m gpu -x -q -k batched_voxel 2>&1 | tail -15
