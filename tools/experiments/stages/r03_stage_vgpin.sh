#!/bin/bash
python -m pytest tests/test_voxelgrid_vs_ref.py -q 2>&1 | tail -8
