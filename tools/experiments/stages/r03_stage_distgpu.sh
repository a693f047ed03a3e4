#!/bin/bash
python -m pytest tests/test_dist.py -m gpu -x -q 2>&1 | tail -30
