#!/bin/bash
python tools/experiments/drift_vs_reference.py 1200 20 2>&1 | grep -v amdgpu.ids | tail -30
