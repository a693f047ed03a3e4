#!/bin/bash
for g in 96 144 192 240 288 384; do
  LIO_KNN_GRID=$g python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-scans 0 --ref-scans 0 --min-seconds 2 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LIO_KNN_GRID $g', b['ms_per_step'], b['roofline']['other_kernels_us']['knn_per_scan_and_search'])"
done
