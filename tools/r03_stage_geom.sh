#!/bin/bash
for g in "96 3" "128 2" "128 3" "96 4" "64 4"; do
  set -- $g
  python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-scans 0 --ref-scans 0 --slots $1 --groups $2 --min-seconds 2 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots $1 groups $2', b['ms_per_step'], b['roofline']['other_kernels_us']['knn_per_scan_and_search'])"
done
