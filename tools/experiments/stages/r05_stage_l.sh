#!/bin/bash
# round 5, call L: hardware queue count at 128 x 4
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05l; mkdir -p $O; cd $R; export TMPDIR=/tmp
for q in 8 4 12 16 24; do
  GPU_MAX_HW_QUEUES=$q timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print("GPU_MAX_HW_QUEUES=$q: ms/scan", d["ms_per_step"], "dev/scan", r["other_kernels_us"]["device_time_per_scan_one_round_in_flight"])
PY
done
