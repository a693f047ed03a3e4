#!/bin/bash
# round 5, call H: slots x groups of the batched engine after the grid-stride linearize
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R; export TMPDIR=/tmp
for sg in "96 4" "128 4" "128 3" "160 3" "160 4" "192 2" "192 3" "256 2" "256 3" "96 5"; do
  set -- $sg
  timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 --slots $1 --groups $2 > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print("slots $1 groups $2: ms/scan", d["ms_per_step"], "knn/search", r["other_kernels_us"]["knn_per_scan_and_search"], "dev/scan", r["other_kernels_us"]["device_time_per_scan_one_round_in_flight"])
PY
done
