#!/bin/bash
# headline throughput against (scans per launch, rounds in flight): bash tools/experiments/sweep_geometry.sh  (on the GPU box)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/sweep
for G in "128 4" "128 6" "128 8" "256 2" "256 3" "256 4" "192 4" "96 6"; do
set -- $G
timeout 300 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 3 --cpu-scans 0 --ref-scans 0 --upload-scans 0 --slots $1 --groups $2 > gpurun_out/sweep/b_$1_$2.json 2> gpurun_out/sweep/b_$1_$2.err
python - <<PY
import json
try:
    d = json.load(open("bench_full.json")); r = d["roofline"]
    print("slots $1 groups $2: ms/scan", d["ms_per_step"], "knn per scan-search", r["other_kernels_us"]["knn_per_scan_and_search"])
except Exception as ex:
    print("slots $1 groups $2: failed", ex)
PY
done
