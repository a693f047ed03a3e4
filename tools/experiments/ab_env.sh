#!/bin/bash
# quick A/B of an environment knob of the library: bash tools/experiments/ab_env.sh VAR v1 v2 ...   ("-" = unset)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/abe
VAR=$1; shift
for rep in 1 2; do
for V in "$@"; do
if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
timeout 300 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 3 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > gpurun_out/abe/b.json 2> gpurun_out/abe/b.err
python - <<PY
import json
try:
    d = json.load(open("bench_full.json")); r = d["roofline"]
    print("$VAR=$V: ms/scan", d["ms_per_step"], "knn per scan-search", r["other_kernels_us"]["knn_per_scan_and_search"], "one-round device us/scan", r["other_kernels_us"]["device_time_per_scan_one_round_in_flight"])
except Exception as ex:
    print("$VAR=$V: failed", ex)
PY
done; done
