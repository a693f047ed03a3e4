#!/bin/bash
# round 4, stage t: knobs re-measured on the pool of 128 scans spread over the map (they were last swept on round 3's eight resident scans):
# kNN waves per SIMD 6 (default) / 7 / 8 (variant libraries through LIO_HIP_LIB), rounds in flight 4 / 6 / 8
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04t
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/tools/experiments/variants
COMMON="--steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0"
run() {  # name, env, extra args
    local name=$1; shift
    local lib=$1; shift
    if [ -n "$lib" ]; then export LIO_HIP_LIB=$lib; else unset LIO_HIP_LIB; fi
    timeout 240 python bench.py $COMMON "$@" > $O/$name.json 2> $O/$name.err
    python - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); r = d["roofline"]
    print("$name", "ms/scan", d["ms_per_step"], "knn us/scan-search", r["other_kernels_us"].get("knn_per_scan_and_search"), "knn launch us", r["avg_launch_us"], "parity", d["batch_vs_oracle_pose"]["parity_ok"])
except Exception as ex:
    print("$name", "FAILED", repr(ex)[:200])
PY
}
run w6_g4 ""
run w7_g4 $V/liblio_hip_knnw7.so
run w8_g4 $V/liblio_hip_knnw8.so
run w6_g6 "" --groups 6
run w6_g8 "" --groups 8
