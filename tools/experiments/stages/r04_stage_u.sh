#!/bin/bash
# round 4, stage u: variant libraries of the kNN kernel against the default build on the headline workload (short form), one line each
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04u
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/tools/experiments/variants
COMMON="--steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0"
run() {
    local name=$1; shift
    local lib=$1; shift
    if [ -n "$lib" ]; then export LIO_HIP_LIB=$lib; else unset LIO_HIP_LIB; fi
    timeout 240 python bench.py $COMMON "$@" > $O/$name.json 2> $O/$name.err
    python - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); r = d["roofline"]
    print("$name", "ms/scan", d["ms_per_step"], "knn us/scan-search", r["other_kernels_us"].get("knn_per_scan_and_search"), "knn launch us", r["avg_launch_us"], "parity", (d.get("batch_vs_oracle_pose") or {}).get("parity_ok"))
except Exception as ex:
    print("$name", "FAILED", repr(ex)[:200])
PY
}
run base ""
for f in $V/*.so; do run $(basename $f .so) $f; done
run base_again ""
