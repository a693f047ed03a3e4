#!/bin/bash
# quick A/B of library variants (tools/build_variant.sh): bash tools/experiments/ab_variants.sh name1 name2 ...   ("base" = the tree's library)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/abv
for rep in 1 2; do
for V in "$@"; do
if [ "$V" = "base" ]; then unset LIO_HIP_LIB; else export LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_$V.so; fi
timeout 300 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 3 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > gpurun_out/abv/b_$V.json 2> gpurun_out/abv/b_$V.err
python - <<PY
import json
try:
    d = json.load(open("bench_full.json")); r = d["roofline"]
    print("$V: ms/scan", d["ms_per_step"], "knn per scan-search", r["other_kernels_us"]["knn_per_scan_and_search"], "one-round device us/scan", r["other_kernels_us"]["device_time_per_scan_one_round_in_flight"])
except Exception as ex:
    print("$V: failed", ex)
PY
done; done
