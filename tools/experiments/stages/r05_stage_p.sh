#!/bin/bash
# round 5, call P: kNN workgroups per slot at 128 scans per launch (LIO_KNN_GRID)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R; export TMPDIR=/tmp
for g in 0 64 96 160 192 256; do
  if [ "$g" = "0" ]; then unset LIO_KNN_GRID; else export LIO_KNN_GRID=$g; fi
  timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print("LIO_KNN_GRID=$g: ms/scan", d["ms_per_step"], "knn/search", r["other_kernels_us"]["knn_per_scan_and_search"], "dev/scan", r["other_kernels_us"]["device_time_per_scan_one_round_in_flight"])
PY
done
