#!/bin/bash
# round 5, call I: kNN sweep with the first listed voxel alone before the bound (LIO_KNN_HOME_FIRST=1): parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R; export TMPDIR=/tmp
LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_home1.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_batch_gpu.py -m gpu -x -q 2>&1 | tail -2
for v in "" home1 "" home1; do
  if [ -n "$v" ]; then export LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_$v.so; else unset LIO_HIP_LIB; fi
  timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print("variant '$v': ms/scan", d["ms_per_step"], "knn/search", r["other_kernels_us"]["knn_per_scan_and_search"], "touched", r["touched_bytes_per_launch"], "frac_touched", r["frac_touched"])
PY
done
