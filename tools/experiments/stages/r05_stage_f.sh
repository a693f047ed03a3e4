#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R; export TMPDIR=/tmp
for v in "" lin128 lin192 lin384; do
  if [ -n "$v" ]; then export LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_$v.so; else unset LIO_HIP_LIB; fi
  timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print("variant '$v': ms/scan", d["ms_per_step"], "lat", d["config"]["single_stream_latency_ms_per_scan"], r["other_kernels_us"])
PY
done
