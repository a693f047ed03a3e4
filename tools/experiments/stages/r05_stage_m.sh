#!/bin/bash
# round 5, call M: stream priorities of the rounds in flight
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R; export TMPDIR=/tmp
for m in 0 1 2 0 1 2; do
  LIO_BATCH_PRIO=$m timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print("LIO_BATCH_PRIO=$m: ms/scan", d["ms_per_step"])
PY
done
