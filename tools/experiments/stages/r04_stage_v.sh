#!/bin/bash
# round 4, stage v: the kNN kernel with its loads really in flight together (probe: offsets / home slots; sweep: the four candidate loads): parity of
# the neighbour sets and of the batched engine, then the headline's short form
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04v
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_lru_gpu.py tests/test_sequence_batch_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("ms/scan", d["ms_per_step"], "knn us/scan-search", r["other_kernels_us"].get("knn_per_scan_and_search"), "knn launch us", r["avg_launch_us"], "one round in flight", r["other_kernels_us"].get("device_time_per_scan_one_round_in_flight"))
PY
