#!/bin/bash
# round 4, stage x: linearize with its loads up front, the single-alignment NDT kernels with the records of all seven offsets in flight: parity, then
# the headline's short form and the localisation leg's short form
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_ndt_gpu.py tests/test_ndt_vs_ref_cuda.py tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_golden_gpu.py tests/test_fastlio_golden.py tests/test_degenerate.py tests/test_localization_boundary.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 > $O/bench.json 2> $O/bench.err
timeout 240 python bench.py --config localize --steps 60 --ref-scans 0 --vgicp-scans 0 --scan-pool 16 > $O/localize.json 2> $O/localize.err
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("ms/scan", d["ms_per_step"], "single-stream latency", d["config"].get("single_stream_latency_ms_per_scan"), r["other_kernels_us"])
d = json.load(open("$O/localize.json")); c = d["config"]
for k in ("resident_map", "resident_map_one_spot_pool", "local_200k_map"):
    if k in c: print(k, c[k].get("ms_per_scan"), (c[k].get("roofline") or {}).get("avg_launch_us"), c[k].get("lm_iterations_avg"), (c[k].get("batched") or {}).get("64_scans_per_call", {}).get("ms_per_scan"))
PY
