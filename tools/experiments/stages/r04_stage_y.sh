#!/bin/bash
# round 4, stage y: folds with their loads in flight (finalize, step_batch, joint fold, NDT report), undistort's loads hoisted: parity, then the short
# forms of the headline, the localisation leg and the streaming leg
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04y
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ndt_gpu.py tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_frontend_gpu.py tests/test_golden_gpu.py tests/test_fastlio_golden.py tests/test_sequence_batch_gpu.py tests/test_overlap_merge_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 > $O/bench.json 2> $O/bench.err
timeout 240 python bench.py --config localize --steps 60 --ref-scans 0 --vgicp-scans 0 --scan-pool 16 > $O/localize.json 2> $O/localize.err
timeout 240 python bench.py --config stream --steps 300 --lru 100000 --ref-scans 0 > $O/stream.json 2> $O/stream.err
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("ms/scan", d["ms_per_step"], "single-stream latency", d["config"].get("single_stream_latency_ms_per_scan"), r["other_kernels_us"])
d = json.load(open("$O/localize.json")); c = d["config"]
for k in ("resident_map", "resident_map_one_spot_pool", "local_200k_map"):
    if k in c: print(k, c[k].get("ms_per_scan"), (c[k].get("roofline") or {}).get("avg_launch_us"), c[k].get("lm_iterations_avg"), (c[k].get("batched") or {}).get("64_scans_per_call", {}).get("ms_per_scan"))
print("merge candidates", c.get("merge_candidates_batched", {}).get("ms_per_alignment_batched"), c.get("merge_candidates_batched", {}).get("ms_per_alignment_loop"))
d = json.load(open("$O/stream.json")); c = d["config"]
print("stream", d["ms_per_step"], c.get("main_ms_median"), c.get("main_ms_p99"), (d.get("roofline") or {}).get("stage_us_per_scan"))
PY
