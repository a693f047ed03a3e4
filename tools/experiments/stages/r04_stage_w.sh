#!/bin/bash
# round 4, stage w: a list of GPU tests (arguments, default: the voxel-grid and batch parity tests), then the headline's short form with the chain's time
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04w
mkdir -p $O
cd $R
export TMPDIR=/tmp
TESTS=${@:-tests/test_voxelgrid_vs_ref.py tests/test_voxelgrid_monster_gpu.py tests/test_voxelgrid_crosscheck.py tests/test_golden_gpu.py tests/test_batch_gpu.py tests/test_fullsize_gpu.py}
timeout 800 python -m pytest $TESTS -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("ms/scan", d["ms_per_step"], "single-stream latency", d["config"].get("single_stream_latency_ms_per_scan"), r["other_kernels_us"])
PY
