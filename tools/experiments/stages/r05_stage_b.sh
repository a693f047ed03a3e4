#!/bin/bash
# round 5, second GPU call: per-instruction issue rates, parity of the patched kNN + the new oracle leg of the sequence test
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 300 python tools/valu_peak/run.py > $O/valu_peak.json 2>&1
python - <<PY
import json
j = json.load(open("$O/valu_peak.json"))
for k, v in j["rates"].items():
    print(f"{k:30s}", {w: round(r["cycles_per_wave_inst_per_simd_at_reported_clock"], 2) for w, r in v.items()})
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_sequence_batch_gpu.py tests/test_lru_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
