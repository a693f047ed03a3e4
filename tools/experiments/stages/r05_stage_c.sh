#!/bin/bash
# round 5, GPU call C: more per-instruction rates, kNN per-lane statistic + cheaper-bound variant A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 300 python tools/valu_peak/run.py > $O/valu_peak.json 2>&1
python - <<PY
import json
j = json.load(open("$O/valu_peak.json"))
for k, v in j["rates"].items():
    print(f"{k:30s}", {w: round(r["cycles_per_wave_inst_per_simd_at_reported_clock"], 2) for w, r in v.items()})
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_sequence_batch_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for v in "" bound2 ""; do
  if [ -n "$v" ]; then export LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_$v.so; else unset LIO_HIP_LIB; fi
  timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/bench_short_$v.json 2> $O/bench_short_$v.err
  echo "variant '$v':"; python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print(" ms/scan", d["ms_per_step"], "latency", d["config"].get("single_stream_latency_ms_per_scan"), "knn us", r["avg_launch_us"], r["other_kernels_us"], "touched", r["touched_bytes_per_launch"])
PY
done
unset LIO_HIP_LIB
if [ -n "$LIO_TEST_VARIANT" ]; then LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_$LIO_TEST_VARIANT.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_batch_gpu.py -m gpu -x -q 2>&1 | tail -2; fi
