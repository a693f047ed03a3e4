#!/bin/bash
# round 5, first GPU call: new tests, VALU peak, kNN patch 01 A/B, compact line of the default run
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 120 python tools/valu_peak/run.py > $O/valu_peak.json 2>&1; head -c 1500 $O/valu_peak.json
timeout 600 python -m pytest tests/test_batch_gpu.py tests/test_gpu_parity.py tests/test_voxelgrid_monster_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for v in "" knn01; do
  if [ -n "$v" ]; then export LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_$v.so; else unset LIO_HIP_LIB; fi
  timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 > $O/bench_short_$v.json 2> $O/bench_short_$v.err
  echo "variant '$v':"; python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print(" ms/scan", d["ms_per_step"], "latency", d["config"].get("single_stream_latency_ms_per_scan"), "knn us", r["avg_launch_us"], r["other_kernels_us"], "upload", d.get("upload_included"))
print(" frac", r["frac"], "alg", r["frac_algorithmic"], "valu", r["frac_valu"], r["valu"].get("peak_measured"))
PY
  cp $R/bench_full.json $O/bench_full_short_$v.json
done
unset LIO_HIP_LIB
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "default run rc $? line bytes $(wc -c < $O/bench_line.json)"
cp $R/bench_full.json $O/bench_full.json
head -c 3000 $O/bench_line.json
