R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R; export TMPDIR=/tmp
for v in "" occ5 occ4; do
 for g in 4 6; do
  if [ -n "$v" ]; then export LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_$v.so; else unset LIO_HIP_LIB; fi
  timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 --groups $g > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print("variant '$v' groups $g: ms/scan", d["ms_per_step"], "knn us", r["avg_launch_us"], "dev/scan", r["other_kernels_us"]["device_time_per_scan_one_round_in_flight"])
PY
 done
done
