export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for P in 0 1 0 1; do
LIO_KNN_PRIOR=$P timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 3 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > gpurun_out/ab/b$P.json 2> gpurun_out/ab/b$P.err
python - <<PY
import json
d = json.load(open("bench_full.json")); r = d["roofline"]
print("PRIOR=$P ms/scan", d["ms_per_step"], r["other_kernels_us"]["knn_per_scan_and_search"], "touched frac", r["frac_touched"], "avg launch", r["avg_launch_us"])
PY
done
timeout 600 python tools/experiments/tf_debug.py 40 > gpurun_out/ab/tf_debug.log 2>&1; tail -42 gpurun_out/ab/tf_debug.log | cut -c1-330
