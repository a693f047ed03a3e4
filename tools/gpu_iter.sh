#!/bin/bash
# One measured iteration on a gpurun box in well under a minute of budget:
#     gpurun --timeout 600 -- 'bash tools/gpu_iter.sh "tests/test_gpu_parity.py tests/test_batch_gpu.py" [profile]'
# 1. the named GPU tests (-x -q; default: neighbour sets + batched engine + voxel grid), 2. the headline's short form (7 s: 128-scan pool, 64 x 4,
# no secondary legs, no CPU baselines) with its one-round-in-flight kernel times, 3. with `profile`: the same command with one round in flight under
# rocprofv3 --kernel-trace --stats, the lio:: batch kernels' averages printed.  LIO_HIP_LIB=<variant .so> selects a variant build of the library.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/iter
mkdir -p $O
cd $R
export TMPDIR=/tmp
TESTS=${1:-tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_voxelgrid_vs_ref.py tests/test_voxelgrid_monster_gpu.py}
timeout 800 python -m pytest $TESTS -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print("ms/scan", d["ms_per_step"], "single-stream latency", d["config"].get("single_stream_latency_ms_per_scan"), r["other_kernels_us"])
PY
if [ "$2" = "profile" ]; then
    cd /tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --groups 1 --min-seconds 1 --cpu-scans 0 --ref-scans 0 > $O/one_round.json 2> $O/prof.err
    find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_round_in_flight.csv \;
    rm -rf $O/prof
    python - <<PY
import csv
rows = [r for r in csv.DictReader(open("$O/kernel_stats_one_round_in_flight.csv")) if "lio::" in r["Name"] and "batch" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print(f"{r['Name'].split('(')[0][:45]:45s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
fi
