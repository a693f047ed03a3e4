#!/bin/bash
# round 4, stage z2: batched radix prefix kernel + prefixed scatter: voxel-grid / batch parity, one-round-in-flight kernel statistics, headline short form
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04z2
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_voxelgrid_vs_ref.py tests/test_voxelgrid_monster_gpu.py tests/test_voxelgrid_crosscheck.py tests/test_golden_gpu.py tests/test_batch_gpu.py tests/test_ndt_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --groups 1 --min-seconds 1 --cpu-scans 0 --ref-scans 0 > $O/one_round.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_round_in_flight.csv \;
rm -rf $O/prof
cd $R
timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json, csv
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("ms/scan", d["ms_per_step"], r["other_kernels_us"])
rows=[r for r in csv.DictReader(open("$O/kernel_stats_one_round_in_flight.csv")) if 'lio::' in r['Name'] and 'batch' in r['Name']]
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])):
    print(f"{r['Name'].split('(')[0][:45]:45s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
