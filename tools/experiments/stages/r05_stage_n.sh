#!/bin/bash
# round 5, call N: XCD-aware (slot, tile) mapping of the batched chain's tile kernels (LIO_VG_XCD=1): parity + A/B + per-kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_voxelgrid_vs_ref.py tests/test_voxelgrid_monster_gpu.py -m gpu -x -q 2>&1 | tail -2
LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_vgxcd.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_voxelgrid_vs_ref.py tests/test_voxelgrid_monster_gpu.py tests/test_ndt_gpu.py -m gpu -x -q 2>&1 | tail -2
for v in "" vgxcd "" vgxcd; do
  if [ -n "$v" ]; then export LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_$v.so; else unset LIO_HIP_LIB; fi
  timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.load(open("$R/bench_full.json")); r = d["roofline"]
print("variant '$v': ms/scan", d["ms_per_step"], "chain/round", r["other_kernels_us"]["downsample_chain_per_round"], "dev/scan", r["other_kernels_us"]["device_time_per_scan_one_round_in_flight"])
PY
done
cd /tmp
LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_vgxcd.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --groups 1 --min-seconds 1 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/one.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_vgxcd.csv \;
rm -rf $O/prof
python - <<PY
import csv
rows = [r for r in csv.DictReader(open("$O/kernel_stats_vgxcd.csv")) if "lio::" in r["Name"] and "batch" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"{r['Name'].split('(')[0][:45]:45s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
