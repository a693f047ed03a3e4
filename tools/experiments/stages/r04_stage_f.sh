#!/bin/bash
# round 4, stage f: the driver's bench command with the pinned-build parity leg and the grown map's kNN figures; the batched kNN launch geometry
# (workgroups per slot) on the pool of 128 scans
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 300 $O/bench.err
head -c 300 $O/bench.json
for g in 96 128 256 384 512; do
  LIO_KNN_GRID=$g timeout 300 python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-scans 0 --ref-scans 0 --min-seconds 2 > $O/grid_$g.json 2> $O/grid_$g.err
  python - <<P
import json
j = json.loads(open("$O/grid_$g.json").read().strip().splitlines()[-1])
print("LIO_KNN_GRID=$g", j["ms_per_step"], j["roofline"]["other_kernels_us"]["knn_per_scan_and_search"], j["roofline"]["avg_launch_us"])
P
done
