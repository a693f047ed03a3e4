#!/bin/bash
# round 4, stage z: centroid kernel's loads grouped: voxel-grid parity, then the kernel statistics of the headline command with ONE round in flight
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04z
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_voxelgrid_vs_ref.py tests/test_voxelgrid_monster_gpu.py tests/test_voxelgrid_crosscheck.py tests/test_golden_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --groups 1 --min-seconds 1 --cpu-scans 0 --ref-scans 0 > $O/one_round.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_round_in_flight.csv \;
rm -rf $O/prof
cd $R
grep "lio::" $O/kernel_stats_one_round_in_flight.csv | cut -d, -f1-4 | sed 's/(.*"/"/' | head -24
