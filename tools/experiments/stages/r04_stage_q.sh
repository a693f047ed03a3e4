#!/bin/bash
# round 4: monster sums, five steps in flight: exactness, the stream leg's worst sweeps, the batched chain's long-run kernel on the headline pool
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04q
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_voxelgrid_monster_gpu.py tests/test_voxelgrid_vs_ref.py -m gpu -x -q > $O/pytest_monster.log 2>&1; echo "monster pytest rc $?" | tee -a $O/pytest_monster.log
tail -3 $O/pytest_monster.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stream.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --groups 1 --min-seconds 1 --cpu-scans 0 --ref-scans 0 > $O/bench_one_round.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_round_in_flight.csv \;
rm -rf $O/prof
cd $R
grep "vg_centroid" $O/kernel_stats_stream.csv | cut -c1-60,150-260
grep "vg_centroid" $O/kernel_stats_one_round_in_flight.csv | cut -c1-60,100-260
