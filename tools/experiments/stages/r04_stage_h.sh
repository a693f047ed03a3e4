#!/bin/bash
# round 4, final evidence: the gpu suite, the driver's bench command, rocprofv3 kernel statistics of the same command (three rounds in flight + the
# secondary legs), of one round in flight, of the streaming / localisation / merge legs.  Lands in gpurun_out/r04h; copied to profiles/r04_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 200 $O/bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 > $O/bench_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --groups 1 --min-seconds 1 --cpu-scans 0 --ref-scans 0 > /dev/null 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_round_in_flight.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stream.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config localize --steps 100 --ref-scans 0 --vgicp-scans 0 --scan-pool 16 > $O/localize_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_localize.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config merge --steps 64 --warmup 32 --scan-pool 32 --min-seconds 1 --ref-scans 0 > $O/merge_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_merge.csv \;
rm -rf $O/prof
head -c 300 $O/bench.json
