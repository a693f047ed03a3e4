#!/bin/bash
# round 4, stage d: parity after the speculative device LM loop / the reverted histogram fusion; kernel statistics of the headline command with ONE
# round in flight (the batched kernels' own durations on the pool of 128 scans), of the streaming leg and of the localisation leg
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --groups 1 --min-seconds 1 --cpu-scans 0 --ref-scans 0 > $O/one_round_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_round_in_flight.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stream.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config localize --steps 100 --ref-scans 0 --vgicp-scans 0 --scan-pool 16 > $O/localize_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_localize.csv \;
rm -rf $O/prof
grep "lio::" $O/kernel_stats_one_round_in_flight.csv | cut -d, -f1-4 | sed 's/(.*"/"/' | head -30
