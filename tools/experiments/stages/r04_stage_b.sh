#!/bin/bash
# round 4, stage b: parity on the reworked downsample kernels (bbox records, scatter prefix, heads prefetch) and the bucketed device loop;
# phase stamps of the filter-pass kernel (diagnostic build); rocprofv3 kernel statistics of the single-scan streaming path and of the
# localisation leg; the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
LIO_HIP_LIB=$R/lidar-slam-detection_amd/python/lsd_amd/liblio_hip_trace.so timeout 300 python tools/experiments/step_trace.py > $O/step_trace.txt 2>&1
cat $O/step_trace.txt | tail -22
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stream.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config localize --steps 100 --ref-scans 0 --vgicp-scans 0 --scan-pool 16 > $O/localize_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_localize.csv \;
rm -rf $O/prof
grep "lio::" $O/kernel_stats_stream.csv | cut -d, -f1-4 | sed 's/(.*"/"/' | head -40
cd $R
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 300 $O/bench.err
head -c 600 $O/bench.json
