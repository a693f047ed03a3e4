#!/bin/bash
# Round-end measurement on the GPU box: gpu tests, the default bench line, rocprofv3 kernel stats of the SAME command, the
# two PMC passes for the kNN kernel's memory-side traffic, front-half and matcher timings.  Everything lands in gpurun_out/final.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/bench.py > $O/prof_bench.json 2> $O/prof.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- python $R/bench.py --steps 32 --warmup 8 --cpu-scans 0 --streams 1 > /dev/null 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- python $R/bench.py --steps 32 --warmup 8 --cpu-scans 0 --streams 1 > /dev/null 2> $O/pmc_write.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_front -o r -- python $R/tools/bench_frontend.py --scans 40 --timing 0 > $O/frontend.json 2> $O/prof_front.err
cd $R
python tools/rocprof_summary.py $(find $O/prof -name "*_results.db" | head -1) "command: rocprofv3 --kernel-trace --stats -- python bench.py (defaults: up to 12 streams, 200 steps, 20 warm-up)" > $O/kernel_stats.txt
python tools/rocprof_summary.py $(find $O/prof_front -name "*_results.db" | head -1) "command: rocprofv3 --kernel-trace --stats -- python tools/bench_frontend.py --scans 40 --timing 0" > $O/frontend_kernel_stats.txt
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*_results.db" | head -1) $(find $O/pmc_write -name "*_results.db" | head -1) > $O/knn_traffic.json
timeout 300 python tools/bench_frontend.py --scans 40 > $O/frontend_timing.json 2>&1
timeout 600 python tools/bench_ndt.py > $O/ndt.jsonl 2>&1
head -12 $O/kernel_stats.txt; cat $O/knn_traffic.json | head -12; tail -1 $O/frontend.json | cut -c1-300; tail -2 $O/ndt.jsonl | cut -c1-300
find $O -name "*_results.db" -delete
