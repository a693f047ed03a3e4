#!/bin/bash
# Round-2 measurement on the GPU box (one gpurun call): the default bench line, rocprofv3 kernel stats of the SAME command, the two PMC
# passes for the batched kNN kernel's memory-side traffic, the config-5 (merge) and config-3 (streaming) lines, matcher timings.
# Everything lands in gpurun_out/r02; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 > $O/prof_bench.json 2> $O/prof.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- python $R/bench.py --steps 32 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --secondary 0 --groups 1 > /dev/null 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- python $R/bench.py --steps 32 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --secondary 0 --groups 1 > /dev/null 2> $O/pmc_write.err
cd $R
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*_results.db" | head -1) $(find $O/pmc_write -name "*_results.db" | head -1) knn_batch_kernel > $O/knn_batch_traffic.json
timeout 600 python bench.py --config merge --steps 8 --warmup 2 > $O/bench_merge.json 2> $O/bench_merge.err; cut -c1-400 $O/bench_merge.json
timeout 900 python bench.py --config stream --steps 2000 --lru 100000 > $O/bench_stream_lru.json 2> $O/bench_stream.err; cut -c1-400 $O/bench_stream_lru.json
timeout 900 python bench.py --config stream --steps 2000 --lru 0 > $O/bench_stream_nolru.json 2>> $O/bench_stream.err; cut -c1-400 $O/bench_stream_nolru.json
timeout 300 python tools/experiments/gicp_time.py > $O/gicp_timing.txt 2>&1
ls $O/prof/*/ 2>/dev/null | head; find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -14 $O/kernel_stats.csv; cat $O/knn_batch_traffic.json | head -12
find $O -name "*_results.db" -delete; rm -rf $O/prof/*/*kernel_trace.csv $O/pmc_fetch $O/pmc_write
