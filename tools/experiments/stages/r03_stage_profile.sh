#!/bin/bash
# Round-3 profiles (one stage of tools/r03_gpu_session.sh): rocprofv3 kernel stats of the default bench command (N = 1) and of the streaming
# single-scan path, the two PMC passes for the batched kNN kernel's memory-side traffic.  Lands in gpurun_out/r03; copy into profiles/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 > $O/bench_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --groups 1 --min-seconds 1 --cpu-scans 0 --ref-scans 0 > /dev/null 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_round_in_flight.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stream.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config localize --steps 100 --ref-scans 0 --vgicp-scans 0 > $O/localize_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_localize.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config merge --steps 40 --warmup 8 --min-seconds 1 > $O/merge_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_merge.csv \;
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- python $R/bench.py --steps 32 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --secondary 0 --groups 1 > /dev/null 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- python $R/bench.py --steps 32 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --secondary 0 --groups 1 > /dev/null 2> $O/pmc_write.err
python $R/tools/pmc_traffic.py $(find $O/pmc_fetch -name "*_results.db" | head -1) $(find $O/pmc_write -name "*_results.db" | head -1) "knn_batch_kernel<2, false>" > $O/knn_batch_traffic.json
python - <<P
import json, re
j = json.load(open("$O/knn_batch_traffic.json"))
j["slots_per_launch"] = int(re.search(r'"--slots", type=int, default=(\d+)', open("$R/bench.py").read()).group(1))  # the launches sampled were bench.py's default
json.dump(j, open("$O/knn_batch_traffic.json", "w"), indent=1)
P
rm -rf $O/pmc_fetch $O/pmc_write
head -12 $O/kernel_stats_stream.csv; cat $O/knn_batch_traffic.json
