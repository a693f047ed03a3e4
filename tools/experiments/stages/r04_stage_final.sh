#!/bin/bash
# round 4, final stage (second session): the whole GPU suite on the load-grouped kernels, the PMC passes of the batched kNN kernel again (its
# instruction stream changed), the driver's own command for the line under profiles/, then the kernel statistics of the headline command with
# one round in flight, of the streaming leg and of the localisation leg.  Ordered by what must not be lost if the budget ends first.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
cd /tmp
P="python $R/bench.py --steps 128 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --secondary 0 --groups 1"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $P > /dev/null 2> $O/pmc_fetch.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- $P > /dev/null 2> $O/pmc_write.err
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU -d $O/pmc_valu -o r -- $P > /dev/null 2> $O/pmc_valu.err
python $R/tools/pmc_traffic.py $(find $O/pmc_fetch -name "*_results.db" | head -1) $(find $O/pmc_write -name "*_results.db" | head -1) "knn_batch_kernel<2, false>" $(find $O/pmc_valu -name "*_results.db" | head -1) > $O/knn_batch_traffic.json
python - <<P2
import json, re
j = json.load(open("$O/knn_batch_traffic.json"))
src = open("$R/bench.py").read()
j["slots_per_launch"] = int(re.search(r'"--slots", type=int, default=(\d+)', src).group(1))  # the launches sampled were bench.py's defaults
j["scan_pool"] = int(re.search(r'"--scan-pool", type=int, default=(\d+)', src).group(1))
json.dump(j, open("$O/knn_batch_traffic.json", "w"), indent=1)
if j.get("hbm_bytes_per_launch", 0) > 0 and j.get("valu_wave_instructions_per_launch", 0) > 0:
    json.dump(j, open("$R/profiles/knn_batch_traffic.json", "w"), indent=1)   # what the bench run below reports as roofline.traffic / valu
P2
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_valu
cat $O/knn_batch_traffic.json
cd $R
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 200 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("value", d["value"], "ms/scan", d["ms_per_step"], "frac", r["frac"], "valu", r["valu"]["frac_of_valu_issue_peak"], "latency", d["config"].get("single_stream_latency_ms_per_scan"))
PY
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --groups 1 --min-seconds 1 --cpu-scans 0 --ref-scans 0 > $O/one_round_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_round_in_flight.csv \;
rm -rf $O/prof
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 > $O/headline_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_headline_short.csv \;
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stream.csv \;
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config localize --steps 100 --ref-scans 0 --vgicp-scans 0 --scan-pool 16 > $O/localize_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_localize.csv \;
rm -rf $O/prof
echo done
