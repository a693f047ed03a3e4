#!/bin/bash
# Round-6 evidence on one GPU box: the GPU suite, the driver's bench command (compact line + full record), rocprofv3 kernel statistics of the same
# command (and of the one-round-in-flight form whose per-kernel durations the roofline quotes), the three PMC passes of the kNN kernel.
# Everything lands in gpurun_out/r06final; the summaries are copied to profiles/ by hand (profiles/README.md says which).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06final; mkdir -p $O; cd $R; export TMPDIR=/tmp
if [ "$1" != "noprof" ]; then
cd /tmp
K='knn_batch_kernel<2, false>'
SHORT="--steps 32 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --secondary 0 --upload-scans 0 --groups 1"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_write.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU -d $O/pmc_valu -o r -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_valu.err
cd $R
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*_results.db" | head -1) $(find $O/pmc_write -name "*_results.db" | head -1) "$K" $(find $O/pmc_valu -name "*_results.db" | head -1) > $O/knn_batch_traffic.json
python - <<PY
import json
import re
src = open("$R/bench.py").read()
slots = int(re.search(r'"--slots", type=int, default=(\d+)', src).group(1)); pool = int(re.search(r'"--scan-pool", type=int, default=(\d+)', src).group(1))
p = "$O/knn_batch_traffic.json"; j = json.load(open(p)); j.update(slots_per_launch=slots, scan_pool=pool); json.dump(j, open(p, "w"), indent=1); print(j)
PY
cp $O/knn_batch_traffic.json $R/profiles/knn_batch_traffic.json   # (the bench runs below read it)
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_valu
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o r -- python $R/bench.py --steps 20 --warmup 5 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --secondary 0 --upload-scans 0 --groups 1 > $O/one_round.json 2> $O/prof1.err
find $O/prof1 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_round_in_flight.csv \;
rm -rf $O/prof1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
cp $R/bench_full.json $O/bench_full_under_rocprof.json
cd $R
fi
T0=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $? bytes $(wc -c < $O/bench.json) wall $(( $(date +%s) - T0 )) s"
cp $R/bench_full.json $O/bench_full.json
head -c 1200 $O/bench.json; echo
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout 600 python tools/experiments/lru_stress.py 160 > $O/lru_stress.txt 2>&1; echo "lru stress rc $?"; tail -2 $O/lru_stress.txt
python - <<PY
import csv, os
f = "$O/kernel_stats_one_round_in_flight.csv"
if os.path.exists(f):
    rows = [r for r in csv.DictReader(open(f)) if "lio::" in r["Name"]]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
        print(f"{r['Name'].split('(')[0][:50]:50s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us {float(r['Percentage']):5.1f}%")
PY
