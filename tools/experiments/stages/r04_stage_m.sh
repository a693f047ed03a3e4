#!/bin/bash
# round 4: sequence rounds as graphs, drives kept clear of the boxes, oracle baseline in the leg; then the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sequence_batch_gpu.py -m gpu -x -q > $O/pytest_seq.log 2>&1; echo "seq pytest rc $?" | tee -a $O/pytest_seq.log
tail -5 $O/pytest_seq.log
timeout 600 python bench.py --config sequences --steps 24 --slots 64 --groups 2 > $O/bench_sequences.json 2> $O/bench_sequences.err; echo "bench seq rc $?"
tail -c 300 $O/bench_sequences.err
LIO_BATCH_GRAPH=0 timeout 600 python bench.py --config sequences --steps 24 --slots 64 --groups 2 --cpu-scans 0 > $O/bench_sequences_nograph.json 2>> $O/bench_sequences.err; echo "bench seq nograph rc $?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config sequences --steps 12 --slots 64 --groups 1 --cpu-scans 0 > $O/sequences_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_sequences.csv \;
rm -rf $O/prof
cd $R
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 300 $O/bench.err
python - <<PY
import json
for f in ["bench_sequences","bench_sequences_nograph"]:
    d=json.load(open("$O/"+f+".json"))
    print(f, d["ms_per_step"], d["device_us_per_round"], d["one_session_at_a_time"]["ms_per_sweep"], d["parity"]["bit_identical_to_the_per_session_engine"], d.get("pose_error_vs_truth"), d.get("cpu_baseline"))
PY
