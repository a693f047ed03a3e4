#!/bin/bash
# round 4, stage s (re-entry after the container was replaced): the whole GPU suite on HEAD (monster sums included), the streaming leg under
# rocprofv3 (the long-run / monster kernel's worst sweeps), then the driver's own command for the line under profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stream.csv \;
rm -rf $O/prof
cd $R
grep "vg_centroid" $O/kernel_stats_stream.csv | cut -c1-60,150-260
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.load(open("$O/bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("single_stream_latency_ms_per_scan"))
d=json.load(open("$O/stream_under_rocprof.json")); print("stream", d["ms_per_step"], d["config"].get("main_ms_median"), d["config"].get("main_ms_p99"))
PY
