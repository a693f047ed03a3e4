#!/bin/bash
# round 4: monster-voxel sums (exact, wave per coordinate), then the whole gpu suite, the sequence leg, the stream leg's kernel statistics
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04n
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_voxelgrid_monster_gpu.py tests/test_voxelgrid_vs_ref.py -m gpu -x -q > $O/pytest_monster.log 2>&1; echo "monster pytest rc $?" | tee -a $O/pytest_monster.log
tail -15 $O/pytest_monster.log
timeout 1300 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
timeout 600 python bench.py --config sequences --steps 24 --slots 64 --groups 2 > $O/bench_sequences.json 2> $O/bench_sequences.err; echo "bench seq rc $?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stream.csv \;
rm -rf $O/prof
cd $R
python - <<PY
import json
d=json.load(open("$O/bench_sequences.json"))
print(d["ms_per_step"], d["device_us_per_round"], d["one_session_at_a_time"]["ms_per_sweep"], d["parity"]["bit_identical_to_the_per_session_engine"], d.get("pose_error_vs_truth"), d["config"]["passes_avg"])
d=json.load(open("$O/stream_under_rocprof.json"))
print(d["ms_per_step"], d["config"].get("main_ms_median"))
PY
grep "vg_centroid_long" $O/kernel_stats_stream.csv | cut -c1-200
