#!/bin/bash
# round 4: monster sums with four steps in flight and a DPP scan: exactness tests, the stream leg's kernel statistics
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04o
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_voxelgrid_monster_gpu.py tests/test_voxelgrid_vs_ref.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/pytest_monster.log 2>&1; echo "monster pytest rc $?" | tee -a $O/pytest_monster.log
tail -5 $O/pytest_monster.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream_under_rocprof.json 2>> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stream.csv \;
rm -rf $O/prof
cd $R
grep "vg_centroid" $O/kernel_stats_stream.csv | cut -c1-60,150-260
timeout 600 python bench.py --config stream --steps 400 --lru 100000 --ref-scans 0 > $O/stream.json 2>> $O/prof.err
python - <<PY
import json
for f in ["stream_under_rocprof","stream"]:
    d=json.load(open("$O/"+f+".json")); print(f, d["ms_per_step"], d["config"].get("main_ms_median"))
PY
