cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/solo; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 32 --warmup 5 --cpu-scans 0 --ref-scans 0 --secondary 0 --groups 1 > /dev/null 2> $O/err.txt
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/solo/kernel_stats.csv')))
for r in rows[:18]:
    print(r['Name'].split('(')[0][-50:], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
rm -rf $O/prof
