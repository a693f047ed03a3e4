#!/bin/bash
# round 5, call U: rocprofv3 kernel statistics of the secondary legs (stream single scan, localize, merge, sequences)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05u; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
prof() {  # name, bench args...
  name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$name -o r -- python $R/bench.py "$@" > $O/$name.json 2> $O/$name.err
  find $O/p_$name -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$name.csv \;
  rm -rf $O/p_$name
  python - <<PY
import csv
rows = [r for r in csv.DictReader(open("$O/kernel_stats_$name.csv")) if "lio::" in r["Name"]]
print("== $name")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:10]:
    print(f"  {r['Name'].split('(')[0][:50]:50s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
}
prof stream_single_scan --config stream --steps 400 --lru 100000 --ref-scans 0
prof localize --config localize --steps 200 --scan-pool 32 --ref-scans 0 --vgicp-scans 0
prof merge --config merge --steps 256 --warmup 64 --scan-pool 64 --min-seconds 2 --ref-scans 0
prof sequences --config sequences --steps 24 --slots 128 --groups 2 --cpu-scans 0
