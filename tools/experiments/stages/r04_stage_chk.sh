#!/bin/bash
# round 4: the streaming leg alone, twice (is the driver line's slower config 3 the box or the code?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04chk
mkdir -p $O
cd $R
export TMPDIR=/tmp
for k in 1 2; do
timeout 200 python bench.py --config stream --steps 300 --lru 100000 --ref-scans 0 > $O/stream$k.json 2> $O/stream$k.err
python - <<PY
import json
d = json.load(open("$O/stream$k.json")); c = d["config"]
print("stream", d["ms_per_step"], c.get("main_ms_median"), c.get("main_ms_p99"), (d.get("roofline") or {}).get("stage_us_per_scan"))
PY
done
