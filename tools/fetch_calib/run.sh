#!/bin/bash
# FETCH_SIZE calibration for the kNN sweep's access shape (tools/fetch_calib/fetch_calib.hip), on the GPU box:
#     gpurun -- 'bash tools/fetch_calib/run.sh'      -> gpurun_out/fetch_calib/calibration.json  (copy to profiles/)
# one rocprofv3 --pmc FETCH_SIZE pass (counters only: no trace domain beside it)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fetch_calib
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib/fetch_calib.hip -o /tmp/fetch_calib || exit 1
/tmp/fetch_calib 1 > $O/expected.jsonl || exit 1
rm -rf $O/pmc
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc -o calib -- /tmp/fetch_calib 3 > $O/run.log 2>&1
DB=$(find $O/pmc -name "*results.db" | head -1)
python - "$DB" "$O/expected.jsonl" > $O/calibration.json <<'PY'
import json, sqlite3, sys
db, exp = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
out = {"what": "rocprofv3 --pmc FETCH_SIZE on tools/fetch_calib/fetch_calib.hip (every byte read once from an 8 GiB buffer: no cache can help); FETCH_SIZE is in KiB",
       "kernels": {}}
for line in open(exp):
    e = json.loads(line)
    k = e["kernel"]
    rows = list(c.execute("select avg(value), count(*) from counters_collection where counter_name = 'FETCH_SIZE' and kernel_name like ?", (f"%{k.split('<')[0]}%{'<' + k.split('<')[1] if '<' in k else ''}%",)))
    kib, n = rows[0]
    if not kib:
        continue
    b = kib * 1024.0
    out["kernels"][k] = dict(e, fetch_size_bytes_reported=b, launches=n,
                             requested_over_reported=round(e["requested_bytes"] / b, 4),
                             sectors64_over_reported=round(e["distinct_64B_sector_bytes"] / b, 4),
                             lines128_over_reported=round(e["distinct_128B_line_bytes"] / b, 4))
print(json.dumps(out, indent=1))
PY
cat $O/calibration.json
rm -rf $O/pmc
