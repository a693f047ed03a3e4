#!/bin/bash
# round 5, call T: memory-side traffic (PMC) of the chain's tile kernels with the plain and with the XCD-aware (slot, tile) mapping
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05t; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SHORT="--steps 32 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --secondary 0 --upload-scans 0 --groups 1"
for v in xcd plain; do
  if [ "$v" = "plain" ]; then export LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_novgxcd.so; else unset LIO_HIP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f_$v -o r -- python $R/bench.py $SHORT > /dev/null 2> $O/f_$v.err
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w_$v -o r -- python $R/bench.py $SHORT > /dev/null 2> $O/w_$v.err
done
unset LIO_HIP_LIB
cd $R
python - <<PY
import json, sqlite3, glob
def avg(db, counter, kernel):
    c = sqlite3.connect(db)
    r = list(c.execute("select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like ?", (counter, f"%{kernel}%")))
    return (r[0][0] or 0.0), r[0][1]
out = {}
for v in ("plain", "xcd"):
    f = glob.glob("$O/f_%s/**/*_results.db" % v, recursive=True)[0]; w = glob.glob("$O/w_%s/**/*_results.db" % v, recursive=True)[0]
    for k in ("radix_scatter_batch", "vg_heads_batch", "radix_hist_batch", "vg_keys_batch", "vg_count_heads_batch", "vg_centroid_batch", "vg_centroid_both_batch", "vg_bbox_batch"):
        fk, n = avg(f, "FETCH_SIZE", k); wk, _ = avg(w, "WRITE_SIZE", k)
        out.setdefault(k, {})[v] = {"fetch_MB_x2": round(2 * fk * 1024 / 1e6, 1), "write_MB": round(wk * 1024 / 1e6, 1), "launches": n}
json.dump({"what": "memory-side bytes per launch of 128 scans (FETCH_SIZE x 2, WRITE_SIZE; separate --pmc passes) of the batched chain's kernels, plain (slot = blockIdx.y) vs XCD-aware (all tiles of a scan on one XCD) mapping", "kernels": out}, open("$O/chain_traffic_xcd.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
rm -rf $O/f_* $O/w_*
