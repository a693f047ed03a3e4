#!/bin/bash
# round 4, stage g: the eight-lanes-per-point NDT evaluation -- parity (the matcher's tests), the localisation leg, its kernel statistics
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ndt_gpu.py tests/test_ndt_vs_ref_cuda.py tests/test_golden_gpu.py tests/test_overlap_merge_gpu.py tests/test_localization_boundary.py tests/test_outer_boundary.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --config localize --steps 200 --scan-pool 32 > $O/localize.json 2> $O/localize.err; echo "localize rc $?"
LIO_NDT_L8=0 timeout 600 python bench.py --config localize --steps 200 --scan-pool 32 --ref-scans 0 --vgicp-scans 0 > $O/localize_l1.json 2> $O/localize_l1.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config localize --steps 100 --ref-scans 0 --vgicp-scans 0 --scan-pool 16 > $O/localize_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_localize.csv \;
rm -rf $O/prof
python - <<P
import json
for f in ("localize", "localize_l1"):
    j = json.loads([l for l in open("$O/%s.json" % f) if l.startswith("{")][-1]); c = j["config"]
    for k in ("resident_map", "resident_map_one_spot_pool", "local_200k_map"):
        d = c[k]; r = d.get("roofline") or j["roofline"]
        print(f, k, d["ms_per_scan"], d["converged"], r["avg_launch_us"], r["evaluations_per_alignment"], {kk: (vv["ms_per_scan"], vv["max_abs_difference_from_the_single_scan_results"]) for kk, vv in d["batched"].items() if kk != "what"})
P
grep "ndt_" $O/kernel_stats_localize.csv | cut -d, -f1-4 | sed 's/(.*"/"/'
