#!/bin/bash
# round 5, call S: one NDT cost launch per LM round of the batched alignments: parity + the localisation leg
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05s; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ndt_gpu.py tests/test_overlap_merge_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --config localize --steps 200 --scan-pool 32 --ref-scans 0 --vgicp-scans 0 > $O/loc.json 2> $O/loc.err; python -c "
import json; d=json.load(open('$R/bench_full_localize.json'))
for n in ('resident_map','resident_map_one_spot_pool','local_200k_map'):
    c=d[n]; print(n, c['ms_per_scan'], {k:(v['ms_per_scan'], v['align_ms_per_scan'], v['max_abs_difference_from_the_single_scan_results']) for k,v in c['batched'].items() if isinstance(v,dict)})
print(d.get('merge_candidates_batched'))"
