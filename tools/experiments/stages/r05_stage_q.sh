#!/bin/bash
# round 5, call Q: joint mode with ONE neighbour-search / linearisation launch per pass over all [sub-map x slot] rows: parity + the merge leg
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dist.py tests/test_batch_gpu.py tests/test_overlap_merge_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --config merge --steps 256 --warmup 64 --scan-pool 64 --min-seconds 2 --ref-scans 0 > $O/merge.json 2> $O/merge.err; python -c "
import json; d=json.load(open('$R/bench_full_merge.json')); print('merge ms/scan', d['ms_per_step'], d['roofline']['other_kernels_us'], d['roofline']['avg_launch_us'], d['collective'].get('states_identical_on_all_ranks'), d['pose_error_vs_truth_m'])"
