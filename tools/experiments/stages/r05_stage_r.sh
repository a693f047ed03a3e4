#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05r; mkdir -p $O; cd $R; export TMPDIR=/tmp
for sl in 32 48 64; do
LIO_MERGE_SLOTS=$sl timeout 300 python bench.py --config merge --slots $sl --steps 256 --warmup 64 --scan-pool 64 --min-seconds 2 --ref-scans 0 > $O/merge.json 2> $O/merge.err; python -c "
import json; d=json.load(open('$R/bench_full_merge.json')); print('merge slots $sl ms/scan', d['ms_per_step'], d['roofline']['other_kernels_us'], d['roofline']['avg_launch_us'])"
done
