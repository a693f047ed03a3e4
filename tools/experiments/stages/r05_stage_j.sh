#!/bin/bash
# round 5, call J: the new reference-pose fields of configs 3 and 4
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R; export TMPDIR=/tmp
true; python -c "
print('stream leg skipped')"
timeout 600 python bench.py --config localize --steps 200 --scan-pool 32 > $O/loc.json 2> $O/loc.err; python -c "
import json; d=json.load(open('$R/bench_full_localize.json')); print('localize', d['ms_per_step'], d['cpu_baseline'].get('gpu_vs_reference_pose')); print(len(open('$O/loc.json').read()))"
tail -3 $O/loc.err | cut -c1-300
