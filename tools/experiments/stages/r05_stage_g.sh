#!/bin/bash
# round 5, call G: exact-redo kernel with six voxels per step: parity (ties), one-round kernel statistics; merge / sequences legs after the grid-stride linearize
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; bash tools/gpu_iter.sh "tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_lru_gpu.py tests/test_fullsize_gpu.py" profile
O=$R/gpurun_out/iter
timeout 300 python bench.py --config merge --steps 256 --warmup 64 --scan-pool 64 --min-seconds 2 --ref-scans 0 > $O/merge.json 2> $O/merge.err; python -c "
import json; d=json.load(open('$R/bench_full_merge.json')); print('merge ms/scan', d['ms_per_step'], d['roofline']['other_kernels_us'], d['roofline']['avg_launch_us'])"
timeout 300 python bench.py --config sequences --steps 24 --slots 64 --groups 2 --cpu-scans 0 > $O/seq.json 2> $O/seq.err; python -c "
import json; d=json.load(open('$R/bench_full_sequences.json')); print('sequences ms/sweep', d['ms_per_step'], d.get('device_us_per_round'))"
