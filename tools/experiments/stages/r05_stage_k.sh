#!/bin/bash
# round 5, call K: strided classify_seq kernels: parity of the sequence batch + its leg at several geometries
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sequence_batch_gpu.py tests/test_lru_gpu.py tests/test_frontend_gpu.py -m gpu -x -q 2>&1 | tail -2
for sg in "64 2" "128 1" "128 2" "96 2"; do
  set -- $sg
  timeout 300 python bench.py --config sequences --steps 24 --slots $1 --groups $2 --cpu-scans 0 > $O/seq.json 2> $O/seq.err; python -c "
import json; d=json.load(open('$R/bench_full_sequences.json')); print('sequences $1 x $2 ms/sweep', d['ms_per_step'], d.get('device_us_per_round'), d.get('parity',{}).get('bit_identical_to_the_per_session_engine'))"
done
