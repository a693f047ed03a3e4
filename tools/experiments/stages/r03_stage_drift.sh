#!/bin/bash
# where does the position error of the long config-3 drive come from: the error against the generating trajectory every 250 sweeps
python bench.py --config stream --grow-to 10000000 --steps 6000 --lru 0 --ref-scans 0 > gpurun_out/stream_drift.json 2> gpurun_out/stream_drift.err
python - <<'P'
import json
j = json.loads(open('gpurun_out/stream_drift.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['pose_error_vs_truth_m'])
print(j['drift'])
P
