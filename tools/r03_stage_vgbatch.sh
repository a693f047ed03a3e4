#!/bin/bash
python -m pytest tests/test_ndt_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
python bench.py --config localize --steps 200 > gpurun_out/localize.json 2> gpurun_out/localize.err
python - <<'P'
import json
j = json.loads(open('gpurun_out/localize.json').read().strip().splitlines()[-1])
c = j['config']
for k in ('resident_map', 'local_200k_map'):
    print(k, c[k]['ms_per_scan'], c[k].get('batched_32_scans_per_call'))
P
tail -3 gpurun_out/localize.err
