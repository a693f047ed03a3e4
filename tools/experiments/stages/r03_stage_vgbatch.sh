#!/bin/bash
echo skip-tests
python bench.py --config localize --steps 200 > gpurun_out/localize.json 2> gpurun_out/localize.err
python - <<'P'
import json
j = json.loads(open('gpurun_out/localize.json').read().strip().splitlines()[-1])
c = j['config']
for k in ('resident_map', 'local_200k_map'):
    print(k, c[k]['ms_per_scan'], c[k].get('batched'))
P
tail -3 gpurun_out/localize.err
