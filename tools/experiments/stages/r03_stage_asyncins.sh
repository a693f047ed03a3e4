#!/bin/bash
python -m pytest tests/test_lru_gpu.py tests/test_frontend_gpu.py tests/test_gpu_parity.py tests/test_fastlio_golden.py tests/test_outer_boundary.py tests/test_wheelspeed.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -40
python bench.py --config stream --grow-to 10000000 --steps 6000 --lru 0 --ref-scans 0 > gpurun_out/stream_async.json 2> gpurun_out/stream_async.err
python bench.py --config stream --steps 300 --lru 100000 --ref-scans 0 > gpurun_out/stream_async_lru.json 2>> gpurun_out/stream_async.err
python - <<'P'
import json
for f in ('stream_async', 'stream_async_lru'):
    j = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    c = j['config']
    print(f, j['ms_per_step'], 'main', c.get('main_ms_median'), 'enq', c.get('enqueue_ms_median'), 'evicted', c.get('voxels_evicted'), 'map', c.get('map_points_end'), 'err', j['pose_error_vs_truth_m'], 'added', c.get('points_added_per_scan'))
P
