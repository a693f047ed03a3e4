#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-scans 0 --ref-scans 0 > gpurun_out/bench_knnopt.json 2> gpurun_out/bench_knnopt.err
python - <<'P'
import json
b = json.loads(open('gpurun_out/bench_knnopt.json').read().strip().splitlines()[-1])
r = b['roofline']
print('ms/scan', b['ms_per_step'], 'knn launch us', r['avg_launch_us'], r['other_kernels_us'], 'lat', b['config'].get('single_stream_latency_ms_per_scan'))
P
