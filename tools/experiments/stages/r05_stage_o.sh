#!/bin/bash
# round 5, call O: XCD-aware chain mapping as the default (+ vg_centroid_batch): parity, A/B against LIO_VG_XCD=0, per-kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; bash tools/gpu_iter.sh "tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_voxelgrid_vs_ref.py tests/test_voxelgrid_monster_gpu.py tests/test_ndt_gpu.py tests/test_sequence_batch_gpu.py tests/test_dist.py" profile
O=$R/gpurun_out/iter
LIO_HIP_LIB=$R/tools/experiments/variants/liblio_hip_novgxcd.so timeout 240 python bench.py --steps 20 --warmup 5 --secondary 0 --min-seconds 2 --cpu-scans 0 --ref-scans 0 --upload-scans 0 > $O/b0.json 2> $O/b0.err
python -c "
import json; d=json.load(open('$R/bench_full.json')); print('plain mapping: ms/scan', d['ms_per_step'], d['roofline']['other_kernels_us'])"
