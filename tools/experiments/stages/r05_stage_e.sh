#!/bin/bash
# round 5, call E: grid-stride linearize_batch / vg_centroid (capped grids): parity + short bench + one-round kernel statistics
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; bash tools/gpu_iter.sh "tests/test_gpu_parity.py tests/test_batch_gpu.py tests/test_voxelgrid_vs_ref.py tests/test_voxelgrid_monster_gpu.py tests/test_sequence_batch_gpu.py tests/test_dist.py tests/test_ndt_gpu.py" profile
