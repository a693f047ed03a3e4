#!/bin/bash
# one gpurun call of round 3: stages named on the command line, logs under gpurun_out/<stage>.log (merged back by gpurun)
#   tools/r03_gpu_session.sh tests stream localize bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for stage in "$@"; do
  t0=$(date +%s)
  case "$stage" in
    tests)    timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests.log 2>&1 ;;
    tests_all) timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/tests.log 2>&1 ;;
    tests_fix) timeout 1500 python -m pytest tests/test_localization_boundary.py tests/test_outer_boundary.py tests/test_ndt_gpu.py -m gpu -q -s > gpurun_out/tests_fix.log 2>&1 ;;
    batch)    timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -s > gpurun_out/batch.log 2>&1 ;;
    stream)   timeout 900 python bench.py --config stream --grow-to 10000000 --steps 6000 --lru 0 > gpurun_out/stream.json 2> gpurun_out/stream.err ;;
    stream_lru) timeout 600 python bench.py --config stream --steps 300 --lru 100000 --ref-scans 0 > gpurun_out/stream_lru.json 2> gpurun_out/stream_lru.err ;;
    localize) timeout 900 python bench.py --config localize --steps 200 > gpurun_out/localize.json 2> gpurun_out/localize.err ;;
    bench)    timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err ;;
    bench_fast) timeout 900 python bench.py --steps 20 --warmup 5 --secondary 0 > gpurun_out/bench_fast.json 2> gpurun_out/bench_fast.err ;;
    merge)    timeout 600 python bench.py --config merge --steps 40 --warmup 8 > gpurun_out/merge.json 2> gpurun_out/merge.err ;;
    gicp)     timeout 600 python tools/experiments/gicp_time.py > gpurun_out/gicp_timing.txt 2>&1 ;;
    smoke)    timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 ;;
    *)        if [ -x "tools/r03_stage_$stage.sh" ]; then timeout 1500 "tools/r03_stage_$stage.sh" > "gpurun_out/$stage.log" 2>&1; else echo "unknown stage $stage"; fi ;;
  esac
  echo "stage $stage rc=$? $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/session.log
done
tail -3 gpurun_out/tests.log 2>/dev/null
