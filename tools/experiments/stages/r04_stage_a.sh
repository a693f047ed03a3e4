#!/bin/bash
# round 4, stage a: GPU parity suite on the new NDT cost kernel / advice fixes, then the driver's bench command with the new measurement legs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r04a/pytest.log
tail -5 gpurun_out/r04a/pytest.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc $?"
tail -c 600 gpurun_out/r04a/bench.err
head -c 1500 gpurun_out/r04a/bench.json
