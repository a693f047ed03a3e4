#!/bin/bash
# round 4: the driver's bench command once more with the final bench.py (the grown-map kNN leg of config 3 with the engine's own poses as priors).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 300 $O/bench.err
head -c 300 $O/bench.json
