#!/bin/bash
# round 4: the driver's command with the sequence_batch leg in the line; kernel statistics of the sequence leg (one round in flight)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04l
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 300 $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --config sequences --steps 12 --slots 64 --groups 1 > $O/sequences_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_sequences.csv \;
rm -rf $O/prof
head -c 400 $O/bench.json
