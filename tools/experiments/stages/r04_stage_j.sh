#!/bin/bash
# round 4: sequence mode of the batched engine (B sessions with their own maps, map_incremental inside the round): its parity test, the whole gpu
# suite (the insert / classify kernels were refactored into shared bodies), the bench leg
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sequence_batch_gpu.py -m gpu -x -q > $O/pytest_seq.log 2>&1; echo "seq pytest rc $?" | tee -a $O/pytest_seq.log
tail -30 $O/pytest_seq.log
timeout 600 python bench.py --config sequences --steps 32 --slots 32 --groups 2 > $O/bench_sequences.json 2> $O/bench_sequences.err; echo "bench rc $?"
tail -c 600 $O/bench_sequences.err
head -c 1500 $O/bench_sequences.json
timeout 1300 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
