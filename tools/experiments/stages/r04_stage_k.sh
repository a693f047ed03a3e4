#!/bin/bash
# round 4: lio_batch_fastlio_main (the reference's entry points for all sessions of a sequence batch) + slots / groups sweep of the sequence leg
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sequence_batch_gpu.py tests/test_frontend_gpu.py -m gpu -x -q > $O/pytest_seq.log 2>&1; echo "seq pytest rc $?" | tee -a $O/pytest_seq.log
tail -30 $O/pytest_seq.log
for cfg in "64 1" "64 2" "128 1" "32 4"; do
  set -- $cfg
  timeout 600 python bench.py --config sequences --steps 24 --slots $1 --groups $2 > $O/bench_sequences_$1x$2.json 2> $O/bench_sequences_$1x$2.err; echo "bench $1 x $2 rc $?"
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_sequences_$1x$2.json"))
    print("$1 x $2", d["ms_per_step"], d["device_us_per_round"], d["one_session_at_a_time"]["ms_per_sweep"], d["parity"]["bit_identical_to_the_per_session_engine"])
except Exception as e:
    print("$1 x $2", "failed", e)
PY
done
