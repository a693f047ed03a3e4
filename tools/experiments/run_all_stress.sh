#!/bin/bash
# every randomised sweep against the oracle / the compiled reference, one after the other (a GPU box; ~6 minutes): bash tools/experiments/run_all_stress.sh [seed]
# Each prints one summary line; the exit code is the number of sweeps that found a mismatch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
S=${1:-0}
fail=0
run() { echo "== $*"; python "$@" 2>&1 | tail -3; [ ${PIPESTATUS[0]} -eq 0 ] || fail=$((fail + 1)); }
run tools/experiments/lru_stress.py 160
run tools/experiments/lru_tie_stress.py 24
NO_LRU=1 run tools/experiments/lru_tie_stress.py 24
run tools/experiments/vg_stress.py 400 $S
run tools/experiments/knn_stress.py 100 $S
run tools/experiments/engine_stress.py 30 $S
run tools/experiments/frontend_stress.py 12 $S
run tools/experiments/seq_stress.py 8 $S
run tools/experiments/batch_stress.py 10 $S
run tools/experiments/ndt_stress.py 40 $S
run tools/experiments/ndt_batch_stress.py 8 $S
run tools/experiments/gicp_stress.py 10 $S
echo "sweeps with a mismatch: $fail"
exit $fail
