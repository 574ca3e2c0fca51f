#!/bin/bash
# the push tail of a source group's loop on the headline: bound on an iteration's in-edges (default 200 per sweep group = 615 K) and entry threshold
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06/gpush; mkdir -p $OUT; cd $ROOT
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings"
run() { python3 bench.py $Q $2 > $OUT/$1.json 2> $OUT/$1.err; python3 - $OUT/$1.json $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:8.4f} event {d['event_ms_per_step']:8.4f} iters {d['iterations_per_step']} parity {d['parity']['ok']}")
except Exception as ex: print(sys.argv[2], 'FAILED', ex)
PY
}
for rep in 1 2; do
run default_$rep ""
run max40k_$rep "--tune gpush_max_edges=40000"
run max60k_$rep "--tune gpush_max_edges=60000"
run max90k_$rep "--tune gpush_max_edges=90000"
run max60k_enter3k_$rep "--tune gpush_max_edges=60000 --tune gpush_enter_pairs=3000"
run max60k_enter4500_$rep "--tune gpush_max_edges=60000 --tune gpush_enter_pairs=4500"
run max60k_enter9k_$rep "--tune gpush_max_edges=60000 --tune gpush_enter_pairs=9000"
done
