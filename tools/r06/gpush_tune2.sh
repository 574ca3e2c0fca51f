#!/bin/bash
# the same bound (in-edges per sweep group: 200 -> 20) on the twitter / friendster groups and on a small window's group
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06/gpush2; mkdir -p $OUT; cd $ROOT
Q="--no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings"
run() { python3 bench.py $Q $2 > $OUT/$1.json 2> $OUT/$1.err; python3 - $OUT/$1.json $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:9.4f} iters {d['iterations_per_step']} parity {d['parity']['ok']}")
except Exception as ex: print(sys.argv[2], 'FAILED', ex)
PY
}
for rep in 1 2; do
run tw8_default_$rep "--config twitter --steps 5 --warmup 2"
run tw8_20pg_$rep "--config twitter --steps 5 --warmup 2 --tune gpush_max_edges=460000"
run yt8_default_$rep "--config youtube --sources 8 --steps 30 --warmup 5"
run yt8_20pg_$rep "--config youtube --sources 8 --steps 30 --warmup 5 --tune gpush_max_edges=4096"
done
run fr10_default "--config friendster --steps 4 --warmup 2"
run fr10_20pg "--config friendster --steps 4 --warmup 2 --tune gpush_max_edges=1000000"
