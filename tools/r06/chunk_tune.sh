#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06/chunk; mkdir -p $OUT; cd $ROOT
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings"
run() { python3 bench.py $Q $2 > $OUT/$1.json 2> $OUT/$1.err; python3 - $OUT/$1.json $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:8.4f} event {d['event_ms_per_step']:8.4f} parity {d['parity']['ok']}")
except Exception as ex: print(sys.argv[2], 'FAILED', ex)
PY
}
for rep in 1 2; do
run default_$rep ""
run chunk16_$rep "--tune chunk_iters=16"
run chunk32_$rep "--tune chunk_iters=32"
run chunk48_$rep "--tune chunk_iters=48"
run chunk64_$rep "--tune chunk_iters=64"
done
