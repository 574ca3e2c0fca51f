#!/bin/bash
# tuning sweep of the run-based binned sweep on the twitter stand-in (single source): block shapes (runtime knobs) and BIN_U (builds)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06/tune
mkdir -p $OUT
cd $ROOT
Q="--config twitter --sources 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings"
run() { # name lib tune
  if [ "$2" != product ]; then export DPPR_LIB=$ROOT/build/$2.so; else unset DPPR_LIB; fi
  python3 bench.py $Q $3 > $OUT/$1.json 2> $OUT/$1.err
  python3 - $OUT/$1.json $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); print(f"{sys.argv[2]:28s} ms/step {d['ms_per_step']:8.3f} sweep_us {d['roofline']['avg_launch_us']:8.1f} parity {d['parity']['ok']}")
except Exception as ex: print(sys.argv[2], 'FAILED', ex)
PY
}
run base product ""
run hb48_t384k product "--tune binned=1,0,48,393216,0,0,0"
run hb60_t384k product "--tune binned=1,0,60,393216,0,0,0"
run hb60_t768k product "--tune binned=1,0,60,786432,0,0,0"
run hb60_t1536k product "--tune binned=1,0,60,1572864,0,0,0"
run hb120_t768k product "--tune binned=1,0,120,786432,0,0,0"
run hb120_t1536k product "--tune binned=1,0,120,1572864,0,0,0"
run hb60_t768k_c64k product "--tune binned=1,0,60,786432,0,65536,0"
run base2 product ""
