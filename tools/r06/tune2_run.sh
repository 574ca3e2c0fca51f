#!/bin/bash
# B-block target against the number of co-resident workgroups of k_bin_reduce (2 x 256): about one heavy block per slot?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06/tune2; mkdir -p $OUT; cd $ROOT
run() { # name config steps tune
  python3 bench.py --config $2 --sources 1 --steps $3 --warmup 2 --no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings $4 > $OUT/$1.json 2> $OUT/$1.err
  python3 - $OUT/$1.json $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); print(f"{sys.argv[2]:28s} ms/step {d['ms_per_step']:8.3f} sweep_us {d['roofline']['avg_launch_us']:8.1f} parity {d['parity']['ok']}")
except Exception as ex: print(sys.argv[2], 'FAILED', ex)
PY
}
run fr_384k friendster 5 ""
run fr_560k friendster 5 "--tune binned=1,0,0,573440,0,0,0"
run fr_752k friendster 5 "--tune binned=1,0,0,770048,0,0,0"
run fr_1m friendster 5 "--tune binned=1,0,0,1048576,0,0,0"
run fr_384k_b friendster 5 ""
run tw_384k twitter 6 ""
run tw_306k twitter 6 "--tune binned=1,0,0,306000,0,0,0"
run tw_480k twitter 6 "--tune binned=1,0,0,491520,0,0,0"
run tw_384k_b twitter 6 ""
