#!/bin/bash
# what ONE rank of the 8-GPU configurations' deals holds, measured on one MI355X: S of the configuration's sources on the same window
# (twitter: 8 / 4 + 4 / 2 x 4 / 1 x 8; friendster: 10 / 5 + 5 / 3 + 3 + 2 + 2 / 2 + 2 + 1 x 6). S = 2 on these windows: one after the other
# on the single-source path; S >= 3: one source group.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06/rank_loads; mkdir -p $OUT
cd $ROOT
Q="--no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings --pick top10"
for spec in "twitter 1 6 2" "twitter 2 5 2" "twitter 3 4 2" "twitter 4 4 2" "friendster 1 5 2" "friendster 2 4 2" "friendster 3 4 2" "friendster 5 4 2"; do
  set -- $spec
  P=$([ "$1" = friendster ] && echo top1000 || echo top10)
  python3 bench.py --config $1 --sources $2 --steps $3 --warmup $4 --no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings --pick $P > $OUT/$1_s$2.json 2> $OUT/$1_s$2.err
  python3 - $OUT/$1_s$2.json $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); print(f"{sys.argv[2]:10s} S={sys.argv[3]} ms/step {d['ms_per_step']:9.3f} value {d['value']:14.1f} kernel {d['roofline']['kernel'].split(' (')[0]} parity {d['parity']['ok']} hbm {d['config']['hbm']['hbm_peak_bytes']/1e9:.1f} GB")
except Exception as ex: print(sys.argv[2], sys.argv[3], 'FAILED', ex)
PY
done
