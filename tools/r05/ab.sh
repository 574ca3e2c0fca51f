#!/bin/bash
# A/B of library builds on ONE box (round 5 form of tools/r04/ab.sh): tools/r05/ab.sh <tag> "<bench args>" lib1 lib2 ...
# (lib = path relative to the repo, or "product"; a lib may carry environment settings: "build/x.so:DPPR_GSWEEP_HOT=0").
# Each library runs the same bench command twice, interleaved (A B A B), so that run-to-run spread is visible.
TAG=$1; ARGS=$2; shift 2
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_$TAG; mkdir -p $OUT
for rep in 1 2; do
  for SPEC in "$@"; do
    L=${SPEC%%:*}; ENVS=""; [ "$L" != "$SPEC" ] && ENVS=${SPEC#*:}
    N=$(basename $L .so)$( [ -n "$ENVS" ] && echo "_$(echo $ENVS | tr '= ' '__')" )
    ( if [ "$L" != product ]; then export DPPR_LIB=$GRAFT_REPO_ROOT/$L; fi
      for kv in $(echo $ENVS | tr ',' ' '); do export $kv; done
      python3 bench.py --no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings $ARGS > $OUT/${N}_$rep.json 2> $OUT/${N}_$rep.err || echo "FAILED $N" )
  done
done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f"{os.path.basename(f):48s} ms/step {d['ms_per_step']:9.4f}  value {d['value']:14.1f}  launch_us {r.get('avg_launch_us')}  parity_ok {d.get('parity', {}).get('ok')}")
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
