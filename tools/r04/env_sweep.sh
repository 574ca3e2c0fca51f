#!/bin/bash
# One box, one library, several settings of ONE environment variable: tools/r04/env_sweep.sh <tag> VAR "v1 v2 ..." [bench args]
TAG=$1; VAR=$2; VALS=$3; shift 3
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/env_$TAG; mkdir -p $OUT
for V in $VALS; do
  env $VAR=$V python3 bench.py --no-cpu-baseline --no-extra --no-merged "$@" > $OUT/${VAR}_$V.json 2> $OUT/${VAR}_$V.err || echo "FAILED $V"
done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json"), key=os.path.getmtime):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f"{os.path.basename(f):40s} ms/step {d['ms_per_step']:9.4f}  launch_us {r.get('avg_launch_us')}  parity_ok {d.get('parity', {}).get('ok')}")
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
