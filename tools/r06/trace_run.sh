#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT; cd $ROOT
Q="--sources 1 --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings"
DPPR_LOOP_TRACE=1 python3 bench.py --config twitter $Q > $OUT/trace_tw1.json 2> $OUT/trace_tw1.err
grep "^\[loop" $OUT/trace_tw1.err | head -230 | tail -115
for v in "fr_default:" "fr_t192k:--tune binned=1,0,0,196608,0,0,0" "fr_default2:" "fr_t192k2:--tune binned=1,0,0,196608,0,0,0"; do
  n=${v%%:*}; f=${v#*:}
  python3 bench.py --config friendster $Q $f > $OUT/$n.json 2> $OUT/$n.err
  python3 - $OUT/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); print(f"{sys.argv[2]:28s} ms/step {d['ms_per_step']:8.3f} sweep_us {d['roofline']['avg_launch_us']:8.1f} parity {d['parity']['ok']}")
except Exception as ex: print(sys.argv[2], 'FAILED', ex)
PY
done
