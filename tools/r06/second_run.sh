#!/bin/bash
# round 6, second GPU call: the run-based binned tables (tests), A/B of the binned sweep against round 5's library on the twitter
# stand-in, and what the rolling ring costs a configs[2] step (idle / spin diagnostics)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_abi.py tests/test_binned_tables_gpu.py -x -q > $OUT/pytest_binned.log 2>&1
tail -15 $OUT/pytest_binned.log
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "bin" > $OUT/pytest_engine_bin.log 2>&1
tail -5 $OUT/pytest_engine_bin.log
Q="--no-extra --no-cpu-baseline --no-merged --no-extra-passes --no-ceilings --steps 20 --warmup 5"
for v in "ring:" "spin3:--ring-spin-ms 3" "idle20:--ring-idle-ms 20" "prestage:--prestage" "ring2:" "prestage2:--prestage"; do
  n=${v%%:*}; f=${v#*:}
  python bench.py $Q $f > $OUT/diag_$n.json 2> $OUT/diag_$n.err
  python - $OUT/diag_$n.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['event_ms_per_step'], d['event_ms_of_each_step'])
PY
done
bash tools/r05/ab.sh binv2 "--config twitter --sources 1 --steps 8 --warmup 2" build/lib_r05.so product
