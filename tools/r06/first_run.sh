#!/bin/bash
# round 6, first GPU call: the rolling-ring bench.py (default line with the configs[3] block, --prestage A/B, friendster and a 2-rank
# twitter run at the driver's step counts) and the tests that changed
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_bench_gpu.py tests/test_overlap_gpu.py -x -q > $OUT/pytest_first.log 2>&1
tail -5 $OUT/pytest_first.log
( time python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
( time python bench.py --steps 20 --warmup 5 --prestage --no-extra --no-cpu-baseline > $OUT/bench_default_prestage.json 2> $OUT/bench_default_prestage.err ) 2>&1 | grep real
( time python bench.py --config friendster --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_friendster_s20.json 2> $OUT/bench_friendster_s20.err ) 2>&1 | grep real
( time python bench.py --gpus 2 --config twitter --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_twitter_2ranks_s20.json 2> $OUT/bench_twitter_2ranks_s20.err ) 2>&1 | grep real
for f in bench_default bench_default_prestage bench_friendster_s20 bench_twitter_2ranks_s20; do
  python - $OUT/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['config']['hbm']['hbm_peak_bytes']/1e9, d['parity']['ok'], d['parity'].get('sources_compared'), d.get('wall_ms_per_step_incl_graph_update'), (d.get('configs3_strong') or {}).get('ms_per_step'))
except Exception as ex:
    print(sys.argv[1], 'NO LINE', ex)
PY
  tail -3 $OUT/$f.err
done
