#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_binned_tables_gpu.py tests/test_engine_gpu.py -x -q -k "bin or incremental_batch_update or seed" > $OUT/pytest_5.log 2>&1
tail -4 $OUT/pytest_5.log
bash tools/prof_timeline.sh r06_lj10 --steps 20 --warmup 5 --no-extra --no-merged --no-extra-passes --no-ceilings > $OUT/lj10_timeline.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' $ROOT/gpurun_out/timeline_r06_lj10/bench.json | head -1
head -24 $ROOT/gpurun_out/timeline_r06_lj10/timeline.txt
bash tools/r05/ab.sh binv2b "--config twitter --sources 1 --steps 8 --warmup 2" build/lib_r05.so product
bash tools/r05/ab.sh binv2lj "--config livejournal --sources 1 --steps 20 --warmup 5" build/lib_r05.so product
bash tools/r05/ab.sh binv2fr "--config friendster --sources 1 --steps 6 --warmup 2" build/lib_r05.so product
