#!/bin/bash
# round 6, fourth GPU call: pipelined binned passes + hand-written grouping: tests, kernel traces (twitter single source, headline)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_binned_tables_gpu.py tests/test_engine_gpu.py -x -q -k "bin or incremental_batch_update or seed or group" > $OUT/pytest_4.log 2>&1
tail -8 $OUT/pytest_4.log
bash tools/prof_timeline.sh r06_tw1 --config twitter --sources 1 --steps 6 --warmup 2 --no-extra --no-merged --no-extra-passes --no-ceilings > $OUT/tw1_timeline.txt 2>&1
head -12 $ROOT/gpurun_out/timeline_r06_tw1/kernel_stats.csv
grep -o '"ms_per_step": [0-9.]*' $ROOT/gpurun_out/timeline_r06_tw1/bench.json | head -1
bash tools/prof_timeline.sh r06_lj10 --steps 20 --warmup 5 --no-extra --no-merged --no-extra-passes --no-ceilings > $OUT/lj10_timeline.txt 2>&1
head -14 $ROOT/gpurun_out/timeline_r06_lj10/kernel_stats.csv
grep -o '"ms_per_step": [0-9.]*' $ROOT/gpurun_out/timeline_r06_lj10/bench.json | head -1
head -40 $ROOT/gpurun_out/timeline_r06_lj10/timeline.txt
