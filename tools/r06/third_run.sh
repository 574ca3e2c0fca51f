#!/bin/bash
# round 6, third GPU call: binned tests after the scan-scratch fix; kernel trace of the twitter single-source run (new tables)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_abi.py tests/test_binned_tables_gpu.py -x -q > $OUT/pytest_binned.log 2>&1
tail -15 $OUT/pytest_binned.log
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "bin" > $OUT/pytest_engine_bin.log 2>&1
tail -5 $OUT/pytest_engine_bin.log
bash tools/prof_timeline.sh r06_tw1 --config twitter --sources 1 --steps 6 --warmup 2 --no-extra --no-merged --no-extra-passes --no-ceilings > $OUT/tw1_timeline.txt 2>&1
head -30 $ROOT/gpurun_out/timeline_r06_tw1/kernel_stats.csv
