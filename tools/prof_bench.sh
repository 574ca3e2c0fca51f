#!/bin/bash
# rocprofv3 kernel-trace + stats of a short bench run; summaries land in gpurun_out/prof_<tag>/
# usage: tools/prof_bench.sh <tag> [bench args...]
set -e
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || true
tail -2 $OUT/bench.log
F=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
cp "$F" $OUT/kernel_stats.csv 2>/dev/null || true
head -25 $OUT/kernel_stats.csv
rm -rf $OUT/raw
