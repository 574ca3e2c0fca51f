#!/bin/bash
# HBM traffic of the iteration kernels from PMC counters, collected as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 passes (TCC slots), no trace domains besides --kernel-trace.
# usage: tools/r02/prof_pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/summary.json
set -e
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/raw_$C -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_$C.log 2>&1 || true
  F=$(find $OUT/raw_$C -name "*counter_collection.csv" | head -1)
  cp "$F" $OUT/$C.csv
  rm -rf $OUT/raw_$C
done
python3 $GRAFT_REPO_ROOT/tools/r02/pmc_summary.py $OUT
