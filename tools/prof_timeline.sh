#!/bin/bash
# ONE rocprofv3 --kernel-trace --stats run of a bench command, from which BOTH profile artefacts are cut, so that they cannot
# disagree: the per-kernel stats (kernel_stats.csv) and the timeline of one batch (timeline.json / timeline.txt: the batch of
# MEDIAN span among all complete batches of the run -- not the last one, which bench.py event-brackets per launch).
# tools/check_profiles.py verifies the pair. usage: tools/prof_timeline.sh <tag> [bench args...] -> gpurun_out/timeline_<tag>/
set -e
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/timeline_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || true
cp "$(find $OUT/raw -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_full.csv
python3 $ROOT/tools/check_profiles.py --cut "$(find $OUT/raw -name '*kernel_trace.csv' | head -1)" $OUT/kernel_stats_full.csv $OUT
grep "^{" $OUT/bench.log | tail -1 > $OUT/bench.json   # the line of THIS run: profiles/rNN_bench_<tag>_1gpu.json next to the stats file it must agree with
cat $OUT/timeline.txt
rm -rf $OUT/raw $OUT/kernel_stats_full.csv
