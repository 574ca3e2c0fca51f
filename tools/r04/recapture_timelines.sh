#!/bin/bash
# Re-capture bench line + kernel stats + batch timeline (ONE profiled run each) after a kernel change that does not alter the
# fabric traffic: tools/r04/recapture_timelines.sh  (the fabric-counter files of tools/r04/capture.sh stay)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CAP=$ROOT/gpurun_out/r04cap; mkdir -p $CAP
one() {
  TAG=$1; shift
  bash $ROOT/tools/prof_timeline.sh r04_$TAG --no-merged --no-extra "$@" > $CAP/timeline_$TAG.log 2>&1 || true
  T=$ROOT/gpurun_out/timeline_r04_$TAG
  cp $T/kernel_stats.csv $CAP/r04_kernel_stats_$TAG.csv; cp $T/timeline.json $CAP/r04_batch_timeline_$TAG.json
  cp $T/timeline.txt $CAP/r04_batch_timeline_$TAG.txt; cp $T/bench.json $CAP/r04_bench_${TAG}_1gpu.json; rm -rf $T
  python3 -c "
import json; d=json.loads(open('$CAP/r04_bench_${TAG}_1gpu.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$TAG', 'ms/step', d['ms_per_step'], 'value', d['value'], 'frac', r['frac'], 'adj', r['frac_group_adjusted'], 'ftraffic', r['frac_traffic'], 'launch_us', r['avg_launch_us'], 'parity', d['parity']['ok'], 'in_region', d['config']['timed_region']['ms_per_step_grouping_in_region'])"
}
one livejournal_group10
one twitter_group --config twitter --steps 6 --warmup 2
one friendster_group --config friendster --steps 4 --warmup 2
one twitter_1src --config twitter --sources 1 --steps 8 --warmup 2
one friendster_1src --config friendster --sources 1 --steps 6 --warmup 2
