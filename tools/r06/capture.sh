#!/bin/bash
# Round-6 profile artefacts of one workload, all from runs of ONE build on ONE box (the round-4 recipe + stamps):
#   1. tools/r06/pmc_fabric.sh      -> profiles/r06_pmc_fabric_<pmctag>.json, stamped with the library's build id (copied into
#                                      profiles/ on the box FIRST, so that the bench line of step 2 carries `traffic` from this very build)
#   2. tools/prof_timeline.sh       -> kernel stats + batch timeline + the bench line of the SAME profiled run
# usage: GIT_COMMIT=<sha> tools/r06/capture.sh <tag> <pmctag> [bench args...]   -> gpurun_out/r06cap/  (copy into profiles/ afterwards,
#        then tools/r06/manifest.py writes profiles/r06_manifest.json)
set -e
TAG=$1; PMCTAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CAP=$ROOT/gpurun_out/r06cap; mkdir -p $CAP
bash $ROOT/tools/r06/pmc_fabric.sh $TAG --no-extra-passes --no-ceilings "$@" > $CAP/pmc_$TAG.log 2>&1 || true
if [ -s $ROOT/gpurun_out/pmcf_$TAG/summary.json ]; then
  cp $ROOT/gpurun_out/pmcf_$TAG/summary.json $CAP/r06_pmc_fabric_$PMCTAG.json
  cp $ROOT/gpurun_out/pmcf_$TAG/summary.json $ROOT/profiles/r06_pmc_fabric_$PMCTAG.json
fi
rm -rf $ROOT/gpurun_out/pmcf_$TAG
bash $ROOT/tools/prof_timeline.sh r06_$TAG --no-merged --no-extra --no-extra-passes "$@" > $CAP/timeline_$TAG.log 2>&1 || true
T=$ROOT/gpurun_out/timeline_r06_$TAG
cp $T/kernel_stats.csv $CAP/r06_kernel_stats_$TAG.csv
cp $T/timeline.json $CAP/r06_batch_timeline_$TAG.json
cp $T/timeline.txt $CAP/r06_batch_timeline_$TAG.txt
cp $T/bench.json $CAP/r06_bench_${TAG}_1gpu.json
rm -rf $T
python3 - $CAP/r06_bench_${TAG}_1gpu.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1].split("/")[-1], "ms/step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "launch_us", r["avg_launch_us"], "traffic", r.get("traffic"), "frac_traffic", r.get("frac_traffic"), "parity", d["parity"]["ok"], "build", d.get("build_id"))
PY
