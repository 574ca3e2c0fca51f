#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command, plus -- from the per-dispatch trace of the same run --
# the dominant kernel's average over the launches that found a frontier (bench.py's roofline.avg_launch_us brackets
# exactly those with hipEvents; the plain --stats average also counts the empty launches at the end of a chunk).
# usage: tools/r02/prof_agree.sh <tag> [bench args...]  -> gpurun_out/agree_<tag>/{kernel_stats.csv,agreement.txt,bench.log}
set -e
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/agree_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || true
cp "$(find $OUT/raw -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
python3 - "$(find $OUT/raw -name '*kernel_trace.csv' | head -1)" $OUT/bench.log > $OUT/agreement.txt <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_gsweep" in r["Kernel_Name"] or "k_pull_resident" in r["Kernel_Name"] or "k_pull_iter" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
busy = [x for x in d if x > 10.0]
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(f"dominant kernel, rocprofv3 kernel trace: {len(d)} launches, average {sum(d)/len(d):.1f} us; "
      f"{len(busy)} of them ran longer than 10 us (found a frontier), average {sum(busy)/len(busy):.1f} us")
print(f"bench.py (same run, under the profiler): roofline.avg_launch_us {line['roofline']['avg_launch_us']} over {line['roofline']['launches']} event-bracketed launches; ms_per_step {line['ms_per_step']}")
PY
cat $OUT/agreement.txt
rm -rf $OUT/raw
