#!/bin/bash
# The two sweeps of the reference's experiment matrix that round 4 left out (scripts/gpu.sh:112-170): vary_source_features (top10 /
# top1000 / top1000000 source files of the workload tool) and vary_batch_ratios (-r 0.01 / 0.001 / 0.0001), through ./pagerank on the
# youtube and LiveJournal stand-ins (tools/sweep.py; -b 100 as in the scripts). -> gpurun_out/r05_sweeps/r05_sweep_<what>_<key>.jsonl
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r05_sweeps; mkdir -p $OUT
for KEY in youtube livejournal; do
  read FILE DIRECTED <<< $(python3 - $KEY <<'PY'
import sys
sys.path.insert(0, ".")
from dynamicppr_amd import datagen
key = sys.argv[1]
print(datagen.ensure_stand_in(key, "/tmp/dppr_data"), datagen.STAND_INS[key].directed)
PY
)
  for WHAT in source_feature batch_ratio; do
    python3 tools/sweep.py $WHAT --data $FILE --directed $DIRECTED --log-dir $OUT/log_$KEY > $OUT/r05_sweep_${WHAT}_${KEY}.jsonl
    echo "== $WHAT $KEY"; cat $OUT/r05_sweep_${WHAT}_${KEY}.jsonl
  done
  rm -rf $OUT/log_$KEY
done
