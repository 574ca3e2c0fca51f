#!/bin/bash
# The reference's experiment matrix (scripts/gpu.sh:13-14 batch size, :83 epsilon) through ./pagerank on the youtube and LiveJournal
# stand-ins (tools/sweep.py), plus the timed region of a c = 1 and a c = 10 batch cut out of a profiled bench run (dispatches per
# batch, floor in microseconds). -> gpurun_out/r04_sweeps/
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r04_sweeps; mkdir -p $OUT
for KEY in youtube livejournal; do
  read FILE DIRECTED SRC <<< $(python3 - $KEY <<'PY'
import sys
sys.path.insert(0, ".")
from dynamicppr_amd import datagen, stream as st
key = sys.argv[1]
cfg = datagen.STAND_INS[key]
path = datagen.ensure_stand_in(key, "/tmp/dppr_data")
V, e1, e2 = datagen.read_bin(path)
W = st.workload_config(cfg.edges, 0.1, 0, 0.01, 100).window
print(path, cfg.directed, int(datagen.top_sources(V, e1, e2, W, cfg.directed, 10)[3]))   # scripts/gpu.sh takes index 3 of the top10 file
PY
)
  for WHAT in batch_size epsilon; do
    python3 tools/sweep.py $WHAT --data $FILE --directed $DIRECTED --source $SRC --log-dir $OUT/log > $OUT/r04_sweep_${WHAT}_${KEY}.jsonl
    echo "== $WHAT $KEY"; cat $OUT/r04_sweep_${WHAT}_${KEY}.jsonl
  done
done
rm -rf $OUT/log
for C in 1 10; do
  bash tools/prof_timeline.sh r04_youtube_c$C --config youtube --batch-edges $C --steps 200 --warmup 20 --no-merged > /dev/null 2>&1 || true
  T=$ROOT/gpurun_out/timeline_r04_youtube_c$C
  cp $T/timeline.txt $OUT/r04_batch_timeline_youtube_c$C.txt; cp $T/timeline.json $OUT/r04_batch_timeline_youtube_c$C.json; cp $T/kernel_stats.csv $OUT/r04_kernel_stats_youtube_c$C.csv
  cp $T/bench.json $OUT/r04_bench_youtube_c${C}_1gpu.json
  head -6 $T/timeline.txt | cut -c1-400
  rm -rf $T
done
