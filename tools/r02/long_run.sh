#!/bin/bash
# Long in-step runs of ./pagerank (slide -> update every batch, the reference driver's flow) with and without the
# renumbering of internal ids: mean per-batch latency over -b batches.   tools/r02/long_run.sh [config] [batches] [nsrc]
set -e
cd $GRAFT_REPO_ROOT
CFG=${1:-youtube}; B=${2:-150}; NS=${3:-1}
make -C dynamicppr_amd/host -s all
eval $(python3 - <<PY
import sys
sys.path.insert(0, ".")
from dynamicppr_amd import datagen
cfg = datagen.STAND_INS["$CFG"]
path = datagen.ensure_stand_in("$CFG", "/tmp/dppr_data")
V, e1, e2 = datagen.read_bin(path)
W = int(len(e1) * 0.1)
src = datagen.top_sources(V, e1, e2, W, cfg.directed, 10) if $NS == 1 else datagen.ranked_sources(V, e1, e2, W, cfg.directed, 10, 1000, $NS)
open("/tmp/long_run_sources.txt", "w").write("\n".join(str(int(s)) for s in src[:$NS]) + "\n")
print(f"BIN={path} DIRECTED={cfg.directed}")
PY
)
for rn in ${RN:-1 0}; do
  DPPR_RENUMBER=$rn dynamicppr_amd/host/pagerank -d $BIN -a 0 -i $DIRECTED -y 1 -w 0.1 -n 0 -r 0.01 -b $B --sources /tmp/long_run_sources.txt 2>&1 \
    | grep -E "ppr_latency|ppr_throughput|stream_batch_count" | tail -3 | tr '\n' ' ' | sed "s/^/renumber=$rn $CFG b=$B nsrc=$NS: /"; echo
done
