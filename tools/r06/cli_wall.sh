#!/bin/bash
# ./pagerank's wall time per batch against the ppr_latency it reports, on a prefix of the twitter / friendster
# stand-ins (window and batch size of the stand-in: -w 0.9 of a prefix of W / 0.9 edges, -n 1 -c C): the overlapped loop (default:
# batch k + 1's graph built by a helper thread through dppr_slide_concurrent while batch k is solved), the serial loop with the id
# lookahead of round 4 (DPPR_NO_OVERLAP=1) and the plain serial loop (+ DPPR_NO_LOOKAHEAD=1). stderr line host_times (DPPR_HOST_TIMES=1).
# -> gpurun_out/r06/r06_instep_wall_<key>.jsonl (one line per mode)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06; mkdir -p $OUT
for KEY in ${1:-twitter friendster}; do
  read FILE DIRECTED SRC C L <<< $(python3 - $KEY <<'PY'
import sys
sys.path.insert(0, ".")
from dynamicppr_amd import datagen, stream as st
key = sys.argv[1]
cfg = datagen.STAND_INS[key]
f = cfg.flags.split()
opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
wl = st.workload_config(cfg.edges, 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1)), int(opt.get("-b", 0)), int(opt.get("-c", 0)), int(opt.get("-l", 0)))
n = int(wl.window / 0.9) + 2
assert int(float(n) * 0.9) >= wl.window
path = datagen.ensure_stand_in(key, "/tmp/dppr_data", limit=n)
V, e1, e2 = datagen.read_bin(path)
W = int(float(n) * 0.9)
src = int(datagen.top_sources(V, e1, e2, W, cfg.directed, 10)[0])
print(path, cfg.directed, src, wl.per_batch, min(10 * wl.per_batch, n - W))
PY
)
  rm -f $OUT/r06_instep_wall_$KEY.jsonl
  for MODE in overlap lookahead; do
    unset DPPR_NO_OVERLAP DPPR_NO_LOOKAHEAD DPPR_BIN_INCREMENTAL
    case $MODE in lookahead*|serial) export DPPR_NO_OVERLAP=1;; esac
    case $MODE in *_tables_by_sorts) export DPPR_BIN_INCREMENTAL=0;; esac   # (rounds 3-4: the binned tables by two sorts per epoch instead of patched)
    [ $MODE = serial ] && export DPPR_NO_LOOKAHEAD=1
    DPPR_HOST_TIMES=1 DPPR_WATCHDOG_S=120 timeout 600 dynamicppr_amd/host/pagerank -d $FILE -a 0 -i $DIRECTED -y 1 -w 0.9 -n 1 -c $C -l $L -s $SRC \
      > $OUT/cli_wall_${KEY}_$MODE.out 2> $OUT/cli_wall_${KEY}_$MODE.err
    python3 - $KEY $MODE $OUT/cli_wall_${KEY}_$MODE.out $OUT/cli_wall_${KEY}_$MODE.err >> $OUT/r06_instep_wall_$KEY.jsonl <<'PY'
import json, re, sys
key, mode, out, err = sys.argv[1:5]
o, e = open(out).read(), open(err).read()
lat = [float(x) for x in re.findall(r"^ppr_latency (\S+)", o, flags=re.M)]
m = re.search(r"host_times batches=(\d+) dynamic_ms=(\S+) graph_update_ms=(\S+) ppr_ms=(\S+)(?: graph_update_beside_ppr_ms=(\S+) overlap=(\d))?", e)
row = {"stand_in": key, "mode": mode, "ppr_latency_ms": lat[-1] if lat else None}
if m:
    b = int(m.group(1))
    row.update(batches=b, wall_ms_per_batch=round(float(m.group(2)) / max(b, 1), 3), graph_update_waited_for_ms_per_batch=round(float(m.group(3)) / max(b, 1), 3),
               ppr_host_ms_per_batch=round(float(m.group(4)) / max(b, 1), 3),
               graph_update_beside_ppr_ms_per_batch=round(float(m.group(5)) / max(b, 1), 3) if m.group(5) else None)
    if row["ppr_latency_ms"]:
        row["wall_over_ppr_latency"] = round(row["wall_ms_per_batch"] / row["ppr_latency_ms"], 4)
print(json.dumps(row))
PY
    tail -1 $OUT/r06_instep_wall_$KEY.jsonl
  done
  unset DPPR_NO_OVERLAP DPPR_NO_LOOKAHEAD DPPR_BIN_INCREMENTAL
done
