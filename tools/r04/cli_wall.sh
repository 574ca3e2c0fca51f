#!/bin/bash
# ./pagerank's wall time per batch against the ppr_latency it reports, on a prefix of the twitter / friendster
# stand-ins (window and batch size of the stand-in: -w 0.9 of a prefix of W / 0.9 edges, -n 1 -c C), with the id lookahead
# (dppr_hint_next_batch, default) and without (DPPR_NO_LOOKAHEAD=1). stderr line host_times (DPPR_HOST_TIMES=1).
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r04; mkdir -p $OUT
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
  for LOOK in 1 0; do
    if [ $LOOK = 0 ]; then export DPPR_NO_LOOKAHEAD=1; else unset DPPR_NO_LOOKAHEAD; fi
    DPPR_HOST_TIMES=1 DPPR_WATCHDOG_S=120 timeout 600 dynamicppr_amd/host/pagerank -d $FILE -a 0 -i $DIRECTED -y 1 -w 0.9 -n 1 -c $C -l $L -s $SRC \
      > $OUT/cli_wall_${KEY}_look$LOOK.out 2> $OUT/cli_wall_${KEY}_look$LOOK.err
    echo "== $KEY lookahead=$LOOK: $(grep -E '^ppr_latency' $OUT/cli_wall_${KEY}_look$LOOK.out | tail -1) | $(grep host_times $OUT/cli_wall_${KEY}_look$LOOK.err)"
  done
  unset DPPR_NO_LOOKAHEAD
done 2>&1 | tee $OUT/r04_cli_wall.txt
