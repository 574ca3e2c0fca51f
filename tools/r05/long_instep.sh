#!/bin/bash
# Does a long in-step run keep its speed while the binned tables are patched under FROZEN block cuts (fresh cuts every 32 slides)?
# ./pagerank on a prefix of the twitter stand-in, 70 batches (two re-cuts on the way), ppr time per block of 10 batches from the
# cumulative `ppr_time` lines of the stdout contract; the same with fresh cuts + sorts every epoch (DPPR_BIN_INCREMENTAL=0).
# -> gpurun_out/r05/r05_long_instep_twitter.txt
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r05; mkdir -p $OUT
NB=${1:-70}
read FILE DIRECTED SRC C L WR <<< $(python3 - $NB <<'PY'
import sys
sys.path.insert(0, ".")
from dynamicppr_amd import datagen, stream as st
nb = int(sys.argv[1])
cfg = datagen.STAND_INS["twitter"]
wl = st.workload_config(cfg.edges, 0.1, 0, 0.01, 100)
n = wl.window + (nb + 1) * wl.per_batch + 2          # the window, then nb batches of the stand-in's size
path = datagen.ensure_stand_in("twitter", "/tmp/dppr_data", limit=n)
V, e1, e2 = datagen.read_bin(path)
ratio = (wl.window + 0.5) / len(e1)                  # -w: the window as a fraction of the file
W = int(float(len(e1)) * ratio)
print(path, cfg.directed, int(datagen.top_sources(V, e1, e2, W, cfg.directed, 10)[0]), wl.per_batch, nb * wl.per_batch, repr(ratio))
PY
)
for MODE in patched sorted; do
  unset DPPR_BIN_INCREMENTAL; [ $MODE = sorted ] && export DPPR_BIN_INCREMENTAL=0
  DPPR_NO_OVERLAP=1 DPPR_WATCHDOG_S=120 timeout 900 dynamicppr_amd/host/pagerank -d $FILE -a 0 -i $DIRECTED -y 1 -w $WR -n 1 -c $C -l $L -s $SRC > $OUT/long_$MODE.out 2> $OUT/long_$MODE.err
  python3 - $MODE $OUT/long_$MODE.out <<'PY'
import re, sys
t = [float(x) for x in re.findall(r"^ppr_time (\S+)", open(sys.argv[2]).read(), flags=re.M)]
per = [t[i] - t[i - 1] for i in range(1, len(t))]
blocks = [sum(per[i:i + 10]) / len(per[i:i + 10]) for i in range(0, len(per), 10)]
print(f"{sys.argv[1]:8s} tables: {len(per)} batches, ppr ms per batch in blocks of 10: " + " ".join(f"{b:.1f}" for b in blocks))
PY
done | tee $OUT/r05_long_instep_twitter.txt
