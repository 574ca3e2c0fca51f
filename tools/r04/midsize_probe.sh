#!/bin/bash
# The window sizes between a resident launch (<= 262 K vertices with an id) and the binned sweep's default threshold (1 M): which
# sweep form is faster there? A seeded R-MAT stream of scale 21 / 20 M edges (window 2 M edges, ~0.6 M vertices with an id),
# single source: k_pull_iter (default below 1 M ids) against binned sweeps with the threshold lowered to 262 144.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r04_midsize; mkdir -p $OUT
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from dynamicppr_amd import datagen
for scale, edges, seed in ((21, 20_000_000, 7), (20, 10_000_000, 9)):
    V, e1, e2 = datagen.rmat_stream(scale, edges, seed)
    datagen.write_bin(f"/tmp/mid_{scale}.bin", V, e1, e2)
PY
for S in 21 20; do
  for T in "" "--tune binned=1,0,0,0,262144"; do
    N=$( [ -z "$T" ] && echo gather || echo binned )
    python3 bench.py --bin /tmp/mid_$S.bin --directed 1 --sources 1 --no-cpu-baseline --no-merged --steps 20 --warmup 4 $T > $OUT/mid_${S}_$N.json 2> $OUT/mid_${S}_$N.err || echo FAILED
    python3 -c "
import json,sys; d=json.loads(open('$OUT/mid_${S}_$N.json').read().strip().splitlines()[-1]); print('scale $S $N', 'window', d['config']['window'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel'][:34], 'launch_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'parity', d['parity']['ok'])"
  done
done
