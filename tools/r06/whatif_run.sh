#!/bin/bash
# what bounds the run-based k_bin_reduce: kernel stats of the twitter single-source run under timing-experiment builds
# (build/whatifN.so: -DDPPR_BIN_WHATIF=N, wrong results on purpose) and the hand-written grouping on the headline
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06/whatif
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" != product ]; then export DPPR_LIB=$ROOT/build/$L.so; else unset DPPR_LIB; fi
  rm -rf $OUT/raw_$L
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_$L -- python3 $ROOT/bench.py --config twitter --sources 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings > $OUT/$L.log 2>&1
  f=$(find $OUT/raw_$L -name '*kernel_stats.csv' | head -1)
  echo "== $L"; python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_bin_reduce", "k_bin_scatter", "k_su_grp", "k_su_apply", "k_su_terms", "radix", "merge_impl")):
        print(f"  {n.split('(')[0][:60]:60s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:10.1f}")
PY
  rm -rf $OUT/raw_$L
done
