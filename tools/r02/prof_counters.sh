#!/bin/bash
# Per-dispatch PMC counters of one kernel over a short bench run, one rocprofv3 pass per counter
# group (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md); only
# --kernel-trace beside --pmc.
# usage: tools/r02/prof_counters.sh <tag> <kernel-substring> "<grp1 counters>;<grp2 counters>;..." [bench args...]
#   -> gpurun_out/ctr_<tag>/<group>.csv + table.txt (per dispatch of the LAST batch, launch order)
set -e
TAG=$1; KERN=$2; GROUPS_=$3; shift 3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ctr_$TAG
mkdir -p $OUT
IFS=';' read -ra GRPS <<< "$GROUPS_"
i=0
for G in "${GRPS[@]}"; do
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/raw_$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_$i.log 2>&1 || true
  F=$(find $OUT/raw_$i -name "*counter_collection.csv" | head -1)
  cp "$F" $OUT/group_$i.csv
  rm -rf $OUT/raw_$i
  i=$((i+1))
done
python3 - "$OUT" "$KERN" > $OUT/table.txt <<'PY'
import csv, sys, os, glob, collections
out, kern = sys.argv[1], sys.argv[2]
cols = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, "group_*.csv"))):
    per = collections.defaultdict(dict)
    for row in csv.DictReader(open(f)):
        if kern not in row["Kernel_Name"]:
            continue
        per[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
    ids = sorted(per)
    for name in sorted({n for d in per.values() for n in d}):
        cols[name] = [per[i].get(name, 0.0) for i in ids]
n = min(len(v) for v in cols.values()) if cols else 0
print("dispatches of", kern, ":", n)
names = list(cols)
print("idx " + " ".join(f"{x:>16s}" for x in names))
lo = max(0, n - 130)
for i in range(lo, n):
    print(f"{i:4d} " + " ".join(f"{cols[x][i]:16.1f}" for x in names))
print("mean " + " ".join(f"{sum(cols[x][lo:n]) / max(n - lo, 1):16.1f}" for x in names))
PY
rm -f $OUT/group_*.csv   # (per-dispatch rows of every kernel: tens of MB; the table is what is kept)
tail -140 $OUT/table.txt
