#!/bin/bash
# kernel timeline of a short bench run -> gpurun_out/trace_<tag>/kernel_trace.csv (start/end per dispatch)
set -e
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || true
F=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python3 - "$F" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last k_su_keys -> analyse that update (one batch)
idx = [i for i, r in enumerate(rows) if "k_su_keys" in r["Kernel_Name"]]
lo = idx[-2]; hi = idx[-1]
seg = rows[lo:hi]
t0 = int(seg[0]["Start_Timestamp"])
busy = 0; prev_end = None; gaps = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if prev_end is not None: gaps.append((s - prev_end, r["Kernel_Name"].split("(")[0][-28:]))
    prev_end = e
span = int(seg[-1]["End_Timestamp"]) - t0
print("dispatches", len(seg), "span_us", span / 1e3, "busy_us", busy / 1e3, "gap_us", (span - busy) / 1e3)
big = sorted(gaps, reverse=True)[:25]
print("largest gaps (us, before kernel):", [(round(g / 1e3, 1), n) for g, n in big])
import collections
c = collections.Counter(); d = collections.Counter()
for r in seg:
    n = r["Kernel_Name"].split("(")[0][-30:]; c[n] += 1; d[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, k in c.most_common(): print(f"{n:32s} x{k:4d} total_us={d[n]/1e3:9.1f}")
sweeps = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in seg
          if "k_gsweep" in r["Kernel_Name"] or "k_pull_iter" in r["Kernel_Name"]]
if sweeps: print("sweep durations in launch order (us):", sweeps)
small = [g for g, _ in gaps if g < 20000]
print("median gap us", sorted(small)[len(small)//2] / 1e3, "n small", len(small), "sum small", sum(small) / 1e3)
PY
cat $OUT/timeline.txt
rm -rf $OUT/raw
