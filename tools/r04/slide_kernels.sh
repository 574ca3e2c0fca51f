#!/bin/bash
# Per-kernel cost of the slides of a twitter / friendster-size stand-in: one rocprofv3 --kernel-trace --stats run of the in-step probe
# (tools/r04/slide_costs.py --child <key> incremental binned renumber lookahead), kernels by total time. usage: tools/r04/slide_kernels.sh <key>
KEY=${1:-friendster}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $ROOT/tools/r04/slide_costs.py --child $KEY 1 1 0 1 > /tmp/c.log 2>&1
python3 - "$ROOT" <<'PY'
import csv, glob, sys
sys.path.insert(0, sys.argv[1] + "/tools")
from check_profiles import short
f = glob.glob("/tmp/st/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:45]:
    print("%-52s calls %6s total_ms %9.2f avg_us %10.1f" % (short(r["Name"])[:52], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
