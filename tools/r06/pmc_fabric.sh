#!/bin/bash
# Fabric traffic of the iteration kernels from PMC counters, by REQUEST SIZE (rounds 4-5): the L2's memory-side read requests
# come in 32 / 64 / 128 bytes (TCC_EA0_RDREQ_32B / _64B / _128B; FETCH_SIZE tallies the 128-byte ones at 64, which is the
# "x2 correction" of /opt/skills/guides/MI355X_MICROARCH.md -- counting by size needs no correction), write requests in 64 or
# 32 bytes (TCC_EA0_WRREQ_64B of TCC_EA0_WRREQ). One rocprofv3 pass per counter group, only --kernel-trace beside --pmc.
# These are L2 <-> fabric bytes: Infinity-Cache hits are in them (the guide: "appear to be counted").
# usage: tools/r06/pmc_fabric.sh <tag> [bench args...]  ->  gpurun_out/pmcf_<tag>/summary.json (per kernel: launches, requests, bytes)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmcf_$TAG
mkdir -p $OUT
i=0
for G in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/raw_$i -- python3 $ROOT/bench.py --no-cpu-baseline --no-merged --no-extra "$@" > $OUT/bench_$i.log 2>&1 || true
  F=$(find $OUT/raw_$i -name "*counter_collection.csv" | head -1)
  cp "$F" $OUT/group_$i.csv
  rm -rf $OUT/raw_$i
  i=$((i+1))
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(out, "group_*.csv"))):
    for row in csv.DictReader(open(f)):
        short = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("dppr::", "")
        if not short.startswith("k_"):
            continue
        a = acc[short][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
res = {}
for k, c in acc.items():
    m = {n: v[0] / v[1] for n, v in c.items() if v[1]}
    if "TCC_EA0_RDREQ_sum" not in m or "TCC_EA0_WRREQ_sum" not in m:
        continue
    rd = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
    wr = 64 * m.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (m["TCC_EA0_WRREQ_sum"] - m.get("TCC_EA0_WRREQ_64B_sum", 0))
    res[k] = {"launches": c["TCC_EA0_RDREQ_sum"][1],
              "read_requests_per_launch": {"32B": m.get("TCC_EA0_RDREQ_32B_sum", 0), "64B": m.get("TCC_EA0_RDREQ_64B_sum", 0), "128B": m.get("TCC_EA0_RDREQ_128B_sum", 0)},
              "write_requests_per_launch": {"64B": m.get("TCC_EA0_WRREQ_64B_sum", 0), "32B": m["TCC_EA0_WRREQ_sum"] - m.get("TCC_EA0_WRREQ_64B_sum", 0)},
              "fabric_read_bytes_per_launch": rd, "fabric_write_bytes_per_launch": wr, "fabric_bytes_per_launch": rd + wr,
              "l2_hit_rate": (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])) if m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0) > 0 else None,
              "fetch_size_raw_bytes_per_launch": 1024 * m["FETCH_SIZE"] if "FETCH_SIZE" in m else None,
              "write_size_bytes_per_launch": 1024 * m["WRITE_SIZE"] if "WRITE_SIZE" in m else None,
              # the method of rounds 1-3 (2 x FETCH_SIZE + WRITE_SIZE), kept for comparison with their files
              "hbm_bytes_per_launch_corrected": 1024 * (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) if "FETCH_SIZE" in m and "WRITE_SIZE" in m else None}
# round 6: the summary says which build of the library ran under the counters (bench.py refuses traffic from another build)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from dynamicppr_amd import engine as _eng
res["_stamp"] = {"build_id": _eng.build_id(), "git_commit": os.environ.get("GIT_COMMIT"), "captured_by": "tools/r06/pmc_fabric.sh"}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
res.pop("_stamp")
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["fabric_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k:34s} launches {v['launches']:6d}  fabric {v['fabric_bytes_per_launch'] / 1e6:9.2f} MB/launch (read {v['fabric_read_bytes_per_launch'] / 1e6:8.2f}, write {v['fabric_write_bytes_per_launch'] / 1e6:8.2f})  L2 hit {v['l2_hit_rate']}")
PY
rm -f $OUT/group_*.csv
