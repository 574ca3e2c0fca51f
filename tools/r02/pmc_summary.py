#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 --pmc CSVs (FETCH_SIZE.csv, WRITE_SIZE.csv).

Units and gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
counter values are KiB; FETCH_SIZE reports exactly half of the bytes of wide coalesced
streaming reads on gfx950 (64-B tally of 128-B requests), WRITE_SIZE reads exactly. Both the
raw and the corrected (2 x FETCH) figures are reported; 8-byte gathers are 'uncalibrated
widths' in the guide, so the corrected number is an upper bound for the gather share."""
import csv
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(d, c + ".csv")
    if not os.path.exists(path):
        continue
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")
        short = name.split("(")[0].replace("void ", "").replace("dppr::", "")
        if not short.startswith("k_"):
            continue
        if row.get("Counter_Name") != c:
            continue
        a = acc[short][c]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
out = {}
for k, v in acc.items():
    f, nf = v["FETCH_SIZE"]
    w, nw = v["WRITE_SIZE"]
    if not nf or not nw:
        continue
    out[k] = {
        "launches": nf,
        "fetch_KiB_per_launch_raw": f / nf,
        "write_KiB_per_launch": w / nw,
        "hbm_bytes_per_launch_raw": (f / nf + w / nw) * 1024,
        "hbm_bytes_per_launch_corrected": (2 * f / nf + w / nw) * 1024,
    }
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"] * kv[1]["launches"]):
    print(f"{k:22s} launches={v['launches']:6d} fetch_raw={v['fetch_KiB_per_launch_raw']:10.1f} KiB "
          f"write={v['write_KiB_per_launch']:10.1f} KiB corrected={v['hbm_bytes_per_launch_corrected'] / 1e6:8.3f} MB/launch")
