#!/usr/bin/env python3
"""Where the idle time of a batch sits: tools/r04/gaps.py <kernel_trace.csv> -- for the batch of median span (cut as
tools/check_profiles.py cuts batches: from an IncrementalBatchUpdate kernel to the next), every gap > 5 us between the end of a
dispatch and the start of the next, with the two kernels around it."""
import csv
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from check_profiles import short

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("k_su_terms")]
batches = [(rows[marks[k]][0], rows[marks[k + 1] - 1][1], marks[k], marks[k + 1]) for k in range(len(marks) - 1)]
# a batch ends where the slide of the next begins: drop the slide's kernels at its tail
def trim(a, b):
    while b > a and rows[b - 1][2].startswith(("k_make", "k_build", "rocprim", "k_deg", "k_keep", "k_tile", "k_compact", "k_su_keys", "k_gather_deg", "k_gtab", "k_cut", "k_bin", "k_hub", "k_assign", "k_res", "k_ginit", "k_grp")):
        b -= 1
    return b
spans = []
for s, e, a, b in batches:
    b = trim(a, b)
    spans.append((rows[b - 1][1] - rows[a][0], a, b))
spans.sort()
span, a, b = spans[len(spans) // 2]
print(f"{len(spans)} batches; median span {span / 1e3:.1f} us, {b - a} dispatches")
busy = sum(rows[i][1] - rows[i][0] for i in range(a, b))
print(f"busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
tot = 0
for i in range(a + 1, b):
    gap = rows[i][0] - rows[i - 1][1]
    if gap > 5000:
        tot += gap
        print(f"  +{(rows[i][0] - rows[a][0]) / 1e3:9.1f} us  gap {gap / 1e3:6.1f} us  after {rows[i - 1][2][:40]:40s} before {rows[i][2][:40]}")
print(f"gaps > 5 us: {tot / 1e3:.1f} us")
