#!/usr/bin/env python3
"""Calibration probe: returning f64 atomic-add rate at random addresses,
by table size and memory scope. Writes gpurun_out/atomics_probe.json."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicppr_amd import engine as eng

out = []
n = 1 << 24
for scope in (0, 1, 2, 3):
    for log2 in (14, 17, 20, 23, 26):
        ms = eng.bench_atomics(1 << log2, n, scope, reps=5)
        rate = n / ms / 1e6  # G atomics/s
        out.append(dict(scope=("agent", "workgroup", "agent, result unused", "plain 8-byte stores")[scope], table_bytes=8 << log2, n=n, ms=ms,
                        gatomics_per_s=rate))
        print(out[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/atomics_probe.json", "w"), indent=1)
