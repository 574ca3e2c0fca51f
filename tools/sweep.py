#!/usr/bin/env python3
"""Experiment sweeps over ./pagerank, after the reference's scripts/gpu.sh + scripts/extract_gpu.py:
vary the batch size (-n 1 -c C -l TOTAL), the variant (-o), or epsilon (-e), log every run's stdout and
scrape the last `ppr_latency` / `ppr_throughput` lines (the stdout contract of gpu/PPRGPU.cuh:170-176).

    tools/sweep.py batch_size --data g.bin --directed 0 --source 1 [--log-dir log]
    tools/sweep.py variant    --data g.bin --directed 0 --source 1        (each -o 0..3 as the engine runs it AND with --push-only: the
                                                                            variants are mechanisms of the push iterations -- gpu/ExpandRev.cuh's
                                                                            four Expand kernels --, which the pull sweeps otherwise replace)
    tools/sweep.py epsilon    --data g.bin --directed 0 --source 1
"""
import argparse
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dynamicppr_amd", "host", "pagerank")
BATCH_SIZES = [1, 10, 100, 1000, 10000, 100000, 1000000]            # scripts/gpu.sh:13
RUN_EDGES = [1000, 100000, 10000, 100000, 1000000, 5000000, 50000000]  # scripts/gpu.sh:14
EPSILONS = ["0.00001", "0.000001", "0.0000001", "0.00000001", "0.000000001", "0.0000000001"]  # scripts/gpu.sh:83


def scrape(text, key):
    """Last occurrence wins, as scripts/extract_gpu.py:18-32 does."""
    vals = re.findall(rf"^{key} (\S+)", text, flags=re.M)
    try:   # (a run without a complete batch prints nan: no value)
        v = float(vals[-1]) if vals else None
    except ValueError:
        return None
    return v if v is not None and v == v and abs(v) != float("inf") else None


def run(args, log_path):
    out = subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    with open(log_path, "w") as f:
        f.write(out)
    return {"ppr_latency_ms": scrape(out, "ppr_latency"), "ppr_throughput": scrape(out, "ppr_throughput")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["batch_size", "variant", "epsilon"])
    ap.add_argument("--data", required=True)
    ap.add_argument("--directed", type=int, required=True)
    ap.add_argument("--source", type=int, required=True)
    ap.add_argument("--log-dir", default="log")
    a = ap.parse_args()
    os.makedirs(a.log_dir, exist_ok=True)
    base = ["-d", a.data, "-a", "0", "-i", str(a.directed), "-y", "1", "-s", str(a.source)]
    name = os.path.basename(a.data)
    rows = []
    if a.what == "batch_size":
        for c, total in zip(BATCH_SIZES, RUN_EDGES):
            r = run(base + ["-n", "1", "-c", str(c), "-l", str(total)],
                    os.path.join(a.log_dir, f"batch_size_{name}_{c}_{a.source}.txt"))
            rows.append({"batch_size": c, **r})
    elif a.what == "variant":
        names = ["OPTIMIZED (eager read, crossing filter)", "FAST_FRONTIER (pre-extracted, crossing filter)",
                 "EAGER (eager read, status-array filter)", "VANILLA (pre-extracted, status-array filter)"]
        for v in range(4):
            for extra in ([], ["--push-only"]):
                r = run(base + ["-n", "0", "-r", "0.01", "-b", "100", "-o", str(v)] + extra,
                        os.path.join(a.log_dir, f"op_gpu_{name}_{v}_{a.source}{'_push_only' if extra else ''}.txt"))
                rows.append({"variant": v, "name": names[v], "push_only": bool(extra), **r})
    else:
        for e in EPSILONS:
            r = run(base + ["-n", "0", "-r", "0.01", "-b", "100", "-e", e],
                    os.path.join(a.log_dir, f"eps_{name}_{e}_{a.source}.txt"))
            rows.append({"epsilon": float(e), **r})
    for row in rows:
        print(json.dumps(row))


if __name__ == "__main__":
    main()
