#!/usr/bin/env python3
"""Experiment sweeps over ./pagerank, after the reference's scripts/gpu.sh + scripts/extract_gpu.py:
vary the batch size (-n 1 -c C -l TOTAL), the variant (-o), or epsilon (-e), log every run's stdout and
scrape the last `ppr_latency` / `ppr_throughput` lines (the stdout contract of gpu/PPRGPU.cuh:170-176).

    tools/sweep.py batch_size --data g.bin --directed 0 --source 1 [--log-dir log]
    tools/sweep.py variant    --data g.bin --directed 0 --source 1        (each -o 0..3 as the engine runs it AND with --push-only: the
                                                                            variants are mechanisms of the push iterations -- gpu/ExpandRev.cuh's
                                                                            four Expand kernels --, which the pull sweeps otherwise replace)
    tools/sweep.py epsilon    --data g.bin --directed 0 --source 1
    tools/sweep.py source_feature --data g.bin --directed 0      (scripts/gpu.sh:112-140 vary_source_features: sources from the workload tool's
                                                                   <data>_top10.txt / _top1000.txt / _top1000000.txt files -- degree ranks [0,10),
                                                                   [10,1000), [1000,1000000), workload/Workload.cpp:45-55; written next to the
                                                                   logs by dynamicppr_amd/tools.py workload when --source-dir has none)
    tools/sweep.py batch_ratio    --data g.bin --directed 0      (scripts/gpu.sh:144-170 vary_batch_ratios: -r 0.01 / 0.001 / 0.0001 x the three
                                                                   source features, -b 100)
The reference runs entries SOURCES_START .. SOURCES_END of each file (scripts/gpu.sh:17-18: index 3); --source-index / --sources-per-feature.
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
SOURCE_FEATURES = ["top10", "top1000", "top1000000"]   # scripts/gpu.sh:113
BATCH_RATIOS = ["0.01", "0.001", "0.0001"]             # scripts/gpu.sh:146


def feature_sources(data, directed, source_dir, index, count):
    """{feature: [ids]} from the workload tool's files (whole-file degree ranks, out-degree: `workload <bin> <directed> 0 1`,
    the files scripts/gpu.sh reads from exp_vids/); files that do not exist yet are written by the in-repo workload tool."""
    import sys
    sys.path.insert(0, ROOT)
    from dynamicppr_amd import tools as dtools
    base = os.path.basename(data)
    paths = {f: os.path.join(source_dir, f"{base}_{f}.txt") for f in SOURCE_FEATURES}
    if not all(os.path.exists(p) for p in paths.values()):
        os.makedirs(source_dir, exist_ok=True)
        dtools.workload(data, directed, 0, 1, out_dir=source_dir)
    out = {}
    for f, p in paths.items():
        if not os.path.exists(p):
            continue    # (a graph with fewer vertices than the rank range begins at has no such file, like the reference's tool)
        ids = [int(x) for x in open(p).read().split()]
        out[f] = [ids[(index + k) % len(ids)] for k in range(min(count, len(ids)))]
    return out


def scrape(text, key):
    """Last occurrence wins, as scripts/extract_gpu.py:18-32 does."""
    vals = re.findall(rf"^{key} (\S+)", text, flags=re.M)
    try:   # (a run without a complete batch prints nan: no value)
        v = float(vals[-1]) if vals else None
    except ValueError:
        return None
    return v if v is not None and v == v and abs(v) != float("inf") else None


OVERLAP = False   # --overlap: ./pagerank's default loop (the graph of batch k + 1 built beside the solve of batch k)


def run(args, log_path):
    # The sweeps reproduce the reference's experiments, whose metric is the TIMED region alone (gpu/PPRGPU.cuh:138-164): by default
    # they run ./pagerank's serial loop (DPPR_NO_OVERLAP=1), in which nothing shares the device with that region. With --overlap the
    # graph build of the next batch runs beside it: shorter wall time per batch, longer ppr_latency.
    env = dict(os.environ) if OVERLAP else dict(os.environ, DPPR_NO_OVERLAP="1")
    out = subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env).stdout
    with open(log_path, "w") as f:
        f.write(out)
    return {"ppr_latency_ms": scrape(out, "ppr_latency"), "ppr_throughput": scrape(out, "ppr_throughput")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["batch_size", "variant", "epsilon", "source_feature", "batch_ratio"])
    ap.add_argument("--data", required=True)
    ap.add_argument("--directed", type=int, required=True)
    ap.add_argument("--source", type=int, default=None, help="source vertex (batch_size / variant / epsilon)")
    ap.add_argument("--source-dir", default=None, help="where the <data>_top10.txt ... files are (default: the log directory)")
    ap.add_argument("--source-index", type=int, default=3, help="first entry of a source file to run (scripts/gpu.sh:17 SOURCES_START)")
    ap.add_argument("--sources-per-feature", type=int, default=1)
    ap.add_argument("--batches", type=int, default=100, help="-b of the ratio-configured runs (scripts/gpu.sh: 100)")
    ap.add_argument("--log-dir", default="log")
    ap.add_argument("--overlap", action="store_true", help="leave ./pagerank's overlapped loop on (default: the serial loop, whose ppr_latency is the timed region undisturbed)")
    a = ap.parse_args()
    global OVERLAP
    OVERLAP = a.overlap
    os.makedirs(a.log_dir, exist_ok=True)
    if a.what in ("batch_size", "variant", "epsilon") and a.source is None:
        ap.error(f"{a.what} needs --source")
    base = ["-d", a.data, "-a", "0", "-i", str(a.directed), "-y", "1"] + (["-s", str(a.source)] if a.source is not None else [])
    name = os.path.basename(a.data)
    rows = []
    if a.what in ("source_feature", "batch_ratio"):
        feats = feature_sources(a.data, a.directed, a.source_dir or a.log_dir, a.source_index, a.sources_per_feature)
        for ratio in (BATCH_RATIOS if a.what == "batch_ratio" else ["0.01"]):
            for feat, ids in feats.items():
                for sid in ids:
                    tag = f"batch_ratio_{ratio}" if a.what == "batch_ratio" else f"source_feature_{feat}"
                    r = run(base + ["-n", "0", "-r", ratio, "-b", str(a.batches), "-s", str(sid)], os.path.join(a.log_dir, f"{tag}_{name}_{sid}.txt"))
                    rows.append({**({"batch_ratio": float(ratio)} if a.what == "batch_ratio" else {}), "source_feature": feat, "source": sid, **r})
    elif a.what == "batch_size":
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
