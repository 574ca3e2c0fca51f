#!/usr/bin/env python3
"""Generate tests/golden/bench_livejournal.npz (dev container; ~10 minutes on 8 cores).

The headline run of bench.py (BASELINE.json configs[2]: soc-LiveJournal1 stand-in, -r 0.01 -b 100, the 10 sources rank 0
draws from a top1000 file) followed by the oracle's restatement of cpu/PPRCPUMTCilkRev.h at -t 1 (oracle/dppr_oracle.c:
orc_cilk_execute / orc_cilk_inc_execute) -- EVERY source through EVERY batch at -t 1, i.e. schedule A itself, no
multi-threaded positioning. Kept per source at each checkpoint batch (25 = the driver's --steps 20 --warmup 5, 35 = the
script's own default): p at a fixed seeded sample of 100 000 vertices and at the 1 000 vertices of largest p, sum(p),
max|r|. bench.py compares ALL ten sources with it at the end of its timed region when stream checksum, sources, eps and
batch number match (parity.sources_compared = 10) -- the live CPU leg then only has to carry `cpu_baseline`.

    python tests/golden/make_bench_golden.py [--threads 8] [--checkpoints 25,35]

Sources are independent: per batch the graph is advanced once and a pool of Python threads (ctypes drops the GIL)
runs one State each over it.
"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from dynamicppr_amd import datagen, stream as st  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SAMPLE = 100_000
TOP = 1_000
EPS = 1e-9
DATA = "/tmp/dppr_data"
SAMPLE_SEED = 20261004


def stream_checksum(e1, e2, n):
    """Checksum of the first n stream edges (what a run up to some batch has read), independent of how long the file is."""
    import xxhash
    h = xxhash.xxh3_64()
    h.update(np.ascontiguousarray(e1[:n], dtype="<i4").tobytes())
    h.update(np.ascontiguousarray(e2[:n], dtype="<i4").tobytes())
    return f"xxh3_64:{h.hexdigest()}"


def record(s, sample):
    p = s.p
    top = np.argpartition(p, -TOP)[-TOP:]
    top = top[np.argsort(-p[top], kind="stable")].astype(np.int32)
    return dict(p_sample=p[sample].copy(), top_ids=top, top_p=p[top].copy(), sum_p=float(np.sum(p)),
                max_abs_r=float(np.max(np.abs(s.r))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", default="livejournal")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--checkpoints", default="25,35")
    ap.add_argument("--sources", type=int, default=10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cps = sorted(int(x) for x in a.checkpoints.split(","))
    cfg = datagen.STAND_INS[a.key]
    f = cfg.flags.split()
    opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
    wl = st.workload_config(cfg.edges, 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1.0)), int(opt.get("-b", 0)),
                            int(opt.get("-c", 0)), int(opt.get("-l", 0)))
    W, c = wl.window, wl.per_batch
    limit = W + (cps[-1] + 1) * c
    t0 = time.time()
    V, e1, e2, _ = datagen.stand_in_stream(a.key, DATA, limit=limit)
    # bench.py, rank 0: 10 ids sampled from degree ranks [10, 1000) of the initial window, seed 1 + rank
    sources = [int(s) for s in datagen.ranked_sources(V, e1, e2, W, cfg.directed, 10, 1000, 10, seed=1)[:a.sources]]
    sample = np.sort(datagen._permutation(V, SAMPLE_SEED)[:SAMPLE]).astype(np.int32)
    out = {"config": np.array([V, W, c, cfg.directed], dtype=np.int64), "eps": np.array([EPS]), "sources": np.array(sources, dtype=np.int32),
           "sample": sample, "checkpoints": np.array(cps, dtype=np.int64),
           "schedule": np.array("cpu/PPRCPUMTCilkRev.h at -t 1 (oracle restatement), every source through every batch")}
    g = orc.Graph(V, e1, e2, cfg.directed, W, c)
    states = [orc.State(V, s, EPS) for s in sources]
    pool = ThreadPoolExecutor(max_workers=max(1, a.threads))
    list(pool.map(lambda s: s.cilk_execute(g), states))
    print(f"[init] {len(states)} from-scratch solves ({time.time() - t0:.0f} s)", flush=True)
    for k in range(1, cps[-1] + 1):
        assert not g.stream_updates()
        g.inc_construct(1)
        list(pool.map(lambda s: s.cilk_inc_execute(g), states))
        if k in cps:
            out[f"checksum.b{k}"] = np.array(stream_checksum(e1, e2, W + k * c))
            for i, s in enumerate(states):
                for name, val in record(s, sample).items():
                    out[f"b{k}.s{i}.{name}"] = np.asarray(val)
            print(f"[batch {k}] checkpoint: sum p of source 0 = {float(np.sum(states[0].p)):.12f}, "
                  f"max|r| {max(float(np.max(np.abs(s.r))) for s in states):.3e}", flush=True)
        print(f"[batch {k}] {time.time() - t0:.0f} s", flush=True)
    path = a.out or os.path.join(HERE, f"bench_{a.key}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
