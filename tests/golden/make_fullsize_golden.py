#!/usr/bin/env python3
"""Generate tests/golden/fullsize_{twitter,friendster}.npz (dev container; tens of minutes of CPU).

BASELINE.json configs[3] / configs[4] at their real window sizes (seeded stand-ins): the oracle's
restatement of cpu/PPRCPUMTCilkRev.h at -t 1 (oracle/dppr_oracle.c: orc_cilk_execute /
orc_cilk_inc_execute) solves EVERY source of the config from scratch and through one batch; what is
kept per source and solve: iteration count, statistics, sum(p), max|r|, p at a fixed seeded sample of
vertices and at the 1 000 vertices of largest p. tests/test_fullsize_golden_gpu.py regenerates the
same stream prefix on the GPU box (checksum compared) and holds the HIP engine -- twitter single
source, twitter 8-source group, friendster 10-source group -- to these values within 1e-9.

    python tests/golden/make_fullsize_golden.py twitter --threads 4
    python tests/golden/make_fullsize_golden.py friendster --threads 3

Sources are independent and the oracle has no global state: a pool of Python threads (ctypes drops
the GIL) runs one State each over two shared read-only graphs (window before / after the batch).
"""
import argparse
import os
import sys
import threading
import time

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


def config_sources(key, V, e1, e2, W, directed):
    """configs[3]: 8 of the top-10 file; configs[4]: the 10 ids of a top1000 file."""
    if key == "twitter":
        return [int(x) for x in datagen.top_sources(V, e1, e2, W, directed, 10)[:8]]
    return [int(x) for x in datagen.ranked_sources(V, e1, e2, W, directed, 10, 1000, 10)]


def workload(key):
    cfg = datagen.STAND_INS[key]
    f = cfg.flags.split()
    opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
    wl = st.workload_config(cfg.edges, 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1.0)), int(opt.get("-b", 0)),
                            int(opt.get("-c", 0)), int(opt.get("-l", 0)))
    return cfg, wl


def record(s, sample):
    p = s.p
    top = np.argpartition(p, -TOP)[-TOP:]
    top = top[np.argsort(-p[top], kind="stable")].astype(np.int32)
    stats = s.stats()
    return dict(p_sample=p[sample].copy(), top_ids=top, top_p=p[top].copy(), sum_p=float(np.sum(p)),
                max_abs_r=float(np.max(np.abs(s.r))), iteration_id=int(s._s.contents.iteration_id),
                stats=np.array([stats["iters"], stats["F"], stats["E"], stats["N"]], dtype=np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("key", choices=["twitter", "friendster", "livejournal"])
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--sources", type=int, default=0, help="only the first N sources (trial runs)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    key = a.key
    cfg, wl = workload(key)
    W, c = wl.window, wl.per_batch
    limit = W + 2 * c
    t0 = time.time()
    V, e1, e2, _ = datagen.stand_in_stream(key, DATA, limit=limit)
    prov = datagen.PROVENANCE[datagen.stand_in_path(key, DATA, limit)]
    print(f"[stream] {key}: {prov['origin']}, {prov['edges']} edges, {prov['checksum']} ({time.time() - t0:.0f} s)", flush=True)
    sources = config_sources(key, V, e1, e2, W, cfg.directed)
    if a.sources:
        sources = sources[:a.sources]
    g0 = orc.Graph(V, e1, e2, cfg.directed, W, c)
    g1 = orc.Graph(V, e1, e2, cfg.directed, W, c)
    assert not g1.stream_updates()
    g1.inc_construct(1)
    del e1, e2
    print(f"[graphs] two windows built ({time.time() - t0:.0f} s)", flush=True)
    sample = np.sort(datagen._permutation(V, 20261003)[:SAMPLE]).astype(np.int32)
    out = {"config": np.array([V, W, c, cfg.directed, limit], dtype=np.int64), "eps": np.array([EPS]),
           "checksum": np.array(prov["checksum"]), "sources": np.array(sources, dtype=np.int32), "sample": sample}
    lock = threading.Lock()
    todo = list(enumerate(sources))

    def worker():
        while True:
            with lock:
                if not todo:
                    return
                i, sv = todo.pop(0)
            t = time.time()
            s = orc.State(V, sv, EPS)
            s.cilk_execute(g0)
            rec0 = record(s, sample)
            t1 = time.time()
            s.reset_stats()
            s.cilk_inc_execute(g1)
            rec1 = record(s, sample)
            with lock:
                for k, rec in enumerate((rec0, rec1)):
                    for name, val in rec.items():
                        out[f"s{i}.k{k}.{name}"] = np.asarray(val)
                print(f"[source {i} = {sv}] init {t1 - t:.0f} s ({rec0['stats'][0]} iterations, sum E {rec0['stats'][2]}), "
                      f"batch {time.time() - t1:.0f} s ({rec1['stats'][0]} iterations, sum E {rec1['stats'][2]}); "
                      f"sum p {rec0['sum_p']:.12f} -> {rec1['sum_p']:.12f}, max|r| {rec1['max_abs_r']:.3e}", flush=True)
            del s

    threads = [threading.Thread(target=worker) for _ in range(max(1, a.threads))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    path = a.out or os.path.join(HERE, f"fullsize_{key}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
