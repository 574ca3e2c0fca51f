"""Shared helpers for the test-suite (oracle side)."""
import glob
import os

import numpy as np

from oracle import oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    names = (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return sorted(n for n in names if not n.startswith(("tools_", "fullsize_", "bench_")))   # tools_*: offline data tools; fullsize_*: tests/test_fullsize_golden_gpu.py; bench_*: bench.py + tests/test_bench_golden.py


def load_golden(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = d["config"]
    meta = dict(V=int(cfg[0]), W=int(cfg[1]), c=int(cfg[2]), batches=int(cfg[3]), total=int(cfg[4]),
                edge_count=int(cfg[5]), source=int(cfg[6]), directed=int(cfg[7]),
                eps=float(d["tolerance"][0]), done=int(d["batches_done"][0]))
    return d, meta


def oracle_graph_from_golden(d, meta):
    return orc.Graph(meta["V"], d["stream.e1"], d["stream.e2"], meta["directed"], meta["W"], meta["c"])


def sorted_csr(row, col):
    """Sort every CSR row ascending (the device CSR is sorted; gpu/PPRRevPushGPU.cuh:60-63)."""
    out = col.copy()
    for u in range(len(row) - 1):
        out[row[u]:row[u + 1]].sort()
    return out


def small_stream(scale=9, edges=6000, seed=11):
    from dynamicppr_amd import datagen
    return datagen.rmat_stream(scale, edges, seed)


def window_directed_edges(g):
    """Directed (src, dst) arrays of the oracle graph's current window."""
    s1, s2 = g.window_edges()
    if g.directed:
        return s1, s2
    return np.concatenate([s1, s2]), np.concatenate([s2, s1])


def invariant_max_err_np(p, r, src, dst, V, source, alpha=0.15):
    """SURVEY.md section 0 invariant, evaluated with numpy from the raw window edges:
    p[u] + a r[u] == a [u==s] + (1-a)/(outdeg(u)+1) * sum_{v in out(u)} p[v]."""
    outdeg = np.bincount(src, minlength=V)
    acc = np.bincount(src, weights=p[dst], minlength=V)
    rhs = (1.0 - alpha) / (outdeg + 1.0) * acc
    rhs[source] += alpha
    return float(np.max(np.abs(p + alpha * r - rhs)))


class Scenario:
    """An oracle graph + state and a HIP engine fed with the same stream."""

    def __init__(self, V, e1, e2, directed, W, c, source, eps, schedule=0, n_epochs=1, **tuning):
        from dynamicppr_amd import engine as eng
        self.V, self.W, self.c, self.directed, self.source, self.eps = V, W, c, directed, source, eps
        self.g = orc.Graph(V, e1, e2, directed, W, c)
        self.s = orc.State(V, source, eps)
        self.e = eng.Engine(V, W, directed, c, n_epochs=n_epochs, schedule=schedule, **tuning)
        w1, w2 = self.g.window_edges()
        self.e.load_window(w1, w2)
        self.slot = self.e.add_source(source)

    def advance_graphs(self):
        """One StreamUpdates + graph rebuild on both sides. False when the stream is over."""
        if self.g.stream_updates():
            return False
        self.g.inc_construct(1)
        b1, b2, ins = self.g.batch()
        n1, n2 = self.g.new_stream()
        self.e.set_batch(b1, b2, ins)
        self.e.slide(n1, n2)
        return True
