"""Shared helpers for the test-suite (oracle side)."""
import glob
import os

import numpy as np

from oracle import oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


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
