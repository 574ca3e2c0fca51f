"""Offline data tools of the reference, restated.

* :func:`encode_snap` -- ``encoder/GraphEncoder.h:20-98`` ``EncodeSnapToBin``: SNAP edge-list text to the
  ``.bin`` stream (``int32 V`` then ``int32`` pairs). Ids are rebased to ``id - min_id``, ``V`` is the id
  RANGE ``max - min + 1`` (isolated ids included), the edge order is shuffled. The reference shuffles
  with an unseeded ``rand() % E`` swap per edge (``:72-79``); the default here is a seeded uniform
  permutation, ``shuffle="glibc"`` replays the reference's swaps with glibc's ``rand()`` sequence and
  gives its encoder's output byte for byte. ``#`` comment lines of SNAP files are skipped (the
  reference's ``file >> vid`` loop never terminates on them).
* :func:`reverse_bin` -- ``encoder/GraphEncoder.h:100-131``: swap the endpoints of every edge.
* :func:`workload` -- ``workload/Workload.cpp:30-62`` + ``workload/Graph.h:178-227``: rank vertices by
  (out or in) degree over the whole file or its first 10 % window and write three id files of 10
  sources each: ranks [0,10) exactly, and 10 distinct connected vertices drawn from [10,1000) and from
  [1000,1e6). File names as the reference writes them: ``<bin>_top[window][rev]{10,1000,1000000}.txt``.

CLI:  python -m dynamicppr_amd.tools encode <snap.txt> [--reverse] [--seed N] [--out file.bin]
      python -m dynamicppr_amd.tools workload <file.bin> <directed> <is_window> <is_out_degree> [--seed N]
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import datagen


class GlibcRand:
    """glibc's ``rand()`` (TYPE_3 additive feedback generator of ``random_r``; ``srand(1)`` is the state
    a program that never seeds starts in). The reference's tools shuffle and sample with unseeded
    ``rand()`` (``encoder/GraphEncoder.h:72-79``, ``workload/Graph.h:203``), so on glibc their output is
    in fact deterministic -- and reproducible here."""

    def __init__(self, seed: int = 1):
        r = [0] * 34
        r[0] = seed & 0xFFFFFFFF
        for i in range(1, 31):
            hi, lo = divmod(r[i - 1] if r[i - 1] < 2 ** 31 else r[i - 1] - 2 ** 32, 127773)
            word = 16807 * lo - 2836 * hi
            if word < 0:
                word += 2147483647
            r[i] = word & 0xFFFFFFFF
        for i in range(31, 34):
            r[i] = r[i - 31]
        self.r = r
        for _ in range(310):
            self._step()

    def _step(self):
        r = self.r
        v = (r[-31] + r[-3]) & 0xFFFFFFFF
        r.append(v)
        if len(r) > 64:
            del r[:len(r) - 34]
        return v

    def rand(self) -> int:
        return self._step() >> 1


def encode_snap(txt_path: str, out_path: str | None = None, reverse: bool = False, randomize: bool = True,
                seed: int = 1, shuffle: str = "numpy") -> str:
    rows = []
    with open(txt_path) as f:
        for line in f:
            if not line.strip() or line.lstrip().startswith("#"):
                continue
            a, b = line.split()[:2]
            rows.append((int(a), int(b)))
    edges = np.array(rows, dtype=np.int64).reshape(-1, 2)
    if len(edges) == 0:
        raise ValueError("no edges in " + txt_path)
    lo, hi = int(edges.min()), int(edges.max())
    V = hi - lo + 1                                          # GraphEncoder.h:44
    e1 = (edges[:, 0] - lo).astype(np.int32)
    e2 = (edges[:, 1] - lo).astype(np.int32)
    if reverse:
        e1, e2 = e2, e1
    if randomize and shuffle == "glibc":
        # the reference's own shuffle, swap for swap (encoder/GraphEncoder.h:72-79): byte-identical output to
        # its encoder built against glibc (tests/test_tools.py compares with a fixture produced by it)
        g, n = GlibcRand(seed), len(e1)
        a, b = e1.tolist(), e2.tolist()
        for i in range(n):
            pos = g.rand() % n
            a[i], a[pos] = a[pos], a[i]
            b[i], b[pos] = b[pos], b[i]
        e1, e2 = np.array(a, dtype=np.int32), np.array(b, dtype=np.int32)
    elif randomize:
        perm = np.random.default_rng(seed).permutation(len(e1))
        e1, e2 = e1[perm], e2[perm]
    if out_path is None:                                     # GraphEncoder.h:81-84 naming
        base = os.path.basename(txt_path)
        base = base[:base.find(".txt")] if ".txt" in base else base
        out_path = base + ("_rev.bin" if reverse else ".bin")
    datagen.write_bin(out_path, V, e1, e2)
    return out_path


def reverse_bin(bin_path: str, out_path: str | None = None) -> str:
    V, e1, e2 = datagen.read_bin(bin_path)
    if out_path is None:
        base = os.path.basename(bin_path)
        out_path = base[:base.find(".bin")] + "_rev.bin"
    datagen.write_bin(out_path, V, e2, e1)
    return out_path


def degrees(V, e1, e2, directed):
    deg = np.bincount(e1, minlength=V).astype(np.int64)
    in_deg = np.bincount(e2, minlength=V).astype(np.int64)
    if not directed:
        deg, in_deg = deg + np.bincount(e2, minlength=V), in_deg + np.bincount(e1, minlength=V)
    return deg, in_deg


def choose_vertex_degree_range(deg, in_deg, num, rank_st, rank_ed, is_out_degree, rng):
    """``Graph::ChooseVertexDegreeRange`` (workload/Graph.h:178-227)."""
    V = len(deg)
    rank_ed = min(rank_ed, V)
    cmp_deg = deg if is_out_degree else in_deg
    idx = np.lexsort((np.arange(V), -cmp_deg))               # descending degree, ties by id
    if rank_ed - rank_st < num:
        raise ValueError("rank range smaller than the number of ids wanted")
    if rank_ed - rank_st == num:
        return idx[rank_st:rank_ed].astype(np.int32)
    pool = [int(u) for u in idx[rank_st:rank_ed] if deg[u] > 0 and in_deg[u] > 0]   # "choose the connected ones"
    if len(pool) < num:
        raise ValueError("not enough connected vertices in the rank range")
    return np.array(rng.choice(pool, size=num, replace=False), dtype=np.int32)


def workload(bin_path: str, directed: int, is_window: int, is_out_degree: int, seed: int = 1,
             out_dir: str | None = None, window_ratio: float = 0.1):
    V, e1, e2 = datagen.read_bin(bin_path)
    if is_window:
        n = int(len(e1) * window_ratio)                      # workload/Graph.h:56
        e1, e2 = e1[:n], e2[:n]
    deg, in_deg = degrees(V, e1, e2, directed)
    rng = np.random.default_rng(seed)
    base = os.path.basename(bin_path)
    feature = "top" + ("window" if is_window else "") + ("" if is_out_degree else "rev")
    out = {}
    for count, (lo, hi) in ((10, (0, 10)), (1000, (10, 1000)), (1000000, (1000, 1000000))):
        if lo >= V:
            continue
        try:
            ids = choose_vertex_degree_range(deg, in_deg, 10, lo, hi, bool(is_out_degree), rng)
        except ValueError:
            continue
        path = os.path.join(out_dir or ".", f"{base}_{feature}{count}.txt")   # Workload.cpp:11-13
        with open(path, "w") as f:
            f.write("".join(f"{int(u)}\n" for u in ids))
        out[count] = (path, ids)
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    e = sub.add_parser("encode")
    e.add_argument("txt"); e.add_argument("--reverse", action="store_true"); e.add_argument("--seed", type=int, default=1)
    e.add_argument("--out", default=None); e.add_argument("--no-shuffle", action="store_true")
    e.add_argument("--shuffle", default="numpy", choices=["numpy", "glibc"])
    w = sub.add_parser("workload")
    w.add_argument("bin"); w.add_argument("directed", type=int); w.add_argument("is_window", type=int)
    w.add_argument("is_out_degree", type=int); w.add_argument("--seed", type=int, default=1)
    w.add_argument("--out-dir", default=None)
    a = ap.parse_args()
    if a.cmd == "encode":
        print("write to file", encode_snap(a.txt, a.out, a.reverse, not a.no_shuffle, a.seed, a.shuffle))
    else:
        for count, (path, ids) in workload(a.bin, a.directed, a.is_window, a.is_out_degree, a.seed, a.out_dir).items():
            print(f"top{count} filename={path}: {list(map(int, ids))}")


if __name__ == "__main__":
    main()
