#!/usr/bin/env python3
"""VERDICT r05 item 2(a), go / no-go count: distinct (head, B-block) pairs / edges of a binned-sweep layout (dppr_binned.hpp).

Pass 1 of the binned sweep stores x[head] once per EDGE; if many edges of a B-block share a head, storing it once per
(head, B-block) run would cut the 16 of 24 streamed bytes per edge that carry values. CPU emulation of the engine's numbering
(dppr_idspace.hpp: hashed, blocks of falling in-degree) and B-cut (dppr_host_graph.hpp bin_cut: every `cap` vertices, every
`target` edges, a row of >= target / 4 edges alone) on the window of a stand-in.   usage: dedup_count.py file.bin W directed [cap target]"""
import sys
import numpy as np


def id_hash(v):
    z = v.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def main():
    path, W, directed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    raw = np.memmap(path, dtype="<i4", mode="r")
    V = int(raw[0])
    body = raw[1:1 + 2 * W]
    e1, e2 = np.ascontiguousarray(body[0::2]), np.ascontiguousarray(body[1::2])
    if not directed:
        e1, e2 = np.concatenate([e1, e2]), np.concatenate([e2, e1])
    Ed = len(e1)
    indeg = np.bincount(e2, minlength=V)
    outdeg = np.bincount(e1, minlength=V)
    live = np.flatnonzero((indeg + outdeg) > 0)
    n = len(live)
    # numbering: block of falling in-degree (top 8 K, 16 K, ... 512 K, rest), hashed inside
    d = indeg[live]
    order = np.argsort(-d, kind="stable")
    rank = np.empty(n, np.int64)
    rank[order] = np.arange(n)
    block = np.zeros(n, np.int64)
    k = 8192
    while k <= 524288:
        block += rank >= k
        k *= 2
    with np.errstate(over="ignore"):
        key = (block.astype(np.uint64) << np.uint64(59)) | (id_hash(live) >> np.uint64(5))
    pos = np.argsort(key, kind="stable")
    newid = np.full(V, -1, np.int64)
    newid[live[pos]] = np.arange(n)
    v, u = newid[e1], newid[e2]
    od = np.bincount(v, minlength=n)
    row_ptr = np.concatenate([[0], np.cumsum(od)])
    for cap, target in ([(int(sys.argv[4]), int(sys.argv[5]))] if len(sys.argv) > 5 else
                        [(3072, min(max(Ed // 256, 16384), 196608)), (7680, 196608), (7680, 786432), (7680, 1 << 22)]):
        cuts = set(range(0, n, cap))
        K = (Ed + target - 1) // target
        cuts.update(int(x) for x in np.searchsorted(row_ptr, np.arange(1, K) * target, side="left"))
        big = np.flatnonzero(od >= max(target // 4, 1))
        cuts.update(int(x) for x in big)
        cuts.update(int(x) + 1 for x in big)
        cuts.add(n)
        cut = np.array(sorted(c for c in cuts if c <= n))
        bblk = np.searchsorted(cut, np.arange(n), side="right") - 1
        nb = len(cut) - 1
        pair = u * np.int64(nb) + bblk[v]
        pair.sort()
        distinct = 1 + int(np.count_nonzero(pair[1:] != pair[:-1]))
        rho = distinct / Ed
        print(f"{path}: live {n} edges {Ed} cap_b {cap} target {target} -> n_b {nb}: distinct (head, B-block) pairs {distinct} "
              f"= {rho:.4f} of the edges; streamed bytes per edge 24 -> {2 + 22 * rho:.2f} (2 + 22 rho)", flush=True)


if __name__ == "__main__":
    main()
