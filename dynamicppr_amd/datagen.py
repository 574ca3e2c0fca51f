"""Seeded synthetic edge streams in the reference's on-disk ``.bin`` format.

The datasets the reference's experiments use (SNAP com-dblp, com-youtube,
soc-LiveJournal1, twitter-2010, com-friendster) are not available offline, so
every benchmark config has an R-MAT stand-in of matching scale:
``a,b,c,d = 0.57,0.19,0.19,0.05``, vertex ids permuted, self loops dropped, edge
order i.i.d. (R-MAT draws are independent, which mirrors the shuffle the
reference's encoder applies, ``encoder/GraphEncoder.h:72-79``).

File format (``GraphVec.h:43-70``): little-endian ``int32 V`` followed by
``(int32 v1, int32 v2)`` per stream edge. For undirected inputs each pair is
written once and the file is opened with ``-i 0``.

The generator is counter based (splitmix64 of ``(seed, edge index, level
group)``) so that a given ``(scale, edges, seed)`` produces the same stream on
every machine and numpy version.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

_U64 = np.uint64
_MASK16 = _U64(0xFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output per uint64 input (vectorised, wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _permutation(n: int, seed: int) -> np.ndarray:
    keys = _splitmix64(np.arange(n, dtype=_U64) ^ _splitmix64(np.array([seed ^ 0xA5A5A5A5], dtype=_U64)))
    return np.argsort(keys, kind="stable").astype(np.int32)


def rmat_stream(scale: int, edges: int, seed: int, a=0.57, b=0.19, c=0.19, chunk: int = 1 << 22):
    """Return ``(V, e1, e2)``: ``edges`` R-MAT edges over ``V = 2**scale`` ids."""
    V = 1 << scale
    ta = int(a * 65536)
    tb = int((a + b) * 65536)
    tc = int((a + b + c) * 65536)
    perm = _permutation(V, seed)
    out1 = np.empty(edges, dtype=np.int32)
    out2 = np.empty(edges, dtype=np.int32)
    filled = 0
    base = 0
    seed_mix = _splitmix64(np.array([seed], dtype=_U64))[0]
    groups = (scale + 3) // 4
    while filled < edges:
        n = min(chunk, max(edges - filled, 1024) + (edges - filled) // 64 + 64)
        idx = np.arange(base, base + n, dtype=_U64)
        base += n
        src = np.zeros(n, dtype=np.int64)
        dst = np.zeros(n, dtype=np.int64)
        level = 0
        for g in range(groups):
            with np.errstate(over="ignore"):
                h = _splitmix64(idx * _U64(64) + _U64(g) + seed_mix)
            for k in range(4):
                if level >= scale:
                    break
                u = ((h >> _U64(16 * k)) & _MASK16).astype(np.int64)
                sbit = (u >= tb).astype(np.int64)            # quadrants c,d -> row bit 1
                dbit = (((u >= ta) & (u < tb)) | (u >= tc)).astype(np.int64)  # quadrants b,d
                src = (src << 1) | sbit
                dst = (dst << 1) | dbit
                level += 1
        keep = src != dst
        s = perm[src[keep]]
        d = perm[dst[keep]]
        take = min(len(s), edges - filled)
        out1[filled:filled + take] = s[:take]
        out2[filled:filled + take] = d[:take]
        filled += take
    return V, out1, out2


def write_bin(path: str, V: int, e1: np.ndarray, e2: np.ndarray) -> None:
    pairs = np.empty(2 * len(e1), dtype="<i4")
    pairs[0::2] = e1
    pairs[1::2] = e2
    with open(path, "wb") as f:
        np.array([V], dtype="<i4").tofile(f)
        pairs.tofile(f)


def read_bin(path: str):
    """Read a reference ``.bin``: returns ``(V, e1, e2)`` (``GraphVec.h:43-70``)."""
    raw = np.memmap(path, dtype="<i4", mode="r")   # no second copy of a multi-GB file
    V = int(raw[0])
    body = raw[1:]
    n = len(body) // 2
    return V, np.ascontiguousarray(body[0:2 * n:2]), np.ascontiguousarray(body[1:2 * n:2])


@dataclass(frozen=True)
class StandIn:
    """A BASELINE.json config and its seeded stand-in."""
    name: str
    scale: int
    edges: int
    seed: int
    directed: int
    flags: str


STAND_INS = {
    "dblp": StandIn("com-dblp.ungraph", 19, 1_049_866, 1, 0, "-n 0 -r 0.01 -b 100"),
    "youtube": StandIn("com-youtube.ungraph", 20, 2_987_624, 2, 0, "-n 0 -r 0.01 -b 100"),
    "livejournal": StandIn("soc-LiveJournal1", 22, 68_993_773, 3, 1, "-n 0 -r 0.01 -b 100"),
    "twitter": StandIn("twitter-2010", 25, 1_468_365_182, 4, 1, "-n 0 -r 0.01 -b 100"),
    "friendster": StandIn("com-friendster", 27, 1_806_067_135, 5, 0, "-n 1 -c 100000 -l 10000000"),
}


def stand_in_path(key: str, cache_dir: str, limit: int | None = None) -> str:
    cfg = STAND_INS[key]
    tag = "" if limit is None or limit >= cfg.edges else f".first{limit}"
    return os.path.join(cache_dir, f"{cfg.name}.rmat{cfg.scale}.s{cfg.seed}{tag}.bin")


GENERATOR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "rmat_gen")
PROVENANCE: dict = {}   # path -> {"origin": "generated (...)" | "cached", "checksum": ..., "edges": ...}


def file_checksum(path: str) -> str:
    """Fast content checksum of a stream file (xxh3-64 when xxhash is importable, else CRC32)."""
    try:
        import xxhash
        h = xxhash.xxh3_64()
        kind = "xxh3_64"
    except ImportError:  # pragma: no cover
        import zlib

        class _Crc:
            def __init__(self):
                self.v = 0

            def update(self, b):
                self.v = zlib.crc32(b, self.v)

            def hexdigest(self):
                return f"{self.v:08x}"
        h, kind = _Crc(), "crc32"
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return f"{kind}:{h.hexdigest()}"


def generate_bin(path: str, scale: int, edges: int, seed: int, limit: int | None = None) -> str:
    """Write the first ``limit`` (default: all) edges of the seeded stream to ``path``. Uses the
    compiled multi-threaded generator (host/rmat_gen, byte-identical output, tests/test_datagen.py)
    when it has been built, the numpy one otherwise. Returns which one ran."""
    limit = edges if limit is None else min(limit, edges)
    tmp = f"{path}.{os.getpid()}.tmp"   # several processes may generate the same file: publish atomically
    if os.path.exists(GENERATOR):
        import subprocess
        subprocess.check_call([GENERATOR, "--scale", str(scale), "--edges", str(edges), "--seed", str(seed),
                               "--limit", str(limit), "--out", tmp], stdout=subprocess.DEVNULL)
        how = "compiled generator host/rmat_gen"
    else:
        V, e1, e2 = rmat_stream(scale, limit, seed)   # the stream is prefix-stable: first `limit` edges
        write_bin(tmp, V, e1, e2)
        how = "numpy generator datagen.rmat_stream"
    os.replace(tmp, path)
    return how


def ensure_stand_in(key: str, cache_dir: str, limit: int | None = None) -> str:
    """Make sure the stand-in file (or its ``limit``-edge prefix) exists; returns its path and records
    whether it was generated now or found cached, with a checksum (``PROVENANCE[path]``)."""
    cfg = STAND_INS[key]
    os.makedirs(cache_dir, exist_ok=True)
    path = stand_in_path(key, cache_dir, limit)
    n = cfg.edges if limit is None else min(limit, cfg.edges)
    if os.path.exists(path) and os.path.getsize(path) == 4 + 8 * n:
        origin = "cached"
    else:
        origin = "generated (" + generate_bin(path, cfg.scale, cfg.edges, cfg.seed, limit) + ")"
    PROVENANCE[path] = {"origin": origin, "checksum": file_checksum(path), "edges": n, "path": path}
    return path


def stand_in_stream(key: str, cache_dir: str | None = None, limit: int | None = None):
    """``(V, e1, e2, StandIn)`` for a named config; cached as ``.bin`` when asked. With ``limit`` only
    the first ``limit`` stream edges are produced (a sliding-window run reads W + batches*c of them);
    ``StandIn.edges`` stays the full stream length, which the workload derivation needs."""
    cfg = STAND_INS[key]
    if cache_dir:
        V, e1, e2 = read_bin(ensure_stand_in(key, cache_dir, limit))
        return V, e1, e2, cfg
    V, e1, e2 = rmat_stream(cfg.scale, cfg.edges if limit is None else min(limit, cfg.edges), cfg.seed)
    return V, e1, e2, cfg


def ranked_sources(V: int, e1: np.ndarray, e2: np.ndarray, W: int, directed: int, lo: int, hi: int, k: int = 10,
                   seed: int = 1) -> np.ndarray:
    """``k`` source vertices sampled from out-degree ranks ``[lo, hi)`` of the initial window -- what
    ``workload/Workload.cpp:45-55`` writes as the "top1000" file (``lo, hi = 10, 1000``). The
    reference samples with unseeded ``rand()``; here the draw is a seeded permutation of the rank
    range so that every run and every rank picks the same ids."""
    ranked = top_sources(V, e1, e2, W, directed, hi)
    pick = np.sort(_permutation(hi - lo, seed)[:k]) + lo
    return ranked[pick].astype(np.int32)


def top_sources(V: int, e1: np.ndarray, e2: np.ndarray, W: int, directed: int, k: int = 10) -> np.ndarray:
    """Vertices of highest out-degree in the initial window, rank order.

    Mirrors ``workload/Graph.h:178-227`` (sort by degree, take rank ranges); ties
    are broken by smaller id so the choice is deterministic.
    """
    deg = np.bincount(e1[:W], minlength=V).astype(np.int64)
    if not directed:
        deg += np.bincount(e2[:W], minlength=V)
    order = np.lexsort((np.arange(V), -deg))
    return order[:k].astype(np.int32)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description="write a seeded R-MAT stream as a reference .bin")
    ap.add_argument("--scale", type=int, required=True)
    ap.add_argument("--edges", type=int, required=True)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    V, e1, e2 = rmat_stream(a.scale, a.edges, a.seed)
    write_bin(a.out, V, e1, e2)
    print(f"wrote {a.out}: V={V} E={len(e1)}")
