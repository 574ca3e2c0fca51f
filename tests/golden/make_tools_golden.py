#!/usr/bin/env python3
"""Generate tests/golden/tools_toy.npz from the REAL reference data tools (dev container only).

Runs oracle/_ref/encoder (encoder/TestEncoder.cpp + GraphEncoder.h) and oracle/_ref/workload
(workload/Workload.cpp + Graph.h), compiled from /root/reference by `make -C oracle ref`, on a toy
SNAP edge list and stores the inputs and what they wrote: the .bin they encoded (forward and
reversed) and the twelve source-id files of the four workload modes. Data only -- no reference
source text.

    make -C oracle ref && python tests/golden/make_tools_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from dynamicppr_amd import datagen  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def toy_snap(seed=5, n=3000, m=12000, id0=100):
    """A SNAP-style edge list: ids start at id0, 14 hubs with clearly separated degrees (so the top
    ranks have no ties), a random rest, a few isolated ids inside the range."""
    rng = np.random.default_rng(seed)
    src, dst = [], []
    for k in range(14):                      # hub k: out-degree 400 - 20 k, in-degree 300 - 15 k
        hub = 7 * k + 3
        outs = rng.choice(np.arange(200, n), size=400 - 20 * k, replace=False)
        ins = rng.choice(np.arange(200, n), size=300 - 15 * k, replace=False)
        src += [hub] * len(outs); dst += outs.tolist()
        src += ins.tolist(); dst += [hub] * len(ins)
    a = rng.integers(200, n, m)
    b = rng.integers(200, n, m)
    keep = a != b
    src += a[keep].tolist(); dst += b[keep].tolist()
    order = rng.permutation(len(src))
    return (np.array(src)[order] + id0).astype(np.int64), (np.array(dst)[order] + id0).astype(np.int64)


def main():
    s, d = toy_snap()
    out = {"snap.src": s, "snap.dst": d}
    with tempfile.TemporaryDirectory() as tmp:
        txt = os.path.join(tmp, "toy.txt")
        with open(txt, "w") as f:
            for a, b in zip(s, d):
                f.write(f"{a}\t{b}\n")
        for rev in (0, 1):
            subprocess.check_call([os.path.join(REF, "encoder"), txt, str(rev)], cwd=tmp, stdout=subprocess.DEVNULL)
            name = "toy_rev.bin" if rev else "toy.bin"
            V, e1, e2 = datagen.read_bin(os.path.join(tmp, name))
            tag = "rev" if rev else "fwd"
            out[f"bin.{tag}.V"], out[f"bin.{tag}.e1"], out[f"bin.{tag}.e2"] = np.array([V]), e1.copy(), e2.copy()
        binp = os.path.join(tmp, "toy.bin")
        for directed in (1, 0):
            for window in (0, 1):
                for outdeg in (1, 0):
                    subprocess.check_call([os.path.join(REF, "workload"), binp, str(directed), str(window), str(outdeg)],
                                          cwd=tmp, stdout=subprocess.DEVNULL)
                    feature = "top" + ("window" if window else "") + ("" if outdeg else "rev")
                    for count in (10, 1000, 1000000):
                        ids = np.loadtxt(os.path.join(tmp, f"toy.bin_{feature}{count}.txt"), dtype=np.int64)
                        out[f"wl.d{directed}.w{window}.o{outdeg}.{count}"] = ids
    np.savez_compressed(os.path.join(HERE, "tools_toy.npz"), **out)
    print("wrote tools_toy.npz:", {k: v.shape for k, v in out.items() if not k.startswith("wl")}, "+ 24 id files")


if __name__ == "__main__":
    main()
