"""Stand-in stream generators: the compiled multi-threaded one (dynamicppr_amd/host/rmat_gen) must
produce byte-identical files to the numpy one (datagen.rmat_stream), whole streams and prefixes,
whatever the thread count. CPU only."""
import os
import subprocess

import numpy as np
import pytest

from dynamicppr_amd import datagen

GEN = datagen.GENERATOR


@pytest.fixture(scope="module", autouse=True)
def built():
    host = os.path.dirname(GEN)
    subprocess.check_call(["make", "-C", host, "-s", "gen"])
    assert os.path.exists(GEN)


@pytest.mark.parametrize("scale,edges,seed,threads", [(12, 50000, 7, 3), (9, 6000, 11, 8), (5, 3000, 1, 1),
                                                        (16, 300000, 4, 5), (13, 1, 3, 2)])
def test_compiled_generator_is_byte_identical_to_numpy(tmp_path, scale, edges, seed, threads):
    V, e1, e2 = datagen.rmat_stream(scale, edges, seed)
    ref = tmp_path / "numpy.bin"
    datagen.write_bin(str(ref), V, e1, e2)
    out = tmp_path / "compiled.bin"
    subprocess.check_call([GEN, "--scale", str(scale), "--edges", str(edges), "--seed", str(seed), "--out", str(out),
                           "--threads", str(threads)], stdout=subprocess.DEVNULL)
    want = ref.read_bytes()
    assert out.read_bytes() == want
    # a prefix of the stream is the stream's prefix (what a bounded sliding-window run needs)
    lim = max(1, edges // 3)
    subprocess.check_call([GEN, "--scale", str(scale), "--edges", str(edges), "--seed", str(seed), "--out", str(out),
                           "--limit", str(lim)], stdout=subprocess.DEVNULL)
    assert out.read_bytes() == want[:4 + 8 * lim]
    V2, a1, a2 = datagen.read_bin(str(out))
    assert V2 == V and np.array_equal(a1, e1[:lim]) and np.array_equal(a2, e2[:lim])


def test_stand_in_cache_records_provenance(tmp_path):
    d = str(tmp_path)
    V, e1, e2, cfg = datagen.stand_in_stream("dblp", d, limit=20000)
    path = datagen.stand_in_path("dblp", d, 20000)
    assert len(e1) == 20000 and cfg.edges == 1_049_866 and V == 1 << cfg.scale
    first = dict(datagen.PROVENANCE[path])
    assert first["origin"].startswith("generated") and first["edges"] == 20000
    datagen.stand_in_stream("dblp", d, limit=20000)
    again = datagen.PROVENANCE[path]
    assert again["origin"] == "cached" and again["checksum"] == first["checksum"]
    # the numpy generator gives the same prefix
    _, n1, n2 = datagen.rmat_stream(cfg.scale, 20000, cfg.seed)
    assert np.array_equal(n1, e1) and np.array_equal(n2, e2)


def test_ranked_sources_are_deterministic_and_in_range():
    V, e1, e2 = datagen.rmat_stream(12, 60000, 5)
    W = 6000
    a = datagen.ranked_sources(V, e1, e2, W, 1, 10, 1000, 10)
    b = datagen.ranked_sources(V, e1, e2, W, 1, 10, 1000, 10)
    assert np.array_equal(a, b) and len(set(a.tolist())) == 10
    ranked = datagen.top_sources(V, e1, e2, W, 1, 1000).tolist()
    pos = [ranked.index(int(s)) for s in a]
    assert all(10 <= p < 1000 for p in pos) and pos == sorted(pos)
