"""tests/golden/bench_livejournal.npz (made by tests/golden/make_bench_golden.py: the headline's ten sources through every batch at
-t 1 on the CPU restatement) is the fixture bench.py compares ALL of rank 0's sources with at the end of its timed region (VERDICT r05
item 4). Here, on the CPU: the fixture belongs to the stream bench.py generates (sources, shape, checksum of the edges read up to
each checkpoint), bench.py's loader accepts it exactly when everything matches, and its contents are self-consistent."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamicppr_amd import datagen, stream as st  # noqa: E402


def headline_stream(tmp):
    cfg = datagen.STAND_INS["livejournal"]
    wl = st.workload_config(cfg.edges, 0.1, 0, 0.01, 100)
    need = wl.window + 36 * wl.per_batch
    V, e1, e2 = datagen.read_bin(datagen.ensure_stand_in("livejournal", str(tmp), need))
    return cfg, wl, V, e1, e2


def test_fixture_matches_the_headline_stream_and_the_loader(tmp_path):
    import bench
    z = np.load(os.path.join(ROOT, "tests", "golden", "bench_livejournal.npz"))
    cfg, wl, V, e1, e2 = headline_stream(tmp_path)
    W, c = wl.window, wl.per_batch
    assert [int(x) for x in z["config"]] == [V, W, c, cfg.directed] and float(z["eps"][0]) == 1e-9
    sources = [int(s) for s in datagen.ranked_sources(V, e1, e2, W, cfg.directed, 10, 1000, 10, seed=1)]   # bench.py, rank 0
    assert [int(x) for x in z["sources"]] == sources
    assert [int(x) for x in z["checkpoints"]] == [25, 35]          # the driver's --steps 20 --warmup 5, and the script's own default
    a = types.SimpleNamespace(config="livejournal", bin=None, batch_edges=None, schedule="eager", eps=1e-9)
    for k in (25, 35):
        g = bench.load_bench_golden(a, V, W, c, cfg.directed, sources, k, e1, e2)
        assert g is not None and len(g["p_sample"]) == 10 and len(g["sample"]) == 100_000
        for i in range(10):
            p_s, ids, p_t = g["p_sample"][i], g["top_ids"][i], g["top_p"][i]
            assert np.all(np.isfinite(p_s)) and np.all(np.diff(p_t) <= 0) and len(np.unique(ids)) == 1000   # (p may dip a hair below 0: negative residuals are pushed too)
            assert float(z[f"b{k}.s{i}.max_abs_r"]) < 1e-9 and 0 < float(z[f"b{k}.s{i}.sum_p"])
            both = np.intersect1d(g["sample"], ids)                  # vertices that are in the sample AND among the top 1000: one value
            if len(both):
                assert np.array_equal(p_s[np.searchsorted(g["sample"], both)], p_t[np.array([np.flatnonzero(ids == v)[0] for v in both])])
    # anything that does not match is refused: another step count, other sources, another stream, another eps
    assert bench.load_bench_golden(a, V, W, c, cfg.directed, sources, 30, e1, e2) is None
    assert bench.load_bench_golden(a, V, W, c, cfg.directed, sources[::-1], 25, e1, e2) is None
    e1b = e1.copy()
    e1b[W + 3] ^= 1
    assert bench.load_bench_golden(a, V, W, c, cfg.directed, sources, 25, e1b, e2) is None
    assert bench.load_bench_golden(types.SimpleNamespace(**{**a.__dict__, "eps": 1e-8}), V, W, c, cfg.directed, sources, 25, e1, e2) is None
    # a rank's shorter source list (a prefix) is served from the same fixture
    assert len(bench.load_bench_golden(a, V, W, c, cfg.directed, sources[:4], 25, e1, e2)["p_sample"]) == 4
