"""bench.py's pure-Python pieces on the CPU: the byte model behind `roofline.frac`, the HBM plan that decides
whether a run is refused up front (item 1), and which sources of a rank run in series."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class _Solver:
    def kernel_name(self, ps):
        return "k_test (a test kernel)"


def _roof(ps, S, as_group):
    a = types.SimpleNamespace(config=None, bin=None, batch_edges=None, no_ceilings=True)
    eng = types.SimpleNamespace(build_id=lambda: "0123456789abcdef")
    base = dict(persist_launches=0, binned_sweeps=0, sweep_launches=0, iterations=10, push_launches=10, push_ms=1.0, sum_F=1000, sum_E=100000, sum_N=1000)
    return bench.roofline_block(a, dict(base, **ps), S, as_group, _Solver(), eng, 0, 5)


def test_sweeps_are_priced_at_what_they_must_move_and_pushes_at_surveys_unit():
    F, E, N, t = 1000, 100000, 1000, 1e-3
    push = _roof({}, 1, False)
    assert push["form"] == "push" and abs(push["achieved"] - (72 * F + 24 * E + 4 * N) / t / 1e9) < 1e-2
    assert abs(push["work_rate_survey_unit"]["GBps"] - push["achieved"]) < 1e-2 and "24 E" in push["bytes_model"]
    for form, ps in (("pull", dict(sweep_launches=10)), ("binned", dict(sweep_launches=10, binned_sweeps=10)), ("resident", dict(persist_launches=10))):
        r = _roof(ps, 1, False)
        assert r["form"] == form and abs(r["achieved"] - (72 * F + 12 * E + 4 * N) / t / 1e9) < 1e-2     # one source: 8 + 4 bytes per active edge
        assert r["work_rate_survey_unit"]["GBps"] > r["achieved"] and "(8 + 4 / 1)" in r["bytes_model"]
        assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-5 and r["frac_traffic"] is None and r["traffic"] is None
    g = _roof(dict(sweep_launches=10), 10, True)                                                          # ten sources share the column entry
    assert abs(g["achieved"] - (72 * F + 8.4 * E + 4 * N) / t / 1e9) < 1e-2 and "(8 + 4 / 10)" in g["bytes_model"]
    s = _roof(dict(sweep_launches=10, binned_sweeps=10), 3, False)                                        # three sources IN SERIES: single-source pricing
    assert "(8 + 4 / 1)" in s["bytes_model"]


def test_hbm_plan_counts_what_the_engine_holds():
    V, W, c = 1 << 25, 146_836_518, 1_468_365
    ring = bench.hbm_plan(V, W, 1, c, 1, False, True, bench.RING)
    pre = bench.hbm_plan(V, W, 1, c, 1, False, True, 26)
    assert ring["resident_epochs"] == 3 and pre["plan_bytes"] - ring["plan_bytes"] == 23 * ring["epoch_bytes"]
    assert 2.5e9 < ring["epoch_bytes"] < 5e9 and 15e9 < ring["plan_bytes"] < 40e9       # (measured on the twitter stand-in: 24.5 GB peak)
    und = bench.hbm_plan(V, W, 0, c, 1, False, True, 3)
    assert und["epoch_bytes"] > 1.8 * ring["epoch_bytes"]                                # an undirected window holds both orientations
    g10 = bench.hbm_plan(1 << 27, 180_606_713, 0, 100_000, 10, True, False, 3)
    assert g10["source_state_bytes"] == 416 * (1 << 27) and 80e9 < g10["plan_bytes"] < 110e9   # friendster group: 89.6 GB measured
    assert 26 * g10["epoch_bytes"] + g10["engine_bytes"] + g10["source_state_bytes"] > 0.75 * 288e9   # what pre-staging the driver's step counts would have asked for


def test_series_policy_constants():
    assert bench.SERIES_MAX == 3 and bench.BIG_WINDOW == 4_000_000 and bench.RING == 3 and bench.CPU_SAMPLE_BATCHES == 6
