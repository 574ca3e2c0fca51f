"""The reference's four -o variants (cpu/PPRCPUMTMain.cpp:26-32) restated at -t 1
(oracle/dppr_oracle.c: orc_variant_*, following cpu/PPRCPUMTCilkRevVariants.h line by line), and
what the engine's two schedules have to do with them:

  * variants {0 OPTIMIZED, 2 EAGER} are one schedule and {1 FAST_FRONTIER, 3 VANILLA} another: the
    status array of 2 / 3 only replaces the threshold-crossing test as the duplicate filter -- p, r
    and every iteration's frontier LIST are bit-identical within a pair (SURVEY.md section 3.2 saw
    the same on the compiled reference);
  * FAST_FRONTIER (snapshot every frontier residual, then push) visits, iteration by iteration,
    exactly the frontier SETS of the oracle's synchronous schedule C -- the schedule the engine's
    DPPR_SCHEDULE_SYNC reproduces set for set on the GPU (tests/test_engine_gpu.py) -- and needs more
    iterations than variant 0, whose eager reads see same-iteration arrivals (the survey's probe:
    43 vs 26 on its stream).
So `./pagerank -o 1|3` (synchronous schedule) and `-o 0|2` (eager schedule) select schedules the
reference's variants define, not aliases picked by taste. CPU only."""
import numpy as np
import pytest

from dynamicppr_amd import datagen
from oracle import oracle as orc


def run(variant, V, e1, e2, directed, W, c, src, eps, batches, sync=False):
    g = orc.Graph(V, e1, e2, directed, W, c)
    s = orc.State(V, src, eps)
    out = []
    for k in range(batches + 1):
        s.trace(True)
        if k == 0:
            s.sync_execute(g) if sync else s.variant_execute(g, variant)
        else:
            assert not g.stream_updates()
            g.inc_construct(1)
            s.sync_inc_execute(g) if sync else s.variant_inc_execute(g, variant)
        out.append((s.p.copy(), s.r.copy(), s.traced_frontiers(), dict(s.stats())))
    return out


CASES = [(1, 11, 600, 20, 1e-9), (0, 11, 600, 20, 1e-9), (1, 5, 3000, 60, 1e-7), (0, 7, 2000, 40, 1e-9)]


@pytest.mark.parametrize("directed,seed,W,c,eps", CASES)
def test_variant_pairs_are_bit_identical(directed, seed, W, c, eps):
    V, e1, e2 = datagen.rmat_stream(10, 30000, seed)
    src = int(datagen.top_sources(V, e1, e2, W, directed, 1)[0])
    runs = {v: run(v, V, e1, e2, directed, W, c, src, eps, 4) for v in range(4)}
    for a, b in ((0, 2), (1, 3)):
        for (pa, ra, fa, sa), (pb, rb, fb, sb) in zip(runs[a], runs[b]):
            assert np.array_equal(pa, pb) and np.array_equal(ra, rb)
            assert len(fa) == len(fb) and all(np.array_equal(x, y) for x, y in zip(fa, fb))   # same lists, same order
            assert sa == sb
    # ... and the two pairs are different schedules with the same answer to within the tolerance
    for (p0, r0, f0, s0), (p1, r1, f1, s1) in zip(runs[0], runs[1]):
        assert np.max(np.abs(p0 - p1)) < eps and np.max(np.abs(r0)) < eps and np.max(np.abs(r1)) < eps
    assert runs[1][0][3]["iters"] > runs[0][0][3]["iters"]        # from-scratch solve: FF needs more iterations


@pytest.mark.parametrize("directed,seed,W,c,eps", CASES)
def test_fast_frontier_is_the_synchronous_schedule(directed, seed, W, c, eps):
    V, e1, e2 = datagen.rmat_stream(10, 30000, seed)
    src = int(datagen.top_sources(V, e1, e2, W, directed, 1)[0])
    ff = run(1, V, e1, e2, directed, W, c, src, eps, 4)
    sy = run(1, V, e1, e2, directed, W, c, src, eps, 4, sync=True)
    for (pf, rf, ff_fr, sf), (ps, rs, sy_fr, ss) in zip(ff, sy):
        assert len(ff_fr) == len(sy_fr)                            # same number of iterations
        for a, b in zip(ff_fr, sy_fr):
            assert np.array_equal(np.sort(a), np.sort(b))          # same frontier set in every iteration
        assert (sf["iters"], sf["F"], sf["E"]) == (ss["iters"], ss["F"], ss["E"])
        # FF zeroes r[u] at the snapshot, schedule C subtracts the snapshot afterwards: rounding only
        assert np.max(np.abs(pf - ps)) < 1e-15 and np.max(np.abs(rf - rs)) < 1e-15
