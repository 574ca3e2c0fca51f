"""Renumbering of the internal ids on long streams (include/dppr.h: dppr_set_renumbering): vertices whose
last edge left the window are parked with their state, parked vertices that come back get their rows back,
and p / r are what a run without renumbering (and the oracle) gives.

The streams here churn through the id range: an edge's endpoints come from a band that drifts upwards, a
few percent reach back to ids the window forgot long ago (revivals), and a handful of hubs stay connected
to everything so that the sources keep pushing into the band."""
import numpy as np
import pytest

from dynamicppr_amd import engine as eng
from oracle import oracle as orc
from tests.util import invariant_max_err_np, sorted_csr, window_directed_edges

pytestmark = pytest.mark.gpu

NORTH_STAR_TOL = 1e-9
SYNC_TOL = 1e-14


def churn_stream(V, n, band, seed, back=0.04, hubs=4):
    rng = np.random.default_rng(seed)
    lo = (np.arange(n) * (V - band) // n).astype(np.int64)
    a = lo + rng.integers(0, band, n)
    b = lo + rng.integers(0, band, n)
    reach = rng.random(n) < back  # an endpoint from anywhere below the band: mostly vertices without an edge by now
    b[reach] = rng.integers(0, np.maximum(lo[reach], 1))
    hub = rng.random(n) < 0.15
    b[hub] = rng.integers(0, hubs, hub.sum())
    same = a == b
    b[same] = (a[same] + 1) % V
    return a.astype(np.int32), b.astype(np.int32)


def live_count(g, V):
    w1, w2 = g.window_edges()
    return len(np.unique(np.concatenate([w1, w2])))


# the launch forms whose idle state a renumbering has to leave intact: resident sweeps (default), per-iteration pull
# sweeps with and without the activity bitmap, push only (hub table + big rows), mixed with one-iteration chunks
FORMS = [dict(), dict(pull_min_frontier=1, persistent=0), dict(pull_min_frontier=1, persistent=0, sweep_bitmap=1, pull_block=256),
         dict(hub_min_degree=3, big_row_edges=8, pull_min_frontier=-1), dict(pull_min_frontier=40, chunk_iters=1),
         dict(pull_min_frontier=1, chunk_iters=3)]
FORM_IDS = ["default", "pull-per-iteration", "pull-bitmap-wg256", "push-hubs-bigrows", "mixed-chunk1", "resident-3-sweeps"]


@pytest.mark.parametrize("tuning", FORMS, ids=FORM_IDS)
@pytest.mark.parametrize("directed", [1, 0])
def test_long_stream_sync_schedule_slot_and_group_renumbered(directed, tuning):
    V, W, c, eps, batches = 4096, 1500, 100, 1e-9, 60
    e1, e2 = churn_stream(V, W + batches * c, 400, 5 + directed)
    g = orc.Graph(V, e1, e2, directed, W, c)
    e = eng.Engine(V, W, directed, c, schedule=eng.SCHEDULE_SYNC, **tuning)
    e.set_renumbering(1, growth_pct=10, min_parked=16)
    e.load_window(*g.window_edges())
    sources = [0, 1, int(e1[0]), 2, 3]  # e1[0]: a band vertex that retires early (a SOURCE is never parked)
    slot_a, slot_b = e.add_source(sources[0]), e.add_source(sources[2])
    gid = e.add_source_group(sources)
    states = [orc.State(V, s, eps) for s in sources]
    for s in states:
        s.sync_execute(g)
    e.init_solve(slot_a, eps)
    e.init_solve(slot_b, eps)
    e.group_init_solve(gid, eps)
    ids_seen = []
    for k in range(1, batches + 1):
        assert not g.stream_updates()
        g.inc_construct(1)
        e.set_batch(*g.batch())
        e.slide(*g.new_stream())
        for s in states:
            s.sync_inc_execute(g)
        e.update(slot_a, eps)
        e.update(slot_b, eps)
        e.group_update(gid, eps)
        ids_seen.append(e.id_space()["ids"])
        if k % 3 == 0 or k == batches:
            for slot, i in ((slot_a, 0), (slot_b, 2)):
                p, r = e.read(slot)
                assert np.max(np.abs(p - states[i].p)) < SYNC_TOL and np.max(np.abs(r - states[i].r)) < SYNC_TOL, (k, i)
            for i, s in enumerate(states):
                p, r = e.group_read(gid, i)
                assert np.max(np.abs(p - s.p)) < SYNC_TOL and np.max(np.abs(r - s.r)) < SYNC_TOL, (k, i)
            # the device graph in external ids is still the oracle's
            row, col, deg = e.read_graph()
            orow, ocol = g.flatten(1)
            assert np.array_equal(row, orow) and np.array_equal(col, sorted_csr(orow, ocol)) and np.array_equal(deg, g.deg())
    sp = e.id_space()
    assert sp["renumberings"] >= 3 and sp["revivals"] > 0 and sp["parked"] > 0, sp
    # the swept id space follows the live vertices instead of everything ever seen
    seen = len(np.unique(np.concatenate([e1[:W + batches * c], e2[:W + batches * c]])))
    assert sp["ids"] < 1.35 * live_count(g, V) + 64 and sp["ids"] + sp["parked"] <= seen + len(sources), (sp, seen)
    st = e.group_stats(gid)
    assert st["sum_F"] == sum(s.stats()["F"] for s in states) and st["sum_E"] == sum(s.stats()["E"] for s in states)


def test_long_stream_eager_schedule_against_cilk_oracle_and_unrenumbered_run():
    V, W, c, eps, batches, directed = 4096, 2000, 120, 1e-9, 45, 1
    e1, e2 = churn_stream(V, W + batches * c, 500, 21)
    g = orc.Graph(V, e1, e2, directed, W, c)
    s = orc.State(V, 0, eps)
    engines = []
    for renumber in (1, 0):
        e = eng.Engine(V, W, directed, c)
        e.set_renumbering(renumber, growth_pct=8, min_parked=8)
        e.load_window(*g.window_edges())
        engines.append((e, e.add_source(0)))
    s.cilk_execute(g)
    for e, slot in engines:
        e.init_solve(slot, eps)
    for k in range(1, batches + 1):
        assert not g.stream_updates()
        g.inc_construct(1)
        s.cilk_inc_execute(g)
        for e, slot in engines:
            e.set_batch(*g.batch())
            e.slide(*g.new_stream())
            e.update(slot, eps)
        if k % 5 == 0:
            src, dst = window_directed_edges(g)
            for e, slot in engines:
                p, r = e.read(slot)
                assert np.max(np.abs(r)) < eps
                assert np.max(np.abs(p - s.p)) < NORTH_STAR_TOL, k
                assert invariant_max_err_np(p, r, src, dst, V, 0) < 1e-13
    a, b = engines[0][0].id_space(), engines[1][0].id_space()
    assert a["renumberings"] >= 2 and b["renumberings"] == 0 and b["parked"] == 0
    assert a["ids"] < b["ids"]


def test_revivals_while_states_lag_behind_prestaged_epochs():
    """A parked vertex comes back in an epoch that is staged while the solver states are still several epochs
    behind: its rows move at slide time, the older epochs never see the id it gets."""
    V, W, c, eps, directed = 4096, 1200, 80, 1e-9, 0
    ahead, rounds = 6, 6
    e1, e2 = churn_stream(V, W + (20 + ahead * rounds) * c, 300, 33, back=0.08)
    g = orc.Graph(V, e1, e2, directed, W, c)
    sources = [0, 1, 2]
    states = [orc.State(V, s, eps) for s in sources]
    e = eng.Engine(V, W, directed, c, n_epochs=ahead + 2, schedule=eng.SCHEDULE_SYNC)
    e.set_renumbering(1, growth_pct=10, min_parked=8)
    e.load_window(*g.window_edges())
    slot = e.add_source(sources[0])
    gid = e.add_source_group(sources)
    for s in states:
        s.sync_execute(g)
    e.init_solve(slot, eps)
    e.group_init_solve(gid, eps)

    def check(tag):
        p, r = e.read(slot)
        assert np.max(np.abs(p - states[0].p)) < SYNC_TOL and np.max(np.abs(r - states[0].r)) < SYNC_TOL, tag
        for i, s in enumerate(states):
            p, r = e.group_read(gid, i)
            assert np.max(np.abs(p - s.p)) < SYNC_TOL and np.max(np.abs(r - s.r)) < SYNC_TOL, (tag, i)

    for k in range(20):  # in step: renumberings happen here
        assert not g.stream_updates()
        g.inc_construct(1)
        e.set_batch(*g.batch())
        e.slide(*g.new_stream())
        for s in states:
            s.sync_inc_execute(g)
        e.update(slot, eps)
        e.group_update(gid, eps)
    check("in step")
    before = e.id_space()
    assert before["renumberings"] >= 1 and before["parked"] > 0
    for rnd in range(rounds):  # `ahead` epochs staged, then applied one by one (the oracle keeps step with the updates)
        staged = []
        for k in range(ahead):
            assert not g.stream_updates()
            g.inc_construct(1)
            e.set_batch(*g.batch())
            staged.append(e.slide(*g.new_stream()))
            for s in states:
                s.sync_inc_execute(g)  # (the oracle's states run ahead with the graph: compared after the last update)
        for ep in staged:
            e.update(slot, eps, epoch=ep)
            e.group_update(gid, eps, epoch=ep)
        check(("staged", rnd))
    after = e.id_space()
    assert after["revivals"] > before["revivals"], (before, after)


def test_smaller_eps_after_parking_settles_the_parked_rows():
    V, W, c, directed = 4096, 1500, 100, 1
    e1, e2 = churn_stream(V, W + 40 * c, 400, 44)
    g = orc.Graph(V, e1, e2, directed, W, c)
    e = eng.Engine(V, W, directed, c)
    e.set_renumbering(1, growth_pct=10, min_parked=16)
    e.load_window(*g.window_edges())
    slot = e.add_source(0)
    gid = e.add_source_group([0, 1, 2])
    e.init_solve(slot, 1e-6)
    e.group_init_solve(gid, 1e-6)
    for k in range(30):
        assert not g.stream_updates()
        g.inc_construct(1)
        e.set_batch(*g.batch())
        e.slide(*g.new_stream())
        e.update(slot, 1e-6)
        e.group_update(gid, 1e-6)
    assert e.id_space()["parked"] > 0
    p, r = e.read(slot)
    big = np.abs(r) > 1e-9
    assert big.any()  # (otherwise the test shows nothing)
    # a frontier scan with the smaller eps sees the parked rows too
    ids = e.inspect(slot, 0, 1e-9)
    assert set(ids.tolist()) == set(np.nonzero(r > 1e-9)[0].tolist())
    assert not g.stream_updates()
    g.inc_construct(1)
    e.set_batch(*g.batch())
    e.slide(*g.new_stream())
    e.update(slot, 1e-9)
    e.group_update(gid, 1e-9)
    src, dst = window_directed_edges(g)
    p, r = e.read(slot)
    assert np.max(np.abs(r)) < 1e-9 and invariant_max_err_np(p, r, src, dst, V, 0) < 1e-13
    for i, s in enumerate([0, 1, 2]):
        p, r = e.group_read(gid, i)
        assert np.max(np.abs(r)) < 1e-9 and invariant_max_err_np(p, r, src, dst, V, s) < 1e-13


@pytest.mark.parametrize("seed", list(range(1, 9)))
def test_randomised_churn_with_random_staging(seed):
    """Random stream shape, thresholds, source mix and staging pattern (in step / a few epochs ahead), synchronous
    schedule: every state equals the oracle's whenever it has caught up."""
    rng = np.random.default_rng(1000 + seed)
    V = int(rng.choice([1024, 4096, 8192]))
    W = int(rng.integers(300, 1500))
    c = int(rng.integers(20, max(21, W // 8)))
    directed = int(rng.integers(0, 2))
    band = int(rng.integers(60, 400))
    ahead_max = int(rng.integers(1, 6))
    batches = 50
    e1, e2 = churn_stream(V, W + batches * c, band, 7000 + seed, back=float(rng.uniform(0.0, 0.15)), hubs=int(rng.integers(2, 8)))
    eps = float(rng.choice([1e-9, 1e-7]))
    nsrc = int(rng.integers(2, 12))
    sources = [int(x) for x in rng.choice(np.unique(np.concatenate([e1[:W], e2[:W]])), nsrc, replace=False)]
    g = orc.Graph(V, e1, e2, directed, W, c)
    states = [orc.State(V, s, eps) for s in sources]
    e = eng.Engine(V, W, directed, c, n_epochs=ahead_max + 2, schedule=eng.SCHEDULE_SYNC)
    e.set_renumbering(1, growth_pct=int(rng.integers(3, 20)), min_parked=int(rng.integers(1, 32)))
    e.load_window(*g.window_edges())
    slot = e.add_source(sources[0])
    gid = e.add_source_group(sources)
    for s in states:
        s.sync_execute(g)
    e.init_solve(slot, eps)
    e.group_init_solve(gid, eps)
    done = 0
    while done < batches:
        n = int(min(batches - done, rng.integers(1, ahead_max + 1)))
        staged = []
        for _ in range(n):
            assert not g.stream_updates()
            g.inc_construct(1)
            e.set_batch(*g.batch())
            staged.append(e.slide(*g.new_stream()))
            for s in states:
                s.sync_inc_execute(g)
        for ep in staged:
            e.update(slot, eps, epoch=ep)
            e.group_update(gid, eps, epoch=ep)
        done += n
        p, r = e.read(slot)
        assert np.max(np.abs(p - states[0].p)) < SYNC_TOL and np.max(np.abs(r - states[0].r)) < SYNC_TOL, (seed, done)
        for i, s in enumerate(states):
            p, r = e.group_read(gid, i)
            assert np.max(np.abs(p - s.p)) < SYNC_TOL and np.max(np.abs(r - s.r)) < SYNC_TOL, (seed, done, i)
    sp = e.id_space()
    print("id space", seed, sp)
    assert sp["ids"] + sp["parked"] <= V


@pytest.mark.parametrize("directed", [1, 0])
def test_every_vertex_has_an_id_and_the_two_zones_touch(directed):
    """A small id range that the stream exhausts: live zone and parked zone fill the whole capacity, a revived vertex
    takes exactly the slot the parked zone gives up."""
    V, W, c, eps, batches = 96, 60, 6, 1e-9, 120
    rng = np.random.default_rng(77 + directed)
    n = W + batches * c
    centre = (np.arange(n) // 3) % V  # the active band wanders around the whole range, several times
    e1 = ((centre + rng.integers(0, 24, n)) % V).astype(np.int32)
    e2 = ((centre + rng.integers(0, 24, n)) % V).astype(np.int32)
    e2[e1 == e2] = (e2[e1 == e2] + 1) % V
    g = orc.Graph(V, e1, e2, directed, W, c)
    sources = [int(e1[0]), int(e2[0]), 5]
    states = [orc.State(V, s, eps) for s in sources]
    e = eng.Engine(V, W, directed, c, schedule=eng.SCHEDULE_SYNC)
    e.set_renumbering(1, growth_pct=2, min_parked=1)
    e.load_window(*g.window_edges())
    slot = e.add_source(sources[0])
    gid = e.add_source_group(sources)
    for s in states:
        s.sync_execute(g)
    e.init_solve(slot, eps)
    e.group_init_solve(gid, eps)
    touched = False
    for k in range(batches):
        assert not g.stream_updates()
        g.inc_construct(1)
        e.set_batch(*g.batch())
        e.slide(*g.new_stream())
        for s in states:
            s.sync_inc_execute(g)
        e.update(slot, eps)
        e.group_update(gid, eps)
        sp = e.id_space()
        assert sp["ids"] + sp["parked"] <= V
        touched |= sp["ids"] + sp["parked"] == V
        p, r = e.read(slot)
        assert np.max(np.abs(p - states[0].p)) < SYNC_TOL and np.max(np.abs(r - states[0].r)) < SYNC_TOL, k
        for i, s in enumerate(states):
            p, r = e.group_read(gid, i)
            assert np.max(np.abs(p - s.p)) < SYNC_TOL and np.max(np.abs(r - s.r)) < SYNC_TOL, (k, i)
    sp = e.id_space()
    assert touched and sp["renumberings"] >= 3 and sp["revivals"] > 10, sp


@pytest.mark.parametrize("form", ["multi-sweep", "one-sweep"])
@pytest.mark.parametrize("seed", [2, 3])
def test_production_mode_soak_against_cilk_oracle(seed, form):
    """tools/soak.py, shortened: eager schedule, default launch forms, renumbering at work, one slot and a 10-source group
    over 120 in-step batches of a medium-size stream (the group's loops as multi-sweep launches, or one launch per
    sweep with the tail as pushes); |p - p_cpu| < 1e-9, |r| < eps and the invariant every 10 batches."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), str(seed), "120", "15"] + (["one-sweep"] if form == "one-sweep" else []), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "120 batches ok" in r.stdout, r.stdout[-2000:]
