"""The untimed region of batch k + 1 beside the timed region of batch k: dppr_set_batch + dppr_slide_concurrent
from a BUILDER thread while the SOLVER thread runs dppr_update / dppr_group_update on the explicitly named epoch k. Two resident
epochs; the builder has its own HIP stream and scratch inside the engine. Bars: the device CSR of every epoch built that way is the
oracle's bit for bit, p / r after every batch are the synchronous oracle's to rounding -- on a stream whose vertices come and go
(fresh ids in most batches, parked vertices that come back: their state rows move while the solver runs), with the id-space
renumbering taking its exclusive slides in between (dppr_renumbering_due)."""
import threading

import numpy as np
import pytest

from dynamicppr_amd import engine as eng
from oracle import oracle as orc
from tests.test_renumbering_gpu import churn_stream
from tests.util import sorted_csr

pytestmark = pytest.mark.gpu
SYNC_TOL = 1e-14


def staged_batches(V, e1, e2, directed, W, c, n):
    """Batch k = 1..n of the stream as the host hands them over, and the window graph after each (the oracle's)."""
    g = orc.Graph(V, e1, e2, directed, W, c)
    out = []
    for _ in range(n):
        assert not g.stream_updates()
        g.inc_construct(1)
        b1, b2, ins = (x.copy() for x in g.batch())
        n1, n2 = (x.copy() for x in g.new_stream())
        row_i, col_i = g.flatten(1)
        row_o, col_o = g.flatten(0)
        out.append(dict(b=(b1, b2, ins), n=(n1, n2), inc=(row_i.copy(), sorted_csr(row_i, col_i)), out=(row_o.copy(), sorted_csr(row_o, col_o)),
                        deg=np.array(g.deg()).copy()))
    return out


def check_epoch_graph(e, epoch, want):
    row, col, deg = e.read_graph(epoch)
    assert np.array_equal(row, want["inc"][0]) and np.array_equal(col[:len(want["inc"][1])], want["inc"][1])
    assert np.array_equal(deg, want["deg"])
    row, col = e.read_out_graph(epoch)
    assert np.array_equal(row, want["out"][0]) and np.array_equal(col[:len(want["out"][1])], want["out"][1])


@pytest.mark.parametrize("directed", [1, 0])
@pytest.mark.parametrize("mode", ["single", "group", "single-push"])
def test_next_graph_is_built_while_this_batch_is_solved(directed, mode):
    V, n_stream, batches = 4096, 9000, 40
    e1, e2 = churn_stream(V, n_stream, 400, 5)
    W, c, eps = 600, 60, 1e-9
    stage = staged_batches(V, e1, e2, directed, W, c, batches)
    g = orc.Graph(V, e1, e2, directed, W, c)
    sources = [0, 1, 2] if mode == "group" else [0]
    states = [orc.State(V, s, eps) for s in sources]
    tune = dict(pull_min_frontier=-1) if mode == "single-push" else {}
    e = eng.Engine(V, W, directed, c, n_epochs=2, schedule=eng.SCHEDULE_SYNC, **tune)
    e.set_renumbering(1, growth_pct=8, min_parked=8)
    e.load_window(*g.window_edges())
    handle = e.add_source_group(sources) if mode == "group" else e.add_source(sources[0])
    for s in states:
        s.sync_execute(g)
    (e.group_init_solve if mode == "group" else e.init_solve)(handle, eps)

    def build(k, concurrent, box):
        e.set_batch(*stage[k - 1]["b"])
        box.append(e.slide(*stage[k - 1]["n"], concurrent=concurrent))

    box = []
    build(1, False, box)
    epoch = box[0]
    check_epoch_graph(e, epoch, stage[0])
    beside = exclusive = 0
    for k in range(1, batches + 1):
        th, box = None, []
        if k < batches and not e.renumbering_due():
            th = threading.Thread(target=build, args=(k + 1, True, box))
            th.start()                                                    # batch k + 1's graph: built WHILE batch k is solved
        (e.group_update if mode == "group" else e.update)(handle, eps, epoch=epoch)
        assert not g.stream_updates()
        g.inc_construct(1)
        for i, s in enumerate(states):
            s.sync_inc_execute(g)
            p, r = e.group_read(handle, i) if mode == "group" else e.read(handle)
            assert np.max(np.abs(p - s.p)) < SYNC_TOL and np.max(np.abs(r - s.r)) < SYNC_TOL, (k, i)
        if th is not None:
            th.join()
            beside += 1
        elif k < batches:
            build(k + 1, False, box)                                      # an exclusive slide: the id space is renumbered here
            exclusive += 1
        if k < batches:
            epoch = box[0]
            check_epoch_graph(e, epoch, stage[k])
    ids = e.id_space()
    assert beside >= batches // 2 and exclusive >= 1 and ids["renumberings"] >= 1 and ids["revivals"] >= 1, (beside, exclusive, ids)
    e.close()


def test_slide_concurrent_needs_two_epochs():
    V, e1, e2 = 64, np.arange(40, dtype=np.int32) % 64, (np.arange(40, dtype=np.int32) * 7 + 1) % 64
    e = eng.Engine(V, 20, 1, 2)
    e.load_window(e1[:20], e2[:20])
    e.set_batch(np.concatenate([e1[:2], e1[20:22]]), np.concatenate([e2[:2], e2[20:22]]), np.array([0, 0, 1, 1], np.uint8))
    with pytest.raises(eng.DpprError):
        e.slide(e1[20:22], e2[20:22], concurrent=True)
    e.slide(e1[20:22], e2[20:22])          # the exclusive form still takes the staged batch
    e.close()


@pytest.mark.parametrize("mode", ["single", "group"])
def test_reads_beside_a_concurrent_slide_see_one_state_of_the_id_space(mode):
    """dppr_read / dppr_group_read on the solver thread while the builder thread runs dppr_set_batch +
    dppr_slide_concurrent. A graph update changes no solver state, so -- in the caller's ids -- p and r read DURING it must be what
    they were before it, also for a parked vertex that the batch revives (its id changes first, its rows move later: the engine
    holds its map lock from the first id it assigns to the end of the row moves, and a read holds it from its copy of the map to
    the end of its gathers)."""
    V, n_stream, batches = 4096, 9000, 40
    e1, e2 = churn_stream(V, n_stream, 400, 5)
    W, c, eps, directed = 600, 60, 1e-9, 1
    stage = staged_batches(V, e1, e2, directed, W, c, batches)
    g = orc.Graph(V, e1, e2, directed, W, c)
    sources = [0, 1, 2] if mode == "group" else [0]
    e = eng.Engine(V, W, directed, c, n_epochs=2, schedule=eng.SCHEDULE_SYNC)
    e.set_renumbering(1, growth_pct=8, min_parked=8)
    e.load_window(*g.window_edges())
    handle = e.add_source_group(sources) if mode == "group" else e.add_source(sources[0])
    (e.group_init_solve if mode == "group" else e.init_solve)(handle, eps)
    read = (lambda: e.group_read(handle, len(sources) - 1)) if mode == "group" else (lambda: e.read(handle))

    def build(k, concurrent, box):
        e.set_batch(*stage[k - 1]["b"])
        box.append(e.slide(*stage[k - 1]["n"], concurrent=concurrent))

    box = []
    build(1, False, box)
    epoch = box[0]
    reads_beside = revived_beside = 0
    for k in range(1, batches):
        (e.group_update if mode == "group" else e.update)(handle, eps, epoch=epoch)
        p0, r0 = read()
        box = []
        if e.renumbering_due():
            build(k + 1, False, box)               # an exclusive slide: the id space is renumbered (vertices get parked)
        else:
            before = e.id_space()["revivals"]
            th = threading.Thread(target=build, args=(k + 1, True, box))
            th.start()
            while True:                            # reads for as long as the builder runs, and one after it
                alive = th.is_alive()
                p, r = read()
                assert np.array_equal(p, p0) and np.array_equal(r, r0), k
                reads_beside += 1
                if not alive:
                    break
            th.join()
            revived_beside += e.id_space()["revivals"] - before
        epoch = box[0]
    ids = e.id_space()
    assert reads_beside >= batches and revived_beside >= 1 and ids["renumberings"] >= 1, (reads_beside, revived_beside, ids)
    e.close()
