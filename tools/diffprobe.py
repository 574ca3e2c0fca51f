import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from dynamicppr_amd import datagen, engine as eng
from tests.util import Scenario
V, e1, e2 = datagen.rmat_stream(9, 7118, 101)
W, c = 2047, 32
src = int(datagen.top_sources(V, e1, e2, W, 1, 3)[1])
worst = 0
for rep in range(30):
    sc = Scenario(V, e1, e2, 1, W, c, src, 1e-9, schedule=eng.SCHEDULE_SYNC, hub_min_degree=5, big_row_edges=11, pull_min_frontier=-1, chunk_iters=2, pull_block=1024)
    sc.s.sync_execute(sc.g); sc.e.init_solve(sc.slot, 1e-9)
    for k in range(4):
        if k:
            sc.advance_graphs(); sc.s.sync_inc_execute(sc.g); sc.e.update(sc.slot, 1e-9)
        p, r = sc.e.read(sc.slot)
        dp, dr = np.max(np.abs(p - sc.s.p)), np.max(np.abs(r - sc.s.r))
        worst = max(worst, dp, dr)
    st, want = sc.e.stats(sc.slot), sc.s.stats()
    assert (st["iterations"], st["sum_F"], st["sum_E"]) == (want["iters"], want["F"], want["E"]), (st, want)
print("worst |dp|,|dr| over 30 runs:", worst, "max p", sc.s.p.max())
