import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from dynamicppr_amd import datagen, engine as eng, stream as st
from oracle import oracle as orc
from tests.util import Scenario
worst = 0
for seed in range(8):
    V, e1, e2 = datagen.rmat_stream(10, 20000, 300 + seed)
    directed = seed % 2
    W, c = 2000, 40
    src = int(datagen.top_sources(V, e1, e2, W, directed, 1)[0])
    sc = Scenario(V, e1, e2, directed, W, c, src, 1e-9)
    sc.s.cilk_execute(sc.g); sc.e.init_solve(sc.slot, 1e-9)
    for k in range(10):
        if k:
            sc.advance_graphs(); sc.s.cilk_inc_execute(sc.g); sc.e.update(sc.slot, 1e-9)
        p, r = sc.e.read(sc.slot)
        worst = max(worst, np.max(np.abs(p - sc.s.p)))
print("small graphs worst |p_gpu - p_cilk|:", worst)
V, e1, e2, cfg = datagen.stand_in_stream("youtube", "/tmp/dppr_data")
wl = st.workload_config(len(e1), 0.1, 0, 0.01, 100)
src = int(datagen.top_sources(V, e1, e2, wl.window, 0, 1)[0])
sc = Scenario(V, e1, e2, 0, wl.window, wl.per_batch, src, 1e-9)
sc.s.cilk_execute(sc.g); sc.e.init_solve(sc.slot, 1e-9)
w2 = 0
for k in range(6):
    if k:
        sc.advance_graphs(); sc.s.cilk_inc_execute(sc.g); sc.e.update(sc.slot, 1e-9)
    p, r = sc.e.read(sc.slot)
    w2 = max(w2, np.max(np.abs(p - sc.s.p)))
print("youtube stand-in worst |p_gpu - p_cilk|:", w2, " max|r|:", np.max(np.abs(r)))
