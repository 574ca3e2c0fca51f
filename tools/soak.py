#!/usr/bin/env python3
"""Long randomised soak at medium size, production mode (eager schedule, default launch forms, renumbering on): a single
source slot and a 10-source group over hundreds of in-step batches, every state compared with the oracle's restatement
of cpu/PPRCPUMTCilkRev (-t 1) every few batches (|p - p_cpu| < 1e-9, |r| < eps, loop invariant).
    python tools/soak.py [seed] [batches] [scale] [one-sweep]      (4th argument: the group's loops as one launch per sweep
    with the tail as pushes -- what large windows run -- instead of multi-sweep launches)
    DPPR_SOAK_MERGE=1: the merged loop (dppr_set_phase_merge, eps / 4) on both states; DPPR_SOAK_TUNE="key=int,...": engine tuning"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicppr_amd import datagen, engine as eng
from oracle import oracle as orc
from tests.util import invariant_max_err_np, window_directed_edges

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 300
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rng = np.random.default_rng(seed)
directed = int(rng.integers(0, 2))
W = int(rng.integers(20000, 60000)); c = int(rng.integers(100, 600)); eps = 1e-9
V, e1, e2 = datagen.rmat_stream(scale, W + (batches + 1) * c, 100 + seed)
srcs = [int(x) for x in datagen.top_sources(V, e1, e2, W, directed, 10)]
g = orc.Graph(V, e1, e2, directed, W, c)
states = [orc.State(V, s, eps) for s in srcs]
tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("DPPR_SOAK_TUNE", "").split(",") if kv}
merge = int(os.environ.get("DPPR_SOAK_MERGE", "0"))
e = eng.Engine(V, W, directed, c, **tune)
if merge:
    e.set_phase_merge(True, 4)
e.set_renumbering(1, growth_pct=5, min_parked=64)
if len(sys.argv) > 4:
    e.set_group_resident(False)
e.load_window(*g.window_edges())
slot = e.add_source(srcs[0]); gid = e.add_source_group(srcs)
for s in states: s.cilk_execute(g)
e.init_solve(slot, eps); e.group_init_solve(gid, eps)
t0 = time.time(); worst = 0.0
for k in range(1, batches + 1):
    assert not g.stream_updates()
    g.inc_construct(1)
    e.set_batch(*g.batch()); e.slide(*g.new_stream())
    for s in states: s.cilk_inc_execute(g)
    e.update(slot, eps); e.group_update(gid, eps)
    if k % 10 == 0 or k == batches:
        src, dst = window_directed_edges(g)
        for i, s in enumerate(states):
            for p, r in ([e.read(slot)] if i == 0 else []) + [e.group_read(gid, i)]:
                dp = float(np.max(np.abs(p - s.p))); worst = max(worst, dp)
                assert dp < 1e-9 and np.max(np.abs(r)) <= (eps / 4 if merge else eps) and invariant_max_err_np(p, r, src, dst, V, srcs[i]) < 1e-13, (k, i, dp)
st = e.stats(slot); sp = e.id_space()
print(f"seed {seed}{' merged loop' if merge else ''} {tune or ''}: directed {directed} V {V} W {W} c {c}: {batches} batches ok, max |p - p_cpu| {worst:.3e}, id space {sp}, "
      f"resident launches {st['persist_launches']} aborts {st['persist_aborts']}, {time.time() - t0:.0f} s")
