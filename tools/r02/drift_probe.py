#!/usr/bin/env python3
"""In-step run (slide -> update per batch) of one stand-in: per block of batches the mean update time, the swept id
space, iterations and pushed edges per batch -- separates workload drift from id-space growth -- and what the slides
cost on the host clock (a renumbering shows up as the block's slowest slide).
    python tools/r02/drift_probe.py [config] [batches] [block] [sources]          (env DPPR_RENUMBER=0|1)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicppr_amd import datagen, engine as eng, stream as st

key = sys.argv[1] if len(sys.argv) > 1 else "youtube"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 400
blk = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cfg = datagen.STAND_INS[key]
f = cfg.flags.split(); opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
wl = st.workload_config(cfg.edges, 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1.0)), int(opt.get("-b", 0)), int(opt.get("-c", 0)), int(opt.get("-l", 0)))
V, e1, e2, _ = datagen.stand_in_stream(key, "/tmp/dppr_data", limit=wl.window + (B + 1) * wl.per_batch)
nsrc = int(sys.argv[4]) if len(sys.argv) > 4 else 1
e = eng.Engine(V, wl.window, cfg.directed, wl.per_batch)
ss = st.SlidingStream(V, e1, e2, cfg.directed, wl)
e.load_window(*ss.serialize_edge_stream())
if nsrc == 1:
    slot = e.add_source(int(datagen.top_sources(V, e1, e2, wl.window, cfg.directed, 1)[0]))
    e.init_solve(slot, 1e-9)
    update, stats = (lambda: e.update(slot, 1e-9)), (lambda: e.stats(slot))
else:
    gid = e.add_source_group([int(x) for x in datagen.ranked_sources(V, e1, e2, wl.window, cfg.directed, 10, 1000, nsrc)])
    e.group_init_solve(gid, 1e-9)
    update, stats = (lambda: e.group_update(gid, 1e-9)), (lambda: e.group_stats(gid))
ms, sl, last = [], [], stats()
for b in range(1, B + 1):
    ss.stream_updates()
    t0 = time.perf_counter()
    e.set_batch(*ss.batch_arrays()); e.slide(*ss.new_arrays())
    sl.append((time.perf_counter() - t0) * 1e3)
    ms.append(update())
    if b % blk == 0:
        s = stats(); sp = e.id_space()
        print(f"batches {b-blk+1:4d}..{b:4d}: update {np.mean(ms[-blk:]):.4f} ms  set_batch+slide mean {np.mean(sl[-blk:]):.2f} max {np.max(sl[-blk:]):.1f} ms"
              f"  ids {sp['ids']:8d} parked {sp['parked']:7d} renumberings {sp['renumberings']:2d}"
              f"  iters/batch {(s['iterations']-last['iterations'])/blk:6.1f}  E/batch {(s['sum_E']-last['sum_E'])/blk/1e6:7.2f} M"
              f"  resident launches {s['persist_launches']-last['persist_launches']}", flush=True)
        last = s
