#!/usr/bin/env python3
"""In-step run (slide -> update per batch) of one stand-in: per block of batches the mean update time, the swept id
space, iterations and pushed edges per batch -- separates workload drift from id-space growth.
    python tools/drift_probe.py [config] [batches] [block]          (env DPPR_RENUMBER=0|1)"""
import os, sys
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
src = int(datagen.top_sources(V, e1, e2, wl.window, cfg.directed, 1)[0])
e = eng.Engine(V, wl.window, cfg.directed, wl.per_batch)
ss = st.SlidingStream(V, e1, e2, cfg.directed, wl)
e.load_window(*ss.serialize_edge_stream())
slot = e.add_source(src)
e.init_solve(slot, 1e-9)
ms, last = [], e.stats(slot)
for b in range(1, B + 1):
    ss.stream_updates(); e.set_batch(*ss.batch_arrays()); e.slide(*ss.new_arrays())
    ms.append(e.update(slot, 1e-9))
    if b % blk == 0:
        s = e.stats(slot); sp = e.id_space()
        print(f"batches {b-blk+1:4d}..{b:4d}: {np.mean(ms[-blk:]):.4f} ms  ids {sp['ids']:8d} parked {sp['parked']:7d} renumberings {sp['renumberings']:2d}"
              f"  iters/batch {(s['iterations']-last['iterations'])/blk:6.1f}  E/batch {(s['sum_E']-last['sum_E'])/blk/1e6:7.2f} M"
              f"  resident launches {s['persist_launches']-last['persist_launches']}", flush=True)
        last = s
