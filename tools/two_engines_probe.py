#!/usr/bin/env python3
"""Probe: two engines on ONE device, driven from two host threads, both with windows that want all
256 CUs for a resident launch (configs[1] stand-in). Prints per-engine time, resident launches,
roll-call give-ups and the loop invariant."""
import os, sys, threading, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dynamicppr_amd import datagen, engine as eng, stream as st
from util import invariant_max_err_np

V, e1, e2, cfg = datagen.stand_in_stream("youtube", "/tmp/dppr_data")
wl = st.workload_config(len(e1), 0.1, 0, 0.01, 100)
srcs = [int(x) for x in datagen.top_sources(V, e1, e2, wl.window, 0, 2)]
out = {}

def drive(i):
    e = eng.Engine(V, wl.window, 0, wl.per_batch)
    ss = st.SlidingStream(V, e1, e2, 0, wl)
    e.load_window(*ss.serialize_edge_stream()); slot = e.add_source(srcs[i]); e.init_solve(slot, 1e-9)
    t = 0.0
    for b in range(30):
        ss.stream_updates(); e.set_batch(*ss.batch_arrays()); e.slide(*ss.new_arrays())
        t0 = time.perf_counter(); e.update(slot, 1e-9); t += time.perf_counter() - t0
    p, r = e.read(slot)
    w1, w2 = ss.serialize_edge_stream()
    s_, d_ = np.concatenate([w1, w2]), np.concatenate([w2, w1])
    stt = e.stats(slot)
    out[i] = dict(ms_per_update=1e3 * t / 30, resident=stt["persist_launches"], gave_up=stt["persist_aborts"],
                  invariant=invariant_max_err_np(p, r, s_, d_, V, srcs[i]), max_r=float(np.max(np.abs(r))))

ths = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
for i in sorted(out): print(i, out[i])
