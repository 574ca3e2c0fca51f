#!/usr/bin/env python3
"""Host + device time of dppr_load_window (numbering, translation, first CSR build) and of one slide on a stand-in.
    python tools/load_probe.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicppr_amd import datagen, engine as eng, stream as st

key = sys.argv[1] if len(sys.argv) > 1 else "twitter"
cfg = datagen.STAND_INS[key]
f = cfg.flags.split(); opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
wl = st.workload_config(cfg.edges, 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1.0)), int(opt.get("-b", 0)), int(opt.get("-c", 0)), int(opt.get("-l", 0)))
t = time.perf_counter()
V, e1, e2, _ = datagen.stand_in_stream(key, "/tmp/dppr_data", limit=wl.window + 3 * wl.per_batch)
print(f"{key}: stream prefix ready in {time.perf_counter() - t:.1f} s (V {V}, window {wl.window}, batch {wl.per_batch})", flush=True)
e = eng.Engine(V, wl.window, cfg.directed, wl.per_batch)
ss = st.SlidingStream(V, e1, e2, cfg.directed, wl)
w = ss.serialize_edge_stream()
t = time.perf_counter(); e.load_window(*w); print(f"load_window {time.perf_counter() - t:.2f} s  ids {e.id_space()['ids']}", flush=True)
for k in range(2):
    ss.stream_updates()
    t = time.perf_counter(); e.set_batch(*ss.batch_arrays()); e.slide(*ss.new_arrays()); print(f"set_batch+slide {1e3 * (time.perf_counter() - t):.1f} ms", flush=True)
