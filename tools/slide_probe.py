#!/usr/bin/env python3
"""f1 measurement: wall time of dppr_set_batch + dppr_slide (the untimed graph update) with the
incremental merge vs the full re-sort, on a stand-in. Writes gpurun_out/slide_probe_<cfg>.json."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicppr_amd import datagen, engine as eng, stream as st

key = sys.argv[1] if len(sys.argv) > 1 else "youtube"
V, e1, e2, cfg = datagen.stand_in_stream(key, "/tmp/dppr_data")
f = cfg.flags.split()
opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
wl = st.workload_config(len(e1), 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1)), int(opt.get("-b", 0)),
                        int(opt.get("-c", 0)), int(opt.get("-l", 0)))
out = {"config": key, "V": V, "window": wl.window, "c": wl.per_batch}
for mode in (1, 0):
    e = eng.Engine(V, wl.window, cfg.directed, wl.per_batch)
    e.set_incremental_graph(bool(mode))
    ss = st.SlidingStream(V, e1, e2, cfg.directed, wl)
    e.load_window(*ss.serialize_edge_stream())
    ts = []
    for k in range(8):
        ss.stream_updates()
        b = ss.batch_arrays(); n = ss.new_arrays()
        t = time.perf_counter()
        e.set_batch(*b); e.slide(*n)
        ts.append(time.perf_counter() - t)
    out["incremental_ms" if mode else "full_resort_ms"] = round(1e3 * sorted(ts[2:])[len(ts[2:]) // 2], 3)
    e.close()
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/slide_probe_{key}.json", "w"))
