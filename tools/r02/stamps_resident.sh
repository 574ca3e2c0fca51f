#!/bin/bash
# Diagnostic: stage times of ONE iteration (the 11th) of the resident sweep, per workgroup.
# Builds a SEPARATE library with -DDPPR_STAMPS (never the product build).
set -e
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics \
   -DDPPR_STAMPS -o /tmp/libdppr_hip_stamps.so dynamicppr_amd/csrc/dppr_engine.hip
DPPR_LIB=/tmp/libdppr_hip_stamps.so python3 - <<'PY'
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from dynamicppr_amd import datagen, engine as eng, stream as st
V, e1, e2, cfg = datagen.stand_in_stream("youtube", "/tmp/dppr_data")
wl = st.workload_config(len(e1), 0.1, 0, 0.01, 100)
src = int(datagen.top_sources(V, e1, e2, wl.window, 0, 1)[0])
e = eng.Engine(V, wl.window, 0, wl.per_batch)
ss = st.SlidingStream(V, e1, e2, 0, wl)
e.load_window(*ss.serialize_edge_stream()); slot = e.add_source(src); e.init_solve(slot, 1e-9)
L = eng.lib()
L.dppr_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
rows = 256
for b in range(3):
    ss.stream_updates(); e.set_batch(*ss.batch_arrays()); e.slide(*ss.new_arrays()); e.update(slot, 1e-9)
buf = np.zeros(rows * 8, dtype=np.uint64)
assert L.dppr_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), rows) == 0
s = buf.reshape(rows, 8).astype(np.int64)
s = s[s[:, 0] > 0]
print("workgroups with stamps:", len(s))
names = ["gathers (incl. repeats, arrivals poll)", "LDS adds", "sync (all adds in)", "finish + x store issued", "sync (counts)"]
t0 = s[:, 0].min()
print("iteration start spread (cycles): median", int(np.median(s[:, 0] - t0)), "max", int((s[:, 0] - t0).max()))
for i, n in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print(f"{n:34s} median {int(np.median(d)):7d} cyc   p90 {int(np.percentile(d, 90)):7d}   max {int(d.max()):7d}")
fa = s[:, 6] - s[:, 0]
print(f"last wave: first gather attempt back after median {int(np.median(fa))} cyc (p90 {int(np.percentile(fa, 90))}), repeat rounds median {int(np.median(s[:, 7]))} p90 {int(np.percentile(s[:, 7], 90))} max {int(s[:, 7].max())}; workgroups without a repeat: {int((s[:, 7] == 0).sum())}")
tot = s[:, 5] - s[:, 0]
print("iteration median", int(np.median(tot)), "max", int(tot.max()), "| span first start -> last end", int(s[:, 5].max() - t0))
print(e.stats(slot))
PY
