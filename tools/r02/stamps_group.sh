#!/bin/bash
# Diagnostic: stage times of ONE sweep of k_gsweep per workgroup: the 13th of a multi-sweep launch on resident-size
# windows, the last dense one-sweep launch otherwise (there the last two stages are not stamped).
# Builds a SEPARATE library with -DDPPR_STAMPS (never the product build).
# usage: tools/r02/stamps_group.sh [config] [sources]
set -e
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics \
   -DDPPR_STAMPS $DPPR_STAMP_FLAGS -o /tmp/libdppr_hip_stamps.so dynamicppr_amd/csrc/dppr_engine.hip
DPPR_LIB=/tmp/libdppr_hip_stamps.so CFG=${1:-youtube} NSRC=${2:-8} python3 - <<'PY'
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from dynamicppr_amd import datagen, engine as eng, stream as st
key, nsrc = os.environ["CFG"], int(os.environ["NSRC"])
cfg = datagen.STAND_INS[key]
f = cfg.flags.split(); opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
wl = st.workload_config(cfg.edges, 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1.0)), int(opt.get("-b", 0)), int(opt.get("-c", 0)), int(opt.get("-l", 0)))
V, e1, e2, _ = datagen.stand_in_stream(key, "/tmp/dppr_data", limit=wl.window + 8 * wl.per_batch)
srcs = [int(x) for x in (datagen.ranked_sources(V, e1, e2, wl.window, cfg.directed, 10, 1000, max(nsrc, 10))[:nsrc] if nsrc > 8 else datagen.top_sources(V, e1, e2, wl.window, cfg.directed, nsrc))]
e = eng.Engine(V, wl.window, cfg.directed, wl.per_batch)
ss = st.SlidingStream(V, e1, e2, cfg.directed, wl)
e.load_window(*ss.serialize_edge_stream()); gid = e.add_source_group(srcs); e.group_init_solve(gid, 1e-9)
L = eng.lib()
L.dppr_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
rows = 2048
for b in range(3):
    ss.stream_updates(); e.set_batch(*ss.batch_arrays()); e.slide(*ss.new_arrays()); e.group_update(gid, 1e-9)
buf = np.zeros(rows * 8, dtype=np.uint64)
assert L.dppr_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), rows) == 0
s = buf.reshape(rows, 8).astype(np.int64)
s = s[s[:, 0] > 0]
print("workgroups with stamps:", len(s))
names = ["bits/cols reload + edge phase", "barrier (all flushes in)", "vertex phase", "barrier", "act_out + count + drain stores", "arrive + wait for everybody"]
onesweep = e.group_stats(gid)["persist_launches"] == 0  # (then columns 5, 6 hold wall-clock entry / exit, not stages)
t0 = s[:, 0].min()
print("sweep start spread (cycles): median", int(np.median(s[:, 0] - t0)), "max", int((s[:, 0] - t0).max()))
for i, n in enumerate(names):
    if (s[:, i + 1] <= 0).any() or (onesweep and i >= 4): break
    d = s[:, i + 1] - s[:, i]
    print(f"{n:34s} median {int(np.median(d)):7d} cyc   p90 {int(np.percentile(d, 90)):7d}   max {int(d.max()):7d}")
if (s[:, 7] > 0).all() and len(s) > 1100:
    d1 = (s[:, 0] - s[:, 7])[1100:]
    print(f"{'  ... workgroups with ONE group':34s} median {int(np.median(d1)):7d} cyc   p90 {int(np.percentile(d1, 90)):7d}")
if (s[:, 7] > 0).all():
    d = s[:, 0] - s[:, 7]
    print(f"{'kernel entry -> first sweep start':34s} median {int(np.median(d)):7d} cyc   p90 {int(np.percentile(d, 90)):7d}   max {int(d.max()):7d}  (last group of the workgroup)")
last = 4 if onesweep else 6
tot = s[:, last] - s[:, 0]
print("sweep (group) median", int(np.median(tot)), "max", int(tot.max()))
if onesweep and (s[:, 5] > 0).all() and (s[:, 6] > 0).all():
    # one-sweep launch: entry / exit of every workgroup on the 100 MHz wall clock -> occupancy over the launch
    a, b = s[:, 5], s[:, 6]
    span = b.max() - a.min()
    life = b - a
    print("launch span", span / 100.0, "us; workgroup lifetime median", float(np.median(life)) / 100.0, "us; alive on average", round(float(life.sum()) / float(span), 1))
    ts = a.min() + (np.arange(25) + 0.5) * span / 25
    print("alive at 25 instants:", [int(((a <= t) & (b > t)).sum()) for t in ts])
    order = np.argsort(a)
    print("entry of workgroup #0/#511/#512/#1024/#1536/#2047 (us after first):", [float(a[order[k]] - a.min()) / 100.0 for k in (0, 511, 512, 1024, 1536, len(a) - 1)])
print(e.group_stats(gid))
PY
