#!/bin/bash
# Diagnostic: per-workgroup timeline of the last DENSE binned sweep (k_bin_scatter / k_bin_reduce). Builds a SEPARATE
# library with -DDPPR_STAMPS (never the product build). usage: tools/r03/stamps_bin.sh <config> [binned tune tuple]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics \
   -DDPPR_STAMPS $DPPR_STAMP_FLAGS -o /tmp/libdppr_hip_stamps.so dynamicppr_amd/csrc/dppr_engine.hip
export DPPR_STAMP_CONFIG=${1:-twitter} DPPR_STAMP_TUNE=${2:-1}
DPPR_LIB=/tmp/libdppr_hip_stamps.so python3 - <<'PY'
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from dynamicppr_amd import datagen, engine as eng, stream as st
key = os.environ["DPPR_STAMP_CONFIG"]
cfg = datagen.STAND_INS[key]
f = cfg.flags.split(); opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
wl = st.workload_config(cfg.edges, 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1.0)), int(opt.get("-b", 0)), int(opt.get("-c", 0)), int(opt.get("-l", 0)))
V, e1, e2, _ = datagen.stand_in_stream(key, "/tmp/dppr_data", limit=wl.window + 3 * wl.per_batch)
directed = cfg.directed
pick = datagen.top_sources(V, e1, e2, wl.window, directed, 10) if key != "friendster" else datagen.ranked_sources(V, e1, e2, wl.window, directed, 10, 1000, 10)
tune = tuple(int(x) for x in os.environ["DPPR_STAMP_TUNE"].split(","))
e = eng.Engine(V, wl.window, directed, wl.per_batch, binned=tune)
ss = st.SlidingStream(V, e1, e2, directed, wl)
e.load_window(*ss.serialize_edge_stream()); slot = e.add_source(int(pick[0])); e.init_solve(slot, 1e-9)
ss.stream_updates(); e.set_batch(*ss.batch_arrays()); e.slide(*ss.new_arrays()); ms = e.update(slot, 1e-9)
st_ = e.stats(slot)
print(f"{key} binned={tune}: batch {ms:.1f} ms, {st_['pull_iterations']} sweeps of which {st_['binned_sweeps']} binned")
L = eng.lib()
L.dppr_debug_bin_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int]
for which, name, stages in ((0, "k_bin_scatter", ["x slice -> LDS", "stream + stores"]), (1, "k_bin_reduce", ["rows -> LDS", "stream + LDS adds", "repair / snapshot"])):
    rows = 16384
    buf = np.zeros(rows * 6, dtype=np.uint64)
    assert L.dppr_debug_bin_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), which, rows) == 0
    s = buf.reshape(rows, 6).astype(np.int64)
    s = s[s[:, 0] > 0]
    n = len(s); end_col = len(stages)
    t0 = s[:, 0].min(); span = (s[:, end_col].max() - t0) / 100.0
    dur = (s[:, end_col] - s[:, 0]) / 100.0
    ent = s[:, 4]
    print(f"{name}: {n} workgroups, span {span:.1f} us, entries {ent.sum()}, busy sum {dur.sum():.0f} us -> {dur.sum() / span:.1f} workgroups busy on average")
    print(f"   workgroup time us: median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}; entries per workgroup: median {int(np.median(ent))} max {int(ent.max())}")
    for i, nm in enumerate(stages):
        d = (s[:, i + 1] - s[:, i]) / 100.0
        print(f"   {nm:22s} median {np.median(d):7.2f} us  p90 {np.percentile(d, 90):7.2f}  max {d.max():7.2f}")
    start = (s[:, 0] - t0) / 100.0; end = (s[:, end_col] - t0) / 100.0
    print("   started by (us) p50/p90/max:", f"{np.median(start):.0f} {np.percentile(start, 90):.0f} {start.max():.0f};", "busy workgroups at 25/50/75/90 % of the span:",
          [int(((start <= span * q) & (end > span * q)).sum()) for q in (0.25, 0.5, 0.75, 0.9)])
    big = np.argsort(-dur)[:5]
    print("   longest workgroups (entries, us, start us):", [(int(ent[i]), round(float(dur[i]), 1), round(float(start[i]), 1)) for i in big])
PY
