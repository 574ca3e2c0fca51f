#!/usr/bin/env python3
"""What a window slide costs (the graph update the reference leaves out of ppr_time, gpu/PPRGPU.cuh:114-135, and rebuilds
from scratch every batch, gpu/SlidingGraphBuilder.cuh:203-221): wall time of dppr_set_batch + dppr_slide on a stand-in,
with the incremental merge of the sorted keys (f1) against the full re-sort, with and without the binned-sweep tables,
the phases of a slide (DPPR_SLIDE_TRACE: each mark synchronises, so the phase sum is an upper bound of the wall time)
and the split of a renumbering slide (DPPR_RENUMBER_TRACE).

    python tools/r04/slide_costs.py <stand-in> [out.jsonl]      # one JSON line per variant
"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(key, incremental, binned, renumber, lookahead=0):
    import numpy as np
    from dynamicppr_amd import datagen, engine as eng, stream as st
    cfg = datagen.STAND_INS[key]
    f = cfg.flags.split()
    opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
    wl = st.workload_config(cfg.edges, 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1)), int(opt.get("-b", 0)),
                            int(opt.get("-c", 0)), int(opt.get("-l", 0)))
    n_slides = 9
    V, e1, e2, _ = datagen.stand_in_stream(key, "/tmp/dppr_data", limit=wl.window + (n_slides + 1) * wl.per_batch)
    e = eng.Engine(V, wl.window, cfg.directed, wl.per_batch, binned=binned)
    e.set_incremental_graph(bool(incremental))
    e.set_renumbering(1 if renumber else 0, growth_pct=1 if renumber else 0, min_parked=1 if renumber else 0)
    ss = st.SlidingStream(V, e1, e2, cfg.directed, wl)
    t = time.perf_counter()
    e.load_window(*ss.serialize_edge_stream())
    load_ms = 1e3 * (time.perf_counter() - t)
    pick = datagen.top_sources(V, e1, e2, wl.window, cfg.directed, 10) if key != "friendster" else datagen.ranked_sources(V, e1, e2, wl.window, cfg.directed, 10, 1000, 10)
    slot = e.add_source(int(pick[0]))
    e.init_solve(slot, 1e-9)
    rows = []
    staged = None
    for k in range(n_slides):
        if staged is None:
            ss.stream_updates()
            b, n = ss.batch_arrays(), ss.new_arrays()
        else:
            b, n = staged
        before = e.id_space()["renumberings"]
        print(f"[probe] slide {k} begins", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        e.set_batch(*b)
        t1 = time.perf_counter()
        e.slide(*n)
        t2 = time.perf_counter()
        rows.append({"k": k, "set_batch_ms": 1e3 * (t1 - t0), "slide_ms": 1e3 * (t2 - t1), "renumbered": e.id_space()["renumberings"] > before})
        if lookahead and k + 1 < n_slides:   # dppr_hint_next_batch: the next batch's id lookups run during this update
            t3 = time.perf_counter()
            ss.stream_updates()
            nb, nn = ss.batch_arrays(), ss.new_arrays()
            h = e.hint_next_batch(nb[0], nb[1], nn[0], nn[1])
            staged = ((h[0], h[1], nb[2]), (h[2], h[3]))
            rows[-1]["hint_ms"] = 1e3 * (time.perf_counter() - t3)
        t4 = time.perf_counter()
        upd = e.update(slot, 1e-9)        # (a renumbering needs every source converged on the newest epoch)
        rows[-1]["update_ms"] = upd
        rows[-1]["update_wall_ms"] = 1e3 * (time.perf_counter() - t4)
    print([l for l in e.debug_dump().splitlines() if "lookahead" in l], file=sys.stderr, flush=True)
    print("PROBE " + json.dumps({"V": V, "window": wl.window, "c": wl.per_batch, "ids": e.id_space()["ids"], "load_window_ms": load_ms, "slides": rows}), flush=True)


def run(key, incremental, binned, renumber, lookahead=0):
    env = dict(os.environ, DPPR_SLIDE_TRACE="1", DPPR_RENUMBER_TRACE="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", key, str(incremental), str(binned), str(renumber), str(lookahead)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    for l in r.stderr.splitlines():
        if "lookahead" in l:
            print(l, flush=True)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("PROBE ")][-1][6:])
    # phases per slide from the trace
    per, cur = [], None
    for line in r.stderr.splitlines():
        if line.startswith("[probe] slide"):
            cur = {}
            per.append(cur)
        m = re.match(r"\[(slide|renumber)\] (.+?)\s+([0-9.]+) (us|ms)$", line)
        if m and cur is not None:
            cur[("renumber: " if m.group(1) == "renumber" else "") + m.group(2).strip()] = cur.get(m.group(2).strip(), 0.0) + float(m.group(3)) * (1e-3 if m.group(4) == "us" else 1.0)
    for row, ph in zip(d["slides"], per):
        row["phases_ms"] = {k: round(v, 3) for k, v in ph.items()}
    return d


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else None


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]) if len(sys.argv) > 6 else 0)
        sys.exit(0)
    key = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"r03_slide_{key}.jsonl")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    big = key in ("twitter", "friendster")
    variants = [("incremental merge", 1, 0, 0, 0), ("full re-sort", 0, 0, 0, 0)]
    if big:
        variants += [("incremental merge + binned-sweep tables", 1, 1, 0, 0)]
    variants += [("incremental merge, renumbering forced", 1, 1 if big else 0, 1, 0)]
    variants += [("incremental merge" + (" + binned-sweep tables" if big else "") + ", id lookahead (dppr_hint_next_batch)", 1, 1 if big else 0, 0, 1)]
    if os.environ.get("SLIDE_COSTS_ONLY"):
        variants = [v for v in variants if os.environ["SLIDE_COSTS_ONLY"] in v[0]]
    with open(out, "w") as f:
        for name, inc, binned, ren, look in variants:
            d = run(key, inc, binned, ren, look)
            plain = [s for s in d["slides"][2:] if not s["renumbered"]]
            ren_rows = [s for s in d["slides"] if s["renumbered"]]
            line = {"config": key, "variant": name, "V": d["V"], "ids": d["ids"], "window": d["window"], "c": d["c"],
                    "load_window_ms": round(d["load_window_ms"], 1),
                    "set_batch_ms": round(median([s["set_batch_ms"] for s in plain]) or 0, 3),
                    "slide_ms": round(median([s["slide_ms"] for s in plain]) or 0, 3),
                    "update_ms": round(median([s["update_ms"] for s in plain]) or 0, 3),
                    "update_wall_ms": round(median([s["update_wall_ms"] for s in plain]) or 0, 3),
                    "hint_ms": round(median([s.get("hint_ms", 0.0) for s in plain]) or 0, 3),
                    "slide_phases_ms": plain[len(plain) // 2]["phases_ms"] if plain else None}
            if ren:
                line["renumbering_slides"] = len(ren_rows)
                line["renumbering_slide_ms"] = round(median([s["slide_ms"] for s in ren_rows]) or 0, 3)
                line["renumbering_slide_phases_ms"] = ren_rows[len(ren_rows) // 2]["phases_ms"] if ren_rows else None
            f.write(json.dumps(line) + "\n")
            print(json.dumps(line), flush=True)
