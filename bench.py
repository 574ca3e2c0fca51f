#!/usr/bin/env python3
"""Headline benchmark: sliding-window dynamic PPR update throughput on MI355X.

Metric (BASELINE.json): mean per-batch PPR update time (ms) + edge-updates/sec.
A "step" is one pass of the hot path over one batch: the reference's timed region
(gpu/PPRGPU.cuh:138-164) = IncrementalBatchUpdate + ExecuteMainLoop(0) +
ExecuteMainLoop(1). Batch upload and CSR rebuild are untimed there, so here all
W+K graph epochs are pre-staged in HBM (dppr_slide) before the timed region and
the K timed steps run back to back.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config youtube]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: independent source vertices shard one per GPU (BASELINE.json
north_star; no data-path collective), every rank streams the same batches over its
own replica of the window graph -> weak scaling; value = N * c * K / max-rank time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="youtube", help="stand-in name (dynamicppr_amd/datagen.py STAND_INS)")
    ap.add_argument("--eps", type=float, default=1e-9)
    ap.add_argument("--schedule", default="eager", choices=["eager", "sync"])
    ap.add_argument("--data-dir", default=os.environ.get("DPPR_DATA", "/tmp/dppr_data"))
    ap.add_argument("--bin", default=None, help="real reference .bin file to use instead of the stand-in")
    ap.add_argument("--directed", type=int, default=None)
    ap.add_argument("--cpu-batches", type=int, default=12, help="batches timed on the CPU oracle (bounded sample)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--group", type=int, default=1, metavar="S",
                    help="solve S (1..8) top-degree sources per GPU together as one source group (multi-source "
                         "batched sweeps, BASELINE.json configs 3 and 5 style); value counts c*K per source")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                    help="engine tuning knob (hub_min_degree, big_row_edges, pull_min_frontier, chunk_iters, pull_block)")
    return ap.parse_args()


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        a.gpus = world

    # torch first: its bundled HIP runtime and ours share one SONAME, the first one loaded wins
    import torch
    import torch.distributed as dist

    from dynamicppr_amd import datagen, engine as eng, shard, stream as st

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        # RCCL (backend "nccl") carries only the barrier and two scalar reductions; if it cannot come up
        # on this node the same three calls work over gloo -- the data path has no collective either way
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
        except Exception as exc:  # noqa: BLE001
            print(f"[rank {rank}] RCCL init failed ({exc}); falling back to gloo", file=sys.stderr, flush=True)
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo")

    # ---------------- workload (untimed) ----------------
    if a.bin:
        V, e1, e2 = datagen.read_bin(a.bin)
        directed = 1 if a.directed is None else a.directed
        name, flags = os.path.basename(a.bin), "-n 0 -r 0.01 -b 100"
    else:
        V, e1, e2, cfg = datagen.stand_in_stream(a.config, a.data_dir or None)
        directed = cfg.directed if a.directed is None else a.directed
        name, flags = f"{cfg.name} stand-in (R-MAT scale {cfg.scale}, seed {cfg.seed})", cfg.flags
    f = flags.split()
    opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
    cfg_type = int(opt.get("-n", 0))
    wl = st.workload_config(len(e1), 0.1, cfg_type, float(opt.get("-r", -1.0)), int(opt.get("-b", 0)),
                            int(opt.get("-c", 0)), int(opt.get("-l", 0)))
    W, c = wl.window, wl.per_batch
    n_steps = a.warmup + a.steps
    max_batches = (len(e1) - W) // max(c, 1)
    if n_steps > max_batches:
        sys.exit(f"stream too short for {n_steps} batches (max {max_batches})")
    sources = datagen.top_sources(V, e1, e2, W, directed, 10)
    source = shard.assign_sources(sources, rank, world, per_rank=1)[0]   # one top-10 source per GPU

    schedule = eng.SCHEDULE_EAGER if a.schedule == "eager" else eng.SCHEDULE_SYNC
    tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.tune}
    e = eng.Engine(V, W, directed, c, n_epochs=n_steps + 1, device=local_rank, schedule=schedule, **tune)
    ss = st.SlidingStream(V, e1, e2, directed, wl)
    w1, w2 = ss.serialize_edge_stream()
    e.load_window(w1, w2)
    if a.group > 1:
        return bench_group(a, e, ss, sources, rank, world, W, c, n_steps, V, e1, name, flags, directed, dist, shard, torch)
    slot = e.add_source(source)
    init_ms = e.init_solve(slot, a.eps)
    L = 0
    for _ in range(n_steps):                      # pre-stage every epoch in HBM
        assert not ss.stream_updates()
        b1, b2, ins = ss.batch_arrays()
        L = len(b1)
        e.set_batch(b1, b2, ins)
        n1, n2 = ss.new_arrays()
        e.slide(n1, n2)

    def device_sync():
        torch.cuda.synchronize()
        e.synchronize()

    # ---------------- warmup ----------------
    for k in range(1, a.warmup + 1):
        e.update(slot, a.eps, epoch=k)
    p0, r0 = e.read(slot)                         # state at the start of the timed region
    e.reset_stats(slot)

    # ---------------- timed region: exactly K steps, barrier + synchronize on both sides ----------------
    ev = [0.0]

    def run_steps():
        for k in range(a.warmup + 1, n_steps + 1):
            ev[0] += e.update(slot, a.eps, epoch=k)

    dt, _ = shard.timed_region(run_steps, device_sync, dist if world > 1 else None)
    ev_ms = ev[0]
    stats = e.stats(slot)
    units = shard.aggregate_units(c * a.steps, dist if world > 1 else None)

    # ---------------- roofline of the dominant kernel (profiled replay of the same K steps) ----------------
    roof = None
    cpu = None
    if rank == 0:
        e.write(slot, p0, r0)
        e.reset_stats(slot)
        e.set_profiling(True)
        # replay starts from the same converged state, so the batch-tail seeding is valid
        _force_converged(e, slot, a.eps)
        for k in range(a.warmup + 1, n_steps + 1):
            e.update(slot, a.eps, epoch=k)
        e.set_profiling(False)
        ps = e.stats(slot)
        push_bytes = 72 * ps["sum_F"] + 24 * ps["sum_E"] + 4 * ps["sum_N"]
        achieved = push_bytes / (ps["push_ms"] * 1e-3) / 1e9 if ps["push_ms"] > 0 else 0.0
        roof = {
            "bound": "hbm",
            "kernel": ("k_pull_resident (one launch = a run of frontier iterations, state kept on chip)"
                       if ps["persist_launches"] else "k_pull_iter / k_push_iter (one frontier iteration)"),
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
            "traffic": pmc_traffic_per_launch(bool(ps["persist_launches"])) if (a.config == "youtube" and not a.bin) else None,
            "iterations_per_launch": round(ps["iterations"] / max(ps["push_launches"], 1), 2),
            "launches": ps["push_launches"], "avg_launch_us": round(1e3 * ps["push_ms"] / max(ps["push_launches"], 1), 3),
            "algorithmic_bytes_per_launch": round(push_bytes / max(ps["push_launches"], 1), 1),
            "whole_batch_algorithmic_GBps": round(stats["algorithmic_bytes"] / (ev_ms * 1e-3) / 1e9, 2),
        }
        if not a.no_cpu_baseline and world == 1:   # the CPU leg runs on rank 0 at N=1 only
            cpu = cpu_baseline(V, e1, e2, directed, W, c, source, a.eps, a.cpu_batches)
            bin_path = a.bin or (datagen.stand_in_path(a.config, a.data_dir) if a.data_dir else None)
            cpu["reference_fifo"] = reference_fifo_baseline(bin_path, directed, flags, source, a.eps, c)

    if rank == 0:
        value = units / dt
        line = {
            "metric": "edge-updates/sec (ppr_throughput); ms_per_step = mean per-batch PPR update time",
            "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{name}, {'directed' if directed else 'undirected'}, -a 0 -y 1 -w 0.1 {flags} "
                                   f"-e {a.eps:g}, one top-10 source per GPU",
                       "V": V, "stream_edges": int(len(e1)), "window": W, "batch_c": c, "records_L": L,
                       "source": source, "schedule": a.schedule, "parallelism": f"sources x{world} (replicated graph)"},
            "event_ms_per_step": round(ev_ms / a.steps, 4), "init_solve_ms": round(init_ms, 3),
            "iterations_per_step": round(stats["iterations"] / a.steps, 2),
            "pull_iterations_per_step": round(stats["pull_iterations"] / a.steps, 2),
            "edges_pushed_per_step": round(stats["sum_E"] / a.steps, 1),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_group(a, e, ss, sources, rank, world, W, c, n_steps, V, e1, name, flags, directed, dist, shard, torch):
    """--group S: the S top-degree sources of this rank are solved together (dppr_group_*)."""
    S = a.group
    mine = [int(sources[(rank * S + k) % len(sources)]) for k in range(S)]
    gid = e.add_source_group(mine)
    init_ms = e.group_init_solve(gid, a.eps)
    L = 0
    for _ in range(n_steps):
        assert not ss.stream_updates()
        b1, b2, ins = ss.batch_arrays()
        L = len(b1)
        e.set_batch(b1, b2, ins)
        e.slide(*ss.new_arrays())

    def device_sync():
        torch.cuda.synchronize()
        e.synchronize()

    for k in range(1, a.warmup + 1):
        e.group_update(gid, a.eps, epoch=k)
    base = e.group_stats(gid)
    ev = [0.0]

    def run_steps():
        for k in range(a.warmup + 1, n_steps + 1):
            ev[0] += e.group_update(gid, a.eps, epoch=k)

    dt, _ = shard.timed_region(run_steps, device_sync, dist if world > 1 else None)
    st = e.group_stats(gid)
    units = shard.aggregate_units(S * c * a.steps, dist if world > 1 else None)
    if rank == 0:
        algo = st["algorithmic_bytes"] - base["algorithmic_bytes"]
        iters = st["iterations"] - base["iterations"]
        line = {
            "metric": "edge-updates/sec (ppr_throughput, summed over sources); ms_per_step = per-batch update time of the group",
            "value": round(units / dt, 1), "unit": "edges/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{name}, {'directed' if directed else 'undirected'}, -a 0 -y 1 -w 0.1 {flags} "
                                   f"-e {a.eps:g}, {S} top-degree sources per GPU solved as one group",
                       "V": V, "stream_edges": int(len(e1)), "window": W, "batch_c": c, "records_L": L,
                       "sources": mine, "parallelism": f"source groups of {S} x{world} GPUs (replicated graph)"},
            "event_ms_per_step": round(ev[0] / a.steps, 4), "init_solve_ms": round(init_ms, 3),
            "iterations_per_step": round(iters / a.steps, 2),
            "edges_pushed_per_step": round((st["sum_E"] - base["sum_E"]) / a.steps, 1),
            "roofline": {"bound": "hbm", "kernel": "k_pull_multi (one sweep for all sources)",
                         "achieved": round(algo / (ev[0] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(algo / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": None,
                         "note": "whole timed region (stream update + seeding + sweeps), algorithmic bytes summed over sources"},
            "cpu_baseline": None,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic_per_launch(resident):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of
    this same workload (tools/prof_pmc.sh -> profiles/r01_final_pmc_traffic_youtube.json; FETCH_SIZE
    and WRITE_SIZE in separate passes, 2 x FETCH + WRITE per the gfx950 correction of
    MI355X_MICROARCH.md). bench.py cannot run the profiler on itself, so this is read back;
    None when the file is missing or was taken with the other kind of launch."""
    path = os.path.join(ROOT, "profiles", "r01_final_pmc_traffic_youtube.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    heads = ("k_pull_resident",) if resident else ("k_pull_iter", "k_push_iter")
    tails = heads if resident else heads + ("k_push_big",)
    launches = sum(v["launches"] for k, v in d.items() if k.startswith(heads))
    total = sum(v["launches"] * v["hbm_bytes_per_launch_corrected"] for k, v in d.items() if k.startswith(tails))
    return round(total / launches, 1) if launches else None


def _force_converged(e, slot, eps):
    """After dppr_write the engine no longer trusts |r| <= eps; the restored state IS a
    converged one (it was read after a completed update), so re-establish that with an empty
    main loop pass (a full Inspect that finds nothing)."""
    e.execute_main_loop(slot, 0, eps)
    e.execute_main_loop(slot, 1, eps)
    e.reset_stats(slot)


def cpu_baseline(V, e1, e2, directed, W, c, source, eps, batches):
    """CPU leg (kind "port"): the oracle's restatement of cpu/PPRCPUMTCilkRev, timed with the
    reference's scope (IncExecuteImpl only, cpu/PPRCPUMTCilk.h:131-137) on a bounded sample of
    the same workload. Headline = all host cores (OpenMP in place of Cilk Plus, which this
    toolchain lacks); the -t 1 figure is reported beside it."""
    from oracle import oracle as orc
    threads = max(1, min(orc.max_threads(), os.cpu_count() or 1))

    def run(nthreads):
        nonlocal batches
        g = orc.Graph(V, e1, e2, directed, W, c)
        s = orc.State(V, source, eps)
        s.cilk_execute(g)
        total, done = 0.0, 0
        for _ in range(batches):
            if g.stream_updates():
                break
            g.inc_construct(1)
            t = time.perf_counter()
            if nthreads == 1:
                s.cilk_inc_execute(g)
            else:
                s.cilk_inc_execute_mt(g, nthreads)
            total += time.perf_counter() - t
            done += 1
        return total, done

    t1, n1 = run(1)
    # the parallel schedule does not scale monotonically (CAS contention on hub vertices, tiny
    # per-iteration work): try a few worker counts on a short sample and time the best one
    cands = sorted({t for t in (8, 16, 32, 40, 64) if t <= threads} | ({threads} if threads <= 16 else set()))
    best, best_rate = 1, n1 / t1 if t1 > 0 else 0.0
    short = max(2, batches // 4)
    full = batches
    for t in cands:
        batches = short
        tt, nn = run(t)
        if tt > 0 and nn / tt > best_rate:
            best, best_rate = t, nn / tt
    batches = full
    threads = best
    tm, nm = run(threads) if threads > 1 else (t1, n1)
    return {"value": round(c * nm / tm, 1) if tm > 0 else None, "unit": "edges/s", "cores": threads,
            "kind": "port", "ms_per_step": round(1e3 * tm / max(nm, 1), 2),
            "t1_value": round(c * n1 / t1, 1) if t1 > 0 else None, "t1_ms_per_step": round(1e3 * t1 / max(n1, 1), 2),
            "sample": f"first {nm} batches of the same stream/source after the from-scratch solve; oracle "
                      f"restatement of cpu/PPRCPUMTCilkRev with OpenMP workers ({threads} threads) and at -t 1, gcc -O2"}


def reference_fifo_baseline(bin_path, directed, flags, source, eps, c, batches=4):
    """The REAL reference code that can be built here: cpu/PPRCPURev.h, the reference's (deprecated)
    single-thread FIFO reverse push, compiled from /root/reference by oracle/Makefile into
    oracle/_ref/ref_driver (the binary travels to the GPU box, the sources do not). Timed scope:
    IncExecuteImpl only. None when the binary or the .bin file is not there."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not bin_path or not os.path.exists(exe) or not os.path.exists(bin_path):
        return None
    f = flags.split()
    if "-b" in f:
        f[f.index("-b") + 1] = str(batches)
    elif "-l" in f:
        f[f.index("-l") + 1] = str(batches * c)
    try:
        out = subprocess.run([exe, "-d", bin_path, "-a", "0", "-i", str(directed), "-y", "1", "-w", "0.1", *f,
                              "-s", str(source), "-e", repr(eps), "--dump", "/dev/null"],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600).stdout
        line = [ln for ln in out.splitlines() if ln.startswith("ref_fifo_inc_ms")][-1].split()
        ms, done = float(line[1]), int(line[3])
        return {"value": round(c * done / (ms * 1e-3), 1), "unit": "edges/s", "cores": 1, "kind": "reference",
                "ms_per_step": round(ms / max(done, 1), 2),
                "sample": f"first {done} batches, cpu/PPRCPURev.h (single-thread FIFO schedule) built from the reference sources"}
    except Exception as exc:  # noqa: BLE001 - a baseline must never break the benchmark line
        return {"error": str(exc)[:200]}


if __name__ == "__main__":
    main()
