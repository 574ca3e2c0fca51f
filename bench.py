#!/usr/bin/env python3
"""Headline benchmark: sliding-window dynamic PPR update throughput on MI355X.

Metric (BASELINE.json): mean per-batch PPR update time (ms) + edge-updates/sec.
A "step" is one pass of the hot path over one batch: the reference's timed region
(gpu/PPRGPU.cuh:138-164) = IncrementalBatchUpdate + ExecuteMainLoop(0) +
ExecuteMainLoop(1). Batch upload and CSR rebuild are untimed there and alternate with
the timed update per batch (gpu/PPRGPU.cuh:109-169); so here: a rolling ring of 3
resident graph epochs, per batch the untimed stream advance + upload + dppr_slide,
then the timed step in its own bracket; value = units / the sum of the K brackets.
--prestage: every epoch built before the timed region, the K steps back to back.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config livejournal] [--sources S]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Default workload = BASELINE.json configs[2], the largest single-GPU configuration:
soc-LiveJournal1 stand-in, directed, -r 0.01 -b 100, the 10 sources of a "top1000" file (ranks
[10, 1000) of the window out-degree, workload/Workload.cpp:49-51) streamed together over ONE
replica of the window graph (a source group, dppr_group_update). Other configs by name:
youtube (configs[1], one top-10 source, the resident single-source path), twitter (configs[3],
one top-10 source per GPU), friendster (configs[4]), dblp (configs[0]).

Multi-GPU: independent source vertices shard across GPUs (BASELINE.json north_star; no data-path
collective), every rank streams the same batches over its own replica of the window graph and
solves ITS sources -> weak scaling; value = (sources summed over ranks) * c * K / max-rank time.

The line also carries `parity` (checked in this very run: |r| < eps and the loop invariant for
every source at the end of the timed region, max |p_gpu - p_cpu| against the CPU leg's batches),
`roofline` (dominant kernel, hipEvent-bracketed launches) and `cpu_baseline`.
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
from dynamicppr_amd import shard  # noqa: E402  (pure Python: sharding / timing helpers; nothing here touches HIP)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBPS = 6300.0  # ... and what a streaming copy reaches there ("~6.3 TB/s achievable")
NORTH_STAR_TOL = 1e-9   # BASELINE.json: "within the repo's 1e-9 tolerance"
RING = 3                # graph epochs resident in HBM on the rolling ring (the one being solved, the one built last, one spare)
BIG_WINDOW = 4_000_000  # stream edges in the window from which a few sources per GPU are solved one after the other (binned sweeps)
CPU_SAMPLE_BATCHES = 6  # batches the live CPU leg follows when the parity comparison comes from a committed fixture (4 multi-threaded + 2 at -t 1)
SERIES_MAX = 3          # ... up to this many sources per GPU (beyond: one source group)

# per config: (sources per GPU, how they are picked). The 8-GPU configurations of BASELINE.json (twitter: 8 top-10 sources,
# friendster: the 10 sources of a top1000 file) are FIXED source sets dealt round-robin over the ranks (0 = "the config's own
# sources": 8 / 4+4 / 2+2+2+2 / 1 each; friendster 10 / 5+5 / 3+3+2+2 / 2+2+1x6) -- strong scaling, and N = 1 is the
# "one GPU, one source group" alternative the multi-GPU numbers have to be read against.
PLANS = {
    "dblp": (1, "top10"), "youtube": (1, "top10"), "livejournal": (10, "top1000"),
    "twitter": (0, "top10"), "friendster": (0, "top1000"),
}

# committed rocprofv3 --pmc summaries (tools/r05/pmc_fabric.sh: requests by size) of the dominant kernel per workload. Each carries
# the build id of the library that ran under the counters ("_stamp"); a file from another build is REFUSED (traffic: null + reason)
PMC_ROUND = "r06"
PMC_FILES = {
    ("youtube", 1): ("pmc_fabric_youtube_1src.json", ("k_pull_resident",)),
    ("dblp", 1): ("pmc_fabric_dblp_1src.json", ("k_pull_resident",)),
    ("livejournal", 10): ("pmc_fabric_livejournal_group10.json", ("k_gsweep",)),
    ("livejournal", 1): ("pmc_fabric_livejournal_1src.json", ("k_bin_scatter", "k_bin_reduce")),
    ("twitter", 8): ("pmc_fabric_twitter_group8.json", ("k_gsweep",)),
    ("twitter", 1): ("pmc_fabric_twitter_1src.json", ("k_bin_scatter", "k_bin_reduce")),
    ("friendster", 1): ("pmc_fabric_friendster_1src.json", ("k_bin_scatter", "k_bin_reduce")),
    ("friendster", 10): ("pmc_fabric_friendster_group10.json", ("k_gsweep",)),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="livejournal", help="stand-in name (dynamicppr_amd/datagen.py STAND_INS)")
    ap.add_argument("--sources", type=int, default=None, metavar="S",
                    help="sources per GPU (1 = single-source path, 2..16 = one source group); default per config")
    ap.add_argument("--pick", default=None, choices=["top10", "top1000"], help="source ranks: [0,10) or [10,1000)")
    ap.add_argument("--eps", type=float, default=1e-9)
    ap.add_argument("--batch-edges", type=int, default=None, metavar="C",
                    help="fixed batch size instead of the config's: the reference's batch-size sweep (-n 1 -c C, scripts/gpu.sh:13)")
    ap.add_argument("--schedule", default="eager", choices=["eager", "sync"])
    ap.add_argument("--data-dir", default=os.environ.get("DPPR_DATA", "/tmp/dppr_data"))
    ap.add_argument("--bin", default=None, help="real reference .bin file to use instead of the stand-in")
    ap.add_argument("--directed", type=int, default=None)
    ap.add_argument("--cpu-batches", type=int, default=None, help="batches timed on the CPU oracle (bounded sample)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-group", action="store_true", help="two sources per GPU on a large window as a source group (default there: one after the other)")
    ap.add_argument("--no-merged", action="store_true", help="skip the extra pass with the merged loop (N = 1 only)")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the calibration probes (line fills, atomics, streaming copy)")
    ap.add_argument("--no-extra-passes", action="store_true", help="skip the at-slide-accounting pass (N = 1 only)")
    ap.add_argument("--no-extra", action="store_true",
                    help="default workload at N = 1 only: do not append the configs[1] (single source, resident path) line")
    ap.add_argument("--group", type=int, default=None, help=argparse.SUPPRESS)  # old spelling of --sources
    ap.add_argument("--prestage", action="store_true",
                    help="build every epoch of the run before the timed region and run the K steps back to back in one bracket (rounds 1-5; A/B)")
    ap.add_argument("--ring-spin-ms", type=float, default=0.0, help=argparse.SUPPRESS)   # diagnostics: untimed streaming copies for this long before every timed step
    ap.add_argument("--ring-idle-ms", type=float, default=0.0, help=argparse.SUPPRESS)   # diagnostics: host sleep before every timed step
    ap.add_argument("--hbm-limit-gb", type=float, default=None, help=argparse.SUPPRESS)   # tests: pretend the device has this much free HBM
    ap.add_argument("--no-strong-block", action="store_true", help="default workload only: do not append the configs[3] strong-scaling block")
    ap.add_argument("--strong-steps", type=int, default=4, help="timed steps of the configs[3] strong-scaling block (warmup 1)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                    help="engine tuning knob (hub_min_degree, big_row_edges, pull_min_frontier, chunk_iters, pull_block, sweep_bitmap, "
                         "binned=MODE[,HA_TILES,HB_TILES,TARGET_EDGES,MIN_IDS])")
    return ap.parse_args()


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` typed as such: become the launcher. Nothing in this process has touched HIP (torch is not
        # even imported yet), and the ranks are CHILD processes -- never an exec of a process that holds the GPU.
        sys.exit(self_launch(a.gpus))
    if world != a.gpus:
        a.gpus = world

    # torch first: its bundled HIP runtime and ours share one SONAME, the first one loaded wins
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:   # more ranks than devices (a smoke run of the N > 1 path on a small node): share them
        print(f"[rank {rank}] only {ndev} device(s): sharing device {local_rank % ndev}", file=sys.stderr, flush=True)
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        # RCCL (backend "nccl") carries only the barrier and a few scalar reductions; where ranks share a device (more ranks
        # than GPUs: RCCL refuses two ranks on one device) or it cannot come up on this node, the same calls work over gloo --
        # the data path has no collective either way
        if world > ndev:
            dist.init_process_group("gloo")
        else:
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()
            except Exception as exc:  # noqa: BLE001
                print(f"[rank {rank}] RCCL init failed ({exc}); falling back to gloo", file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("gloo")
    ctx = Ctx(rank=rank, local_rank=local_rank, world=world, ndev=ndev, torch=torch, dist=dist if world > 1 else None)

    line, parity, ranks_seen = run_workload(a, ctx)
    default_workload = (a.config == "livejournal" and not a.bin and not a.batch_edges and a.sources is None and a.group is None
                        and a.schedule == "eager" and not a.tune)
    if default_workload and not a.no_extra:
        # the default run also reports configs[1] (com-youtube stand-in, one top-10 source: the single-source resident
        # path) under its own key: a second, short run of this script as a child process once this one's numbers are in
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            line["configs1_single_source"] = extra_line(["--config", "youtube", "--steps", "40", "--warmup", "5", "--no-cpu-baseline",
                                                         "--data-dir", a.data_dir])
        # ... and, at EVERY N, BASELINE.json configs[3] (twitter-2010 stand-in, its fixed 8 top-10 sources dealt round-robin over
        # the N ranks): the weak-scaled default workload is an uninformative N x by construction, this block is the strong-
        # scaling curve of an 8-GPU configuration out of the same driver command. In-process on every rank
        # (the ranks of a torchrun job cannot start another one); a few steps on the rolling epoch ring.
        if not a.no_strong_block:
            block = strong_block(a, ctx)
            if rank == 0:
                line["configs3_strong"] = block
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not parity["ok"]:
        print(f"PARITY FAILED: {parity}", file=sys.stderr, flush=True)
        sys.exit(3)
    if ranks_seen != world:
        print(f"RANK CENSUS FAILED: {ranks_seen} of {world} ranks answered", file=sys.stderr, flush=True)
        sys.exit(5)


class Ctx:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def strong_block(a, ctx):
    """BASELINE.json configs[3] on the ranks of this run: `--config twitter` (the 8 fixed sources dealt round-robin), few steps,
    no CPU leg / extra passes. Every rank runs it; rank 0 returns the summary (None elsewhere). A failure that every rank sees
    (host memory, HBM plan) is reported in the block instead of taking the headline down."""
    import copy
    b = copy.copy(a)
    b.config, b.steps, b.warmup = "twitter", max(1, a.strong_steps), 1
    b.sources = b.group = b.pick = b.batch_edges = b.bin = b.directed = None
    b.no_cpu_baseline = b.no_merged = b.no_extra_passes = b.no_ceilings = True
    b.cpu_batches, b.tune, b.prestage = None, [], False
    t0 = time.perf_counter()
    # every rank must take the same decision: the slowest "can I?" wins (host memory for a 1.4 GB stream prefix + id maps per rank)
    ok_here = 1
    try:
        import psutil
        ok_here = int(psutil.virtual_memory().available >= (6 << 30) * ctx.world)   # (one node: every rank sees the same pool)
    except Exception:  # noqa: BLE001
        pass
    if ctx.world > shard.CONFIG_SOURCE_SETS["twitter"][0]:
        return {"skipped": f"more ranks ({ctx.world}) than the configuration has sources"} if ctx.rank == 0 else None
    ok_all = shard.aggregate_min(ok_here, ctx.dist)
    if not ok_all:
        return {"skipped": "not enough host memory for every rank's copy of the stream prefix (about 6 GB per rank)"} if ctx.rank == 0 else None
    try:
        line, parity, seen = run_workload(b, ctx)
    except SystemExit as ex:   # (run_workload's refusals -- HBM plan, stream too short -- are symmetric over the ranks)
        return {"error": str(ex.code)} if ctx.rank == 0 else None
    if ctx.rank != 0:
        return None
    keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "per_rank_ms_per_step", "ranks_seen", "event_ms_per_step",
            "wall_ms_per_step_incl_graph_update", "iterations_per_step", "single_gpu_group_alternative")
    out = {k: line.get(k) for k in keep}
    out["workload"] = line["config"]["workload"]
    out["parallelism"] = line["config"]["parallelism"]
    out["hbm"] = line["config"]["hbm"]
    out["parity"] = {k: parity.get(k) for k in ("ok", "max_abs_residual", "invariant_max_err", "sources_checked")}
    rf = line.get("roofline") or {}
    out["roofline"] = {k: rf.get(k) for k in ("kernel", "frac", "achieved", "avg_launch_us", "launches", "bytes_model")}
    out["block_wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def hbm_plan(V, W, directed, c, S, group, binned, n_resident):
    """Bytes the engine will hold (dppr_create / dppr_load_window allocations, include/dppr.h), so that a plan that cannot fit is
    refused BEFORE anything is allocated: per resident epoch both CSRs and the batch (+ the binned-sweep tables), per engine the
    window ring, four key arrays and the sort scratch (+ the binned sweep's values and both word orders), per source its state."""
    Ed = W if directed else 2 * W
    L = (2 if directed else 4) * c
    epoch = 8 * (V + 1) + 12 * Ed + 8 * (V // 64 + 3) + 21 * L + (8 * Ed if binned else 4 * Ed)
    engine = 8 * W + 20 * V + (3 + (1 if directed else 0)) * 8 * Ed + 9 * Ed + (40 * Ed if binned else 0)
    state = (416 if S > 8 else 208 if S > 4 else 104 if S > 2 else 52) * V if group else S * 44 * V
    return {"resident_epochs": n_resident, "epoch_bytes": int(epoch), "engine_bytes": int(engine), "source_state_bytes": int(state),
            "plan_bytes": int(n_resident * epoch + engine + state)}


def run_workload(a, ctx):
    """One workload on the ranks of this run: returns (line, parity, ranks_seen); the line is complete on rank 0."""
    torch, D, rank, world, local_rank = ctx.torch, ctx.dist, ctx.rank, ctx.world, ctx.local_rank
    from dynamicppr_amd import datagen, engine as eng, stream as st

    # ---------------- workload (untimed) ----------------
    n_steps = a.warmup + a.steps
    S, pick = PLANS.get(a.config, (1, "top10"))
    S = a.sources or a.group or S
    # the dominant kernel's launches are event-bracketed on a few FURTHER batches after the timed region (an event
    # pair per launch widens the dispatch gaps from ~4 to ~10 us: it must not sit in the timed batches)
    n_prof = min(a.steps, 5)
    pick = a.pick or pick
    if a.bin:
        V, e1, e2 = datagen.read_bin(a.bin)
        stream_len = len(e1)
        directed = 1 if a.directed is None else a.directed
        name, flags, provenance = os.path.basename(a.bin), "-n 0 -r 0.01 -b 100", {"origin": "file", "path": a.bin}
    else:
        cfg = datagen.STAND_INS[a.config]
        stream_len = cfg.edges
        directed = cfg.directed if a.directed is None else a.directed
        name, flags = f"{cfg.name} stand-in (R-MAT scale {cfg.scale}, seed {cfg.seed})", cfg.flags
    if a.batch_edges:
        flags = f"-n 1 -c {a.batch_edges} -l {a.batch_edges * (n_steps + n_prof + 2)}"
    f = flags.split()
    opt = {f[i]: f[i + 1] for i in range(0, len(f), 2)}
    wl = st.workload_config(stream_len, 0.1, int(opt.get("-n", 0)), float(opt.get("-r", -1.0)), int(opt.get("-b", 0)),
                            int(opt.get("-c", 0)), int(opt.get("-l", 0)))
    W, c = wl.window, wl.per_batch
    max_batches = (stream_len - W) // max(c, 1)
    if n_steps + n_prof > max_batches:
        sys.exit(f"stream too short for {n_steps + n_prof} batches (max {max_batches})")
    if not a.bin:
        # the run reads W + n_steps * c stream edges: only that prefix of the seeded stream is generated
        # (rank 0 writes the file, the others wait and read it)
        need = min(stream_len, W + (n_steps + n_prof + 1) * c)
        if rank == 0:   # rank 0 generates (or finds) the file; its record of that is the one the line carries
            path = datagen.ensure_stand_in(a.config, a.data_dir, need)
        if D:
            D.barrier()
        if rank != 0:
            path = datagen.ensure_stand_in(a.config, a.data_dir, need)
        provenance = datagen.PROVENANCE[path]
        V, e1, e2 = datagen.read_bin(path)
    scaling = "weak"
    if S == 0:              # an 8-GPU configuration: ITS sources, dealt round-robin over the ranks (shard.assign_sources)
        pool = shard.config_source_pool(a.config, V, e1, e2, W, directed)
        sources = shard.assign_sources(pool, rank, world)
        if not sources:
            sys.exit(f"--gpus {world}: more ranks than the configuration has sources ({len(pool)})")
        S = len(sources)
        scaling = "strong"
    elif pick == "top1000":   # 10 ids sampled from degree ranks [10, 1000): a "top1000" file; every rank its own draw
        sources = [int(s) for s in datagen.ranked_sources(V, e1, e2, W, directed, 10, 1000, max(S, 10), seed=1 + rank)[:S]]
    else:                   # the top-10 file, dealt round-robin over the ranks
        sources = shard.assign_sources(datagen.top_sources(V, e1, e2, W, directed, 10), rank, world, per_rank=S)

    schedule = eng.SCHEDULE_EAGER if a.schedule == "eager" else eng.SCHEDULE_SYNC
    tune = {kv.split("=")[0]: (tuple(int(x) for x in kv.split("=")[1].split(",")) if "," in kv else int(kv.split("=")[1])) for kv in a.tune}
    if "merge_phases" in tune:
        a.no_merged = True   # (the whole run is on the merged loop: tuning / A-B runs)
    # Two or three sources on a window whose single-source path runs binned sweeps are cheaper one after the other than as a group (a
    # group's sweep costs about the same for 2 as for 8 sources). Measured per batch on one MI355X (tools/r06/rank_loads.sh): twitter stand-in
    # 1 / 2 in series / 4 as a group / 8 as a group: 69.8 / 133.3 / 212.6 / 245 ms; friendster 1 / 2 in series / 3 as a group / 5 / 10 as a
    # group: 99.9 / 206.5 / 421.7 / 496.5 / 531 ms -- three friendster sources in series would take 3 x 100, so up to THREE go in series
    # (twitter: 3 x 70 = a group of 3-4; from 4 on the group wins) -- what a rank of the 8-GPU deals of configs[3] / [4] holds
    pair_as_singles = 2 <= S <= SERIES_MAX and W >= BIG_WINDOW and not a.force_group
    as_group = S > 1 and not pair_as_singles

    # ---------------- the epochs resident in HBM: a rolling ring (default) or every epoch of the run (--prestage) ----------------
    # The reference keeps ONE graph and alternates the untimed graph update with the timed update per batch
    # (gpu/PPRGPU.cuh:109-169); so does this loop: RING epochs stay resident (the one being solved, the one built last, one spare),
    # every timed step is bracketed on its own, and the untimed stream advance + batch upload + dppr_slide sit between the
    # brackets. --prestage is the form of rounds 1-5 (all W + K + profile epochs built before the timed region, the K steps back
    # to back in ONE bracket): fine for configs[0..2], 200+ GB for configs[3] / [4] at the driver's step counts.
    n_resident = (n_steps + n_prof + 1) if a.prestage else RING
    live_ids_guess = min(V, 2 * (W if directed else 2 * W))
    binned = (not as_group) and live_ids_guess >= (1 << 20) and (W if directed else 2 * W) >= (1 << 22)
    plan = hbm_plan(V, W, directed, c, S, as_group, binned, n_resident)
    free0, total_hbm = torch.cuda.mem_get_info(local_rank)
    if a.hbm_limit_gb is not None:
        free0 = min(free0, int(a.hbm_limit_gb * 1e9))
    share = max(1, -(-world // ctx.ndev))   # ranks sharing this device (a smoke run of the N > 1 path on a small node)
    fits = plan["plan_bytes"] * share <= 0.94 * free0
    if not shard.aggregate_min(int(fits), D):   # (every rank takes the same decision: a refusal on one device must not leave the others at a barrier)
        sys.exit(("" if not fits else f"[rank {rank}] another rank's plan does not fit; this rank's would: ") + f"bench.py: HBM plan does not fit: {plan['plan_bytes'] / 1e9:.1f} GB per rank ({n_resident} resident epochs x "
                 f"{plan['epoch_bytes'] / 1e9:.2f} GB + engine {plan['engine_bytes'] / 1e9:.1f} GB + source state "
                 f"{plan['source_state_bytes'] / 1e9:.1f} GB) x {share} rank(s) on this device against {free0 / 1e9:.1f} GB free of "
                 f"{total_hbm / 1e9:.1f} GB" + ("; drop --prestage (rolling ring of 3 epochs)" if a.prestage else ""))
    min_free = [free0]

    def sample_hbm():
        min_free[0] = min(min_free[0], torch.cuda.mem_get_info(local_rank)[0])

    def make_solver(e):
        return (PairSolver(e, sources) if pair_as_singles else GroupSolver(e, sources)) if S > 1 else SingleSolver(e, sources[0])

    class Follower:
        """One engine following the stream: stage(k) = the UNTIMED part of batch k (stream advance, batch upload, graph update:
        gpu/PPRGPU.cuh:114-135), solver.update(eps, k) = the timed part."""

        def __init__(self, setup=None):
            gp = {k: tune.pop(k) for k in ("gpush_enter_pairs", "gpush_max_edges") if k in tune}   # (tuning runs: dppr_set_group_push)
            self.e = eng.Engine(V, W, directed, c, n_epochs=n_resident, device=local_rank, schedule=schedule, **tune)
            tune.update(gp)
            if gp:
                self.e.set_group_push(gp.get("gpush_enter_pairs", -1), 0, gp.get("gpush_max_edges", 0))
            if setup:
                setup(self.e)
            self.ss = st.SlidingStream(V, e1, e2, directed, wl)
            self.e.load_window(*self.ss.serialize_edge_stream())
            self.solver = make_solver(self.e)
            self.staged, self.L, self.keep_at, self.w_keep = 0, 0, None, None
            self.step_ms = []
            self.stage_s = 0.0
            sample_hbm()

        def stage(self, k):
            while self.staged < k:
                t = time.perf_counter()
                assert not self.ss.stream_updates()
                b1, b2, ins = self.ss.batch_arrays()
                self.L = len(b1)
                self.e.set_batch(b1, b2, ins)
                self.e.slide(*self.ss.new_arrays())
                self.staged += 1
                self.stage_s += time.perf_counter() - t
                if self.staged == self.keep_at:
                    self.w_keep = [x.copy() for x in self.ss.serialize_edge_stream()]   # the window at the end of the timed region
                sample_hbm()

        def sync(self):
            torch.cuda.synchronize()
            self.e.synchronize()

        def follow(self, eps, timed_dist=None):
            """init solve, W warmup batches, K timed batches. Returns (init_ms, seconds of the K brackets [max over ranks when
            timed_dist is given], sum of the event times, wall seconds of the K batches incl. the untimed graph updates)."""
            self.keep_at = n_steps
            init_ms = self.solver.init_solve(eps)
            if a.prestage:
                self.stage(n_steps + n_prof)
            for k in range(1, a.warmup + 1):
                self.stage(k)
                self.solver.update(eps, k)
            self.solver.begin_timed()
            ev, acc = [0.0], [0.0]
            if a.prestage:
                def run_steps():
                    for k in range(a.warmup + 1, n_steps + 1):
                        ms = self.solver.update(eps, k)
                        ev[0] += ms
                        self.step_ms.append(round(ms, 4))
                dt, _ = shard.timed_region(run_steps, self.sync, timed_dist)
            else:
                def run_steps():
                    for k in range(a.warmup + 1, n_steps + 1):
                        self.stage(k)
                        self.sync()
                        if a.ring_idle_ms > 0:
                            time.sleep(a.ring_idle_ms * 1e-3)
                        if a.ring_spin_ms > 0:
                            t_spin = time.perf_counter()
                            while (time.perf_counter() - t_spin) * 1e3 < a.ring_spin_ms:
                                eng.bench_stream_copy(1 << 28, reps=1, device=local_rank)
                        t = time.perf_counter()
                        ms = self.solver.update(eps, k)
                        ev[0] += ms
                        self.step_ms.append(round(ms, 4))
                        self.sync()
                        acc[0] += time.perf_counter() - t
                dt, _ = shard.timed_region(run_steps, self.sync, timed_dist, summed=acc)
            return init_ms, dt, ev[0], shard.timed_region.last_wall

        def close(self):
            self.e.close()

    # ---------------- warmup + timed region: exactly K steps, barrier + synchronize on both sides ----------------
    # CPU leg: on streams the -t 1 oracle follows in seconds per batch, it follows TWO sources through every batch up to the end of
    # the timed region (multi-threaded port for positioning, -t 1 for the last two batches) and p is compared THERE; on larger
    # streams (a twitter batch is minutes at -t 1) the comparison lives in tests/test_fullsize_golden_gpu.py and the leg is a short sample
    cpu_follow = stream_len < 100_000_000
    cpu_batches = a.cpu_batches if a.cpu_batches is not None else (n_steps if cpu_follow else 2)
    blocks = shard.line_blocks(rank, world, a.no_cpu_baseline, a.no_merged, a.schedule)
    want_cpu = blocks["cpu_baseline"]   # (at N > 1 too: rank 0's line carries the same blocks at every N)
    fw = Follower()
    e, solver = fw.e, fw.solver
    init_ms, dt, ev_ms, wall_s = fw.follow(a.eps, D)
    L = fw.L
    stats = solver.stats()
    units = shard.aggregate_units(S * c * a.steps, D)
    total_sources = shard.aggregate_units(S, D)
    ranks_seen, rank_dts, backend = shard.rank_census(shard.timed_region.last_local, D)

    # `value` is measured under the reference's bracket (gpu/PPRGPU.cuh:138-164, gpu/StreamUpdate.cuh:7-33): CopyOutDegree and the
    # grouping of the batch's records by tail run INSIDE the timed region (the engine's default since ABI 4). What they cost, run on
    # their own (a whole-batch resident launch does both inside the launch: the figure is then what the separate kernels would cost):
    grouping_ms = e.time_batch_grouping(epoch=n_steps, reps=5) * (1 if as_group else S)

    # ---------------- parity at the end of the timed region: every source of this rank ----------------
    w1, w2 = fw.w_keep
    src_e, dst_e = (w1, w2) if directed else (np.concatenate([w1, w2]), np.concatenate([w2, w1]))
    max_r, max_inv = 0.0, 0.0
    golden = load_bench_golden(a, V, W, c, directed, sources, n_steps, e1, e2) if rank == 0 else None
    gold_dp = gold_n = None
    for i, s in enumerate(sources):
        p, r = solver.read(i)
        max_r = max(max_r, float(np.max(np.abs(r))))
        max_inv = max(max_inv, invariant_max_err(p, r, src_e, dst_e, V, s))
        if golden:
            d1 = float(np.max(np.abs(p[golden["sample"]] - golden["p_sample"][i])))
            d2 = float(np.max(np.abs(p[golden["top_ids"][i]] - golden["top_p"][i])))
            gold_dp, gold_n = max(gold_dp or 0.0, d1, d2), (gold_n or 0) + 1
    parity = {"eps": a.eps, "max_abs_residual": max_r, "invariant_max_err": max_inv, "sources_checked": len(sources),
              "max_abs_dp_vs_cpu_t1": None, "tolerance": NORTH_STAR_TOL,
              "ok": bool(max_r < a.eps and max_inv < 1e-12)}
    if golden:
        # ALL of the rank's sources against the committed -t 1 states of this very run (tests/golden/make_bench_golden.py: schedule A,
        # every source through every batch at -t 1; stream checksum, sources, eps and batch number matched) -- VERDICT r05 item 4
        parity.update({"sources_compared": gold_n, "max_abs_dp": gold_dp,
                       "compared_with": f"{golden['file']}: p of all {gold_n} sources after batch {n_steps} (the end of the timed region) at "
                                        f"{len(golden['sample'])} sampled vertices + the 1000 of largest p, CPU -t 1 (schedule A) throughout",
                       "ok": bool(parity["ok"] and gold_dp < NORTH_STAR_TOL)})
    p_end = [solver.read(i)[0] for i in range(min(2, len(sources)))] if want_cpu else []   # p at the END of the timed region
    p_head = p_end if p_end else [solver.read(0)[0]]   # (what the at-slide-accounting pass must arrive at too)
    # ... and the other sources' too, should the CPU leg turn out cheap enough to follow all of them (below)
    p_rest = ([solver.read(i)[0] for i in range(2, len(sources))]
              if want_cpu and not golden and len(sources) > 2 and (len(sources) - 2) * V * 8 <= (1 << 30) else [])

    # ---------------- roofline of the dominant kernel ----------------
    roof = cpu = p_cpu = None
    ps = None
    if rank == 0:
        # the n_prof batches that FOLLOW the timed region on the same stream, with a hipEvent pair around every launch of the dominant kernel
        ps = solver.profile(a.eps, fw.stage, n_steps, n_prof)
    sample_hbm()
    hbm_peak = int(total_hbm - min_free[0])
    fw.close()
    if rank == 0:
        all_iters = {"launches": ps["push_launches"], "avg_launch_us": round(1e3 * ps["push_ms"] / max(ps["push_launches"], 1), 3),
                     "algorithmic_bytes_per_launch": round((72 * ps["sum_F"] + 24 * ps["sum_E"] + 4 * ps["sum_N"]) / max(ps["push_launches"], 1), 1)}
        if ps.get("sweep_launches", 0) > 0 and not ps["persist_launches"]:
            # one launch per sweep (k_gsweep, k_bin_scatter + k_bin_reduce, k_pull_iter): the dominant kernel's OWN launches, bytes
            # and time -- the push iterations of the same loops (cheap tails, other kernels) are reported beside it, not mixed in,
            # so that frac can be recomputed from a rocprofv3 kernel-stats file of the same command (tools/check_profiles.py)
            ps = dict(ps, sum_F=ps["sweep_F"], sum_E=ps["sweep_E"], sum_N=ps["sweep_F"], push_ms=ps["sweep_ms"], push_launches=ps["sweep_launches"],
                      iterations=ps["sweep_launches"])
        roof = roofline_block(a, ps, S, as_group, solver, eng, local_rank, n_prof)
        roof["all_iteration_launches"] = all_iters
        S_eff = S if as_group else 1
        e_pull = stats["sum_E"] if roof["form"] == "resident" else stats["sweep_E"] if roof["form"] in ("pull", "binned") else 0
        roof["whole_batch_algorithmic_GBps"] = round((stats["algorithmic_bytes"] - (24 - (8 + 4 / S_eff)) * e_pull) / max(ev_ms * 1e-3, 1e-12) / 1e9, 2)
        if want_cpu:
            t_cpu0 = time.perf_counter()
            if golden and a.cpu_batches is None and n_steps > CPU_SAMPLE_BATCHES:
                # All of the rank's sources are held to the committed -t 1 states above: the live CPU leg need not follow every batch to
                # the end of the timed region for the comparison's sake and is the BOUNDED sample the baseline is meant to be (two sources,
                # from-scratch solve + a few batches: about half a minute of CPU work instead of two)
                cpu_batches = CPU_SAMPLE_BATCHES
            follow_all = cpu_batches == n_steps
            cpu = cpu_baseline(V, e1, e2, directed, W, c, sources[:len(p_end)], a.eps, cpu_batches, p_end if follow_all else [], stream_len)
            cpu_leg_s = time.perf_counter() - t_cpu0
            p_cpu = cpu.pop("p_cpu") if follow_all else (cpu.pop("p_cpu"), None)[1]   # (states at the END of the timed region, or nothing to compare with)
            worst = cpu.pop("max_abs_dp")
            if cpu.get("compared") is None:
                cpu.pop("compared", None)
            if worst is not None:
                parity["max_abs_dp_vs_cpu_t1"] = worst
                parity["cpu_compared"] = cpu.pop("compared")
                parity["ok"] = bool(parity["ok"] and worst < NORTH_STAR_TOL)
                # ALL sources of the rank against the CPU when that fits in about a minute, else two -- and the line says why
                rest = len(sources) - len(p_end)
                projected = cpu_leg_s / max(len(p_end), 1) * rest
                if rest > 0 and p_rest and projected <= 60.0:
                    more = cpu_baseline(V, e1, e2, directed, W, c, sources[len(p_end):], a.eps, cpu_batches, p_rest, stream_len)
                    parity["max_abs_dp_vs_cpu_t1"] = max(worst, more["max_abs_dp"])
                    parity["cpu_compared"] = f"p of all {len(sources)} sources" + parity["cpu_compared"][parity["cpu_compared"].index(" after batch"):]
                    parity["ok"] = bool(parity["ok"] and parity["max_abs_dp_vs_cpu_t1"] < NORTH_STAR_TOL)
                elif rest > 0:
                    parity["cpu_compared_why_not_all"] = (
                        f"{len(p_end)} of {len(sources)} followed LIVE by the CPU leg ({cpu_leg_s:.0f} s for {len(p_end)} sources through {cpu_batches} batches; "
                        f"the other {rest} would add about {projected:.0f} s)"
                        + (f"; all {gold_n} are compared with the committed -t 1 states (sources_compared, max_abs_dp)" if golden else
                           "; all of a rank's sources are compared at this size in tests/test_fullsize_gpu.py"))
            if stream_len < 10_000_000 and not a.bin:   # the reference's own FIFO binary needs the whole file
                full = datagen.ensure_stand_in(a.config, a.data_dir)
                cpu["reference_fifo"] = reference_fifo_baseline(full, directed, flags, sources[0], a.eps, c)

    # ---------------- the same K steps once more with the MERGED loop (include/dppr.h dppr_set_phase_merge) ----------------
    # not the reference's schedule, therefore not `value`: one loop for residuals of both signs, run to eps / 4. A second engine
    # follows the same stream from its start; its p is compared with the same CPU -t 1 states at the end of the timed region.
    merged = None
    if shard.line_blocks(rank, world, a.no_cpu_baseline, a.no_merged, a.schedule)["merged_loop"]:   # (a.no_merged may have been set by --tune above)
        f2 = Follower(setup=lambda en: en.set_phase_merge(True, 4))
        _, dt2, _, _ = f2.follow(a.eps, None)
        st2 = f2.solver.stats()
        mr, mi, md = 0.0, 0.0, None
        for i, sv in enumerate(sources):
            p2, r2 = f2.solver.read(i)
            mr = max(mr, float(np.max(np.abs(r2))))
            mi = max(mi, invariant_max_err(p2, r2, src_e, dst_e, V, sv))
            if want_cpu and cpu is not None and i < len(p_cpu or []):
                md = max(md or 0.0, float(np.max(np.abs(p2 - p_cpu[i]))))
            if golden:
                md = max(md or 0.0, float(np.max(np.abs(p2[golden["sample"]] - golden["p_sample"][i]))),
                         float(np.max(np.abs(p2[golden["top_ids"][i]] - golden["top_p"][i]))))
        f2.close()
        merged = {"schedule": "one loop for residuals of both signs, |r| > eps / 4 (dppr_set_phase_merge; NOT the reference's two loops: reported beside `value`, never as it)",
                  "ms_per_step": round(1e3 * dt2 / a.steps, 4), "value": round(S * c * a.steps / dt2, 1), "unit": "edges/s",
                  "iterations_per_step": round(st2["iterations"] / a.steps, 2), "speedup_vs_value": round(dt / dt2, 3),
                  "parity": {"max_abs_residual": mr, "invariant_max_err": mi, "max_abs_dp_vs_cpu_t1": md, "tolerance": NORTH_STAR_TOL,
                             "ok": bool(mr <= a.eps / 4 and mi < 1e-12 and (md is None or md < NORTH_STAR_TOL))}}
    # ---------------- the other accounting beside it: grouping + CopyOutDegree at slide time (rounds 3-4) ----------------
    # The same K steps of the same stream on an engine under dppr_set_batch_grouping(1): the records are grouped when the batch is
    # uploaded (dppr_slide), so the event-timed batch time is the at-slide accounting's. N = 1 on streams whose from-scratch solve
    # is cheap; elsewhere the figure is the subtraction (batch time - the grouping kernels on their own).
    at_slide = {"event_ms_per_step": round(ev_ms / a.steps - grouping_ms, 4), "how": "estimated: event_ms_per_step - grouping_ms_per_step"}
    if rank == 0 and world == 1 and stream_len < 100_000_000 and not a.no_extra_passes:
        f3 = Follower(setup=lambda en: en.set_batch_grouping(1))
        _, _, ev3, _ = f3.follow(a.eps, None)
        worst3 = max(float(np.max(np.abs(f3.solver.read(i)[0] - p_head[i]))) for i in range(len(p_head)))
        f3.close()
        at_slide = {"event_ms_per_step": round(ev3 / a.steps, 4), "how": "measured: the same K steps of the same stream on an engine under dppr_set_batch_grouping(1)",
                    "max_abs_dp_vs_headline_state": worst3, "sources_compared": len(p_head)}

    line = None
    if rank == 0:
        value = units / dt
        line = {
            "metric": "edge-updates/sec (ppr_throughput, summed over sources); ms_per_step = mean per-batch PPR update time",
            "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": scaling,
            "ranks_seen": ranks_seen, "backend": backend, "per_rank_ms_per_step": [round(1e3 * t / a.steps, 4) for t in rank_dts],
            "launcher": os.environ.get("DPPR_BENCH_LAUNCHER", "torchrun" if world > 1 else "single process"),
            "devices_visible": ctx.ndev, "build_id": eng.build_id(),
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{name}, {'directed' if directed else 'undirected'}, -a 0 -y 1 -w 0.1 {flags} "
                                   f"-e {a.eps:g}, "
                                   + (f"the configuration's {total_sources} sources" if scaling == "strong" else f"{S} source(s) per GPU")
                                   + f" from degree ranks {'[10,1000) (a top1000 file)' if pick == 'top1000' else '[0,10) (the top10 file)'}"
                                   + (f", dealt round-robin over {world} GPU(s)" if scaling == "strong" else "")
                                   + (f", a GPU's {S} sources solved one after the other (single-source path)" if pair_as_singles else
                                      ", a GPU's sources streamed together as one source group over one graph replica" if S > 1 else ""),
                       "V": V, "stream_edges": int(stream_len), "window": W, "batch_c": c, "records_L": L,
                       "sources": sources, "schedule": a.schedule,
                       "timed_region": {
                           "what": "IncrementalBatchUpdate replay + ExecuteMainLoop(0) + ExecuteMainLoop(1) per batch (gpu/PPRGPU.cuh:138-164), "
                                   "event-bracketed inside dppr_update / dppr_group_update; batch upload and graph rebuild outside, as in the reference",
                           "loop": ("pre-staged: every epoch of the run built before the timed region, the K steps back to back in one barrier + synchronize bracket" if a.prestage else
                                    f"the reference's loop shape (gpu/PPRGPU.cuh:109-169) on a rolling ring of {RING} resident epochs: per batch the untimed stream advance + batch "
                                    "upload + dppr_slide, then the timed step between synchronize + host-clock brackets; value = units / the SUM of the K brackets "
                                    "(MAX over ranks), the barrier + synchronize pair around the whole run"),
                           "grouping": "in_region", "copy_out_degree": "in_region",
                           "grouping_ms_per_step": round(grouping_ms, 4),
                           "at_slide_accounting": at_slide,
                           "note": "`value` / `ms_per_step` are measured under the reference's bracket: CopyOutDegree and the grouping of the batch's "
                                   "records by tail run inside the timed region (one ranking launch up to 4 Ki records, hand-written LDS radix placement beyond; a "
                                   "whole-batch resident launch does both itself). grouping_ms_per_step = those kernels run on their own; "
                                   "at_slide_accounting = the batch time when both are done at slide time instead (the accounting of rounds 3-4)"},
                       "hbm": dict(plan, hbm_peak_bytes=hbm_peak, hbm_total_bytes=int(total_hbm),
                                   note="plan = what the engine was expected to allocate (refused up front when it cannot fit); hbm_peak_bytes = device total - "
                                        "the least free memory seen (hipMemGetInfo after every graph update; the whole device, other processes included)"),
                       "parallelism": (f"{total_sources} sources dealt round-robin over {world} GPU(s) (rank 0: {S}), replicated graph, no collective"
                                       if scaling == "strong" else f"{S} source(s) per GPU x {world} GPU(s), replicated graph, no collective"),
                       "stream_file": provenance},
            "event_ms_per_step": round(ev_ms / a.steps, 4), "event_ms_of_each_step": fw.step_ms, "init_solve_ms": round(init_ms, 3),
            "wall_ms_per_step_incl_graph_update": round(1e3 * wall_s / a.steps, 4),
            "graph_update_ms_per_batch_host_clock": round(1e3 * fw.stage_s / max(fw.staged, 1), 4),
            "per_source_edges_per_s": round(value / total_sources, 1),
            "iterations_per_step": round(stats["iterations"] / a.steps, 2),
            "pull_iterations_per_step": round(stats["pull_iterations"] / a.steps, 2),
            "edges_pushed_per_step": round(stats["sum_E"] / a.steps, 1),
            "parity": parity, "roofline": roof, "cpu_baseline": cpu, "merged_loop": merged,
        }
        if scaling == "strong":
            # what the multi-GPU value has to be read against: ALL of the configuration's sources as one source group on ONE GPU
            # (this very script at --gpus 1; the committed line of that run is quoted when this is an N > 1 run)
            alt = next((pth for pth in (os.path.join(ROOT, "profiles", f"r0{rnd}_bench_{a.config}_group_1gpu.json") for rnd in (6, 5, 4, 3))
                        if os.path.exists(pth)), "")
            line["single_gpu_group_alternative"] = (
                {"this_run": True, "ms_per_step": line["ms_per_step"], "value": line["value"]} if world == 1 else
                ({k: json.loads(open(alt).read().splitlines()[-1]).get(k) for k in ("ms_per_step", "value", "unit", "steps")} | {"source": os.path.relpath(alt, ROOT)})
                if os.path.exists(alt) else None)
    return line, parity, ranks_seen


def roofline_block(a, ps, S, as_group, solver, eng, local_rank, n_prof):
    """`roofline` of the dominant kernel from its hipEvent-bracketed launches (ps = the engine's statistics of the profiled batches)."""
    # Algorithmic bytes per launch. SURVEY.md 8(d) prices a traversed edge at 24 bytes per source (4 column entry + 4 degree + 16
    # residual read-modify-write): that IS what a push iteration must move. A SWEEP (k_gsweep, k_pull_iter, k_pull_resident, and the
    # binned k_bin_scatter + k_bin_reduce) performs no residual RMW per edge: per active (edge, source) pair it must read the 8-byte
    # snapshot value, and the 4-byte column entry once for the S sources that share the graph -- 72 F + (8 + 4 / S) E + 4 N (VERDICT
    # r05 item 7: priced so that algorithmic <= what the counters see; tools/check_profiles.py enforces frac <= 1.1 x frac_traffic).
    # The binned sweep hands its values over through memory (written once and read once per run): its counter traffic is ~1.5 x this
    # minimum by design -- the price of streaming instead of gathering. SURVEY's unit is carried beside it as work_rate_survey_unit.
    S_eff = S if as_group else 1
    form = ("resident" if ps["persist_launches"] else "binned" if ps.get("binned_sweeps") else "pull" if ps.get("sweep_launches") else "push")
    pull_form = form in ("resident", "pull", "binned")
    e_price = (8 + 4 / S_eff) if pull_form else 24.0
    survey_bytes = 72 * ps["sum_F"] + 24 * ps["sum_E"] + 4 * ps["sum_N"]
    model_bytes = 72 * ps["sum_F"] + e_price * ps["sum_E"] + 4 * ps["sum_N"]
    t_s = ps["push_ms"] * 1e-3
    launches = max(ps["push_launches"], 1)
    launch_s = t_s / launches
    traffic, traffic_src, traffic_detail = pmc_traffic_per_launch(a.config if not a.bin and not a.batch_edges else None, S, eng.build_id())
    ceilings = measure_ceilings(eng, local_rank) if not a.no_ceilings else None
    achieved = model_bytes / t_s / 1e9 if t_s > 0 else 0.0
    survey_rate = survey_bytes / t_s / 1e9 if t_s > 0 else 0.0
    return {
        "bound": "hbm", "kernel": solver.kernel_name(ps), "form": form,
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 5),
        "bytes_model": ("72 F + 24 E + 4 N (SURVEY.md 8(d), one source)" if not pull_form else
                        f"72 F + (8 + 4 / {S_eff}) E + 4 N: a sweep reads the 8-byte snapshot value per active (edge, source) pair and the column entry once "
                        f"for its {S_eff} source(s); no residual read-modify-write per edge (F, E, N summed over the sources)"),
        "work_rate_survey_unit": {"GBps": round(survey_rate, 2) if t_s > 0 else None,
                                  "over_peak": round(survey_rate / HBM_PEAK_GBPS, 5) if t_s > 0 else None,
                                  "bytes_per_launch": round(survey_bytes / launches, 1),
                                  "note": "72 F + 24 E + 4 N per source: SURVEY.md 8(d)'s unit as written -- a WORK rate for a pull form (which does not move "
                                          "these bytes; > 1 is possible for a group), kept for comparison with rounds 1-5"},
        "frac_traffic": round(traffic / launch_s / 1e9 / HBM_PEAK_GBPS, 5) if traffic and launch_s > 0 else None,
        "traffic": traffic, "traffic_source": traffic_src,
        "traffic_kind": "L2 <-> fabric bytes per launch (read requests counted by size 32 / 64 / 128 B + write requests 64 / 32 B; "
                        "Infinity-Cache hits are in them: fabric traffic, an upper bound of HBM traffic)",
        "traffic_fetch_size_method": traffic_detail,
        "ceilings_measured_in_this_run": ceilings,
        "frac_traffic_of_stream_copy": (round(traffic / launch_s / 1e9 / ceilings["stream_copy_GBps"], 5)
                                        if traffic and launch_s > 0 and ceilings and ceilings.get("stream_copy_GBps") else None),
        "iterations_per_launch": round(ps["iterations"] / launches, 2),
        "launches": ps["push_launches"], "avg_launch_us": round(1e3 * ps["push_ms"] / launches, 3),
        "algorithmic_bytes_per_launch": round(model_bytes / launches, 1),
        "launches_from": f"the {n_prof} batches that follow the timed region on the same stream",
        "build_id": eng.build_id(),
        "note": "achieved = bytes_model of the hipEvent-bracketed launches of the dominant kernel / their time; frac = achieved / peak. "
                "traffic = what the counters saw per launch (a committed rocprofv3 --pmc pass of THIS build of the library, refused "
                "otherwise); frac_traffic = traffic / the launch time measured here / peak. ceilings_measured_in_this_run: random "
                "128-byte line fills, returning f64 atomics and a streaming copy on this device, this run (DESIGN.md section 6)",
    }


def load_bench_golden(a, V, W, c, directed, sources, n_steps, e1, e2):
    """The committed -t 1 states of this run (tests/golden/bench_<config>.npz, made by tests/golden/make_bench_golden.py) when
    EVERYTHING matches: window / batch shape, eps, the sources in order, a checkpoint at batch n_steps and the checksum of the
    stream edges read up to there. None otherwise (other workloads, other step counts)."""
    path = os.path.join(ROOT, "tests", "golden", f"bench_{a.config}.npz")
    if a.bin or a.batch_edges or a.schedule != "eager" or not os.path.exists(path):
        return None
    try:
        z = np.load(path)
        if [int(x) for x in z["config"]] != [V, W, c, directed] or float(z["eps"][0]) != a.eps:
            return None
        if [int(x) for x in z["sources"]][:len(sources)] != list(sources) or n_steps not in [int(x) for x in z["checkpoints"]]:
            return None
        import xxhash
        h = xxhash.xxh3_64()
        n = W + n_steps * c
        h.update(np.ascontiguousarray(e1[:n], dtype="<i4").tobytes())
        h.update(np.ascontiguousarray(e2[:n], dtype="<i4").tobytes())
        if f"xxh3_64:{h.hexdigest()}" != str(z[f"checksum.b{n_steps}"]):
            return None
        k = n_steps
        return {"file": os.path.relpath(path, ROOT), "sample": z["sample"],
                "p_sample": [z[f"b{k}.s{i}.p_sample"] for i in range(len(sources))],
                "top_ids": [z[f"b{k}.s{i}.top_ids"] for i in range(len(sources))],
                "top_p": [z[f"b{k}.s{i}.top_p"] for i in range(len(sources))]}
    except Exception as ex:  # noqa: BLE001 -- a missing / unreadable fixture never takes the line down; the live CPU leg still compares
        print(f"bench.py: golden fixture {path} not used ({type(ex).__name__}: {ex})", file=sys.stderr)
        return None


def launch_command(n, argv, port):
    """The torchrun command line of an N-rank run of this script (one process per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n):
    """`python bench.py --gpus N` without torchrun: run `python -m torch.distributed.run --nproc-per-node N bench.py <same args>`
    as a child, pass its stderr through, print the ONE JSON line of its rank 0 and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:   # a free rendezvous port (the driver passes its own when it launches torchrun itself)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DPPR_BENCH_LAUNCHER="self")
    r = subprocess.run(launch_command(n, sys.argv[1:], port), stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    for ln in r.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        print("bench.py: the ranks exited 0 but printed no line", file=sys.stderr)
        return 4
    return r.returncode


def extra_line(args):
    """Run this script on another workload (child process, its own engine) and keep the figures that matter."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=900)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"],
                "steps": d["steps"], "warmup": d["warmup"], "roofline_kernel": d["roofline"]["kernel"],
                "roofline_frac": d["roofline"]["frac"], "avg_launch_us": d["roofline"]["avg_launch_us"], "parity_ok": d["parity"]["ok"],
                "merged_loop": ({k: d["merged_loop"][k] for k in ("ms_per_step", "value", "iterations_per_step")} | {"parity_ok": d["merged_loop"]["parity"]["ok"]})
                if d.get("merged_loop") else None}
    except Exception as ex:  # the extra line never takes the headline down with it
        return {"error": f"{type(ex).__name__}: {ex}"}


class SingleSolver:
    """One source on the single-source path (dppr_update: resident / per-iteration sweeps, push)."""

    def __init__(self, e, source):
        self.e, self.source = e, source
        self.slot = e.add_source(source)

    def init_solve(self, eps, epoch=-1):
        return self.e.init_solve(self.slot, eps, epoch)

    def update(self, eps, epoch):
        return self.e.update(self.slot, eps, epoch=epoch)

    def read(self, i):
        return self.e.read(self.slot)

    def begin_timed(self):
        self.e.reset_stats(self.slot)

    def stats(self):
        return self.e.stats(self.slot)

    def profile(self, eps, stage, k0, n):
        """The n batches that follow batch k0 on the same stream with a hipEvent pair around every launch of the dominant kernel
        (the extra events would perturb a sub-millisecond timed batch: they are never in the timed region)."""
        e, slot = self.e, self.slot
        e.reset_stats(slot)
        e.set_profiling(True)
        for k in range(k0 + 1, k0 + n + 1):
            stage(k)
            e.update(slot, eps, epoch=k)
        e.set_profiling(False)
        return e.stats(slot)

    def kernel_name(self, ps):
        if ps["persist_launches"]:
            return "k_pull_resident (one launch = a run of frontier iterations, state kept on chip)"
        if ps.get("binned_sweeps"):
            return "k_bin_scatter + k_bin_reduce (one frontier iteration as two streaming passes; a launch = the pair)"
        return "k_pull_iter (one frontier iteration as a gather sweep)" if ps.get("sweep_launches") else "k_push_iter (one frontier iteration)"


class PairSolver:
    """Two or three sources on the single-source path, one after the other, over the same resident epochs."""

    def __init__(self, e, sources):
        self.e, self.parts = e, [SingleSolver(e, s) for s in sources]

    def init_solve(self, eps, epoch=-1):
        return sum(p.init_solve(eps, epoch) for p in self.parts)

    def update(self, eps, epoch):
        return sum(p.update(eps, epoch) for p in self.parts)

    def read(self, i):
        return self.parts[i].read(0)

    def begin_timed(self):
        for p in self.parts:
            p.begin_timed()

    def stats(self):
        every = [p.stats() for p in self.parts]
        return {k: sum(st[k] for st in every) for k in every[0]}

    def profile(self, eps, stage, k0, n):
        return self.parts[0].profile(eps, stage, k0, n)   # (the kernel is the same for both: the first source follows the further batches)

    def kernel_name(self, ps):
        return self.parts[0].kernel_name(ps)


class GroupSolver:
    """2..16 sources solved together (dppr_group_update: multi-source sweeps, k_gsweep)."""

    def __init__(self, e, sources):
        self.e, self.sources = e, sources
        self.gid = e.add_source_group(sources)

    def init_solve(self, eps, epoch=-1):
        return self.e.group_init_solve(self.gid, eps, epoch)

    def update(self, eps, epoch):
        return self.e.group_update(self.gid, eps, epoch=epoch)

    def read(self, i):
        return self.e.group_read(self.gid, i)

    def begin_timed(self):
        self.e.group_reset_stats(self.gid)

    def stats(self):
        return self.e.group_stats(self.gid)

    def profile(self, eps, stage, k0, n):
        """The n batches that follow batch k0 on the same stream with a hipEvent pair around every sweep."""
        self.e.group_reset_stats(self.gid)
        self.e.set_profiling(True)
        for k in range(k0 + 1, k0 + n + 1):
            stage(k)
            self.e.group_update(self.gid, eps, epoch=k)
        self.e.set_profiling(False)
        return self.e.group_stats(self.gid)

    def kernel_name(self, ps):
        return "k_gsweep (one frontier iteration of all sources of the group)"


def invariant_max_err(p, r, src, dst, V, source, alpha=0.15):
    """p[u] + a r[u] == a [u==s] + (1-a)/(outdeg(u)+1) * sum_{v in out(u)} p[v]."""
    outdeg = np.bincount(src, minlength=V)
    acc = np.bincount(src, weights=p[dst], minlength=V)
    rhs = (1.0 - alpha) / (outdeg + 1.0) * acc
    rhs[source] += alpha
    return float(np.max(np.abs(p + alpha * r - rhs)))


def measure_ceilings(eng, device):
    """SURVEY.md 8(d)'s calibrated ceilings, measured on this device in this run (dppr_bench_*): random 128-byte line fills out of a
    1 GiB table (what a gather-bound sweep is held against), returning f64 atomics into a V-sized table (a push iteration), and a
    streaming copy (what HBM sustains for a kernel that only streams)."""
    out = {}
    try:
        lines = 1 << 26
        ms = eng.bench_line_fills(1 << 30, lines, reps=3, device=device)
        out["line_fills_per_s"] = round(lines / (ms * 1e-3), 1)
        out["line_fill_GBps"] = round(lines * 128 / (ms * 1e-3) / 1e9, 1)
        n = 1 << 26
        ms = eng.bench_atomics(1 << 23, n, scope=0, reps=3, device=device)
        out["returning_f64_atomics_per_s"] = round(n / (ms * 1e-3), 1)
        out["atomics_algorithmic_GBps"] = round(n * 24 / (ms * 1e-3) / 1e9, 1)   # (24 B of SURVEY 8(d) per traversed edge = per atomic)
        nbytes = 1 << 30
        ms = eng.bench_stream_copy(nbytes, reps=5, device=device)
        out["stream_copy_GBps"] = round(2 * nbytes / (ms * 1e-3) / 1e9, 1)
    except Exception as ex:  # noqa: BLE001 -- a calibration probe never takes the line down
        out["error"] = f"{type(ex).__name__}: {ex}"
    return out


def pmc_traffic_per_launch(config, S, build_id):
    """Fabric bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of this
    same workload (tools/r06/pmc_fabric.sh: read requests by size, write requests, FETCH_SIZE / WRITE_SIZE,
    each in its own pass). bench.py cannot run the profiler on itself, so the figure is read back from
    profiles/ and labelled with its file; (None, None, None) when there is none for this workload."""
    name, heads = PMC_FILES.get((config, S), (None, ()))
    rel = f"profiles/{PMC_ROUND}_{name}" if name else None
    path = os.path.join(ROOT, rel) if rel else None
    if not path or not os.path.exists(path):
        return None, (f"no counter profile of this round for this workload ({rel})" if rel else None), None
    d = json.load(open(path))
    stamp = d.get("_stamp") or {}
    if stamp.get("build_id") != build_id:   # (kernels changed since the capture: the bytes are another build's)
        return None, f"REFUSED: {rel} was captured on build {stamp.get('build_id')}, this library is build {build_id}", None
    d = {k: v for k, v in d.items() if not k.startswith("_")}
    per_launch = raw = corrected = 0.0   # (several heads = the stages of ONE iteration, e.g. k_bin_scatter + k_bin_reduce: their bytes add up)
    for head in heads:
        rows = [v for k, v in d.items() if k.startswith(head)]
        launches = sum(v["launches"] for v in rows)
        if not launches:
            return None, None, None
        per_launch += sum(v["launches"] * v["fabric_bytes_per_launch"] for v in rows) / launches
        raw += sum(v["launches"] * ((v.get("fetch_size_raw_bytes_per_launch") or 0) + (v.get("write_size_bytes_per_launch") or 0)) for v in rows) / launches
        corrected += sum(v["launches"] * (v.get("hbm_bytes_per_launch_corrected") or 0) for v in rows) / launches
    detail = {"FETCH_SIZE_plus_WRITE_SIZE_raw": round(raw, 1), "two_x_FETCH_SIZE_plus_WRITE_SIZE": round(corrected, 1),
              "note": "rounds 1-3 reported 2 x FETCH_SIZE + WRITE_SIZE; FETCH_SIZE tallies 128-byte requests at 64, so the factor is right only "
                      "where every read request is a 128-byte one -- `traffic` counts the requests by size instead"}
    return round(per_launch, 1), f"committed profile {rel} (build {build_id}, commit {stamp.get('git_commit')})", detail


def cpu_baseline(V, e1, e2, directed, W, c, sources, eps, batches, p_end, stream_len):
    """CPU leg (kind "port"): the oracle's restatement of cpu/PPRCPUMTCilkRev, timed with the reference's scope
    (IncExecuteImpl only, cpu/PPRCPUMTCilk.h:131-137) on the same stream. `sources` (one or two of the rank's) are
    solved from scratch and followed through `batches` batches: with OpenMP workers (in place of Cilk Plus, which this
    toolchain lacks) up to the last two, which run at -t 1. When `p_end` is given (the GPU's p of those sources at the
    end of the timed region, `batches` = every batch up to there), the -t 1 state is compared with it."""
    from oracle import oracle as orc
    threads = max(1, min(orc.max_threads(), os.cpu_count() or 1, 16))
    g = orc.Graph(V, e1, e2, directed, W, c)
    states = [orc.State(V, s, eps) for s in sources]
    for st in states:            # from-scratch solve (untimed here, like INIT_GRAPH_CALC_TIME in the reference)
        st.cilk_init()
        st.cilk_main_loop_mt(g, 0, threads)
    t_mt = t_1 = 0.0
    n_mt = n_1 = done = 0
    for k in range(batches):
        if g.stream_updates():
            break
        g.inc_construct(1)
        done += 1
        serial = k >= batches - 2 or threads == 1
        for st in states:
            t = time.perf_counter()
            if serial:
                st.cilk_inc_execute(g)
            else:
                st.cilk_inc_execute_mt(g, threads)
            dt = time.perf_counter() - t
            if serial:
                t_1, n_1 = t_1 + dt, n_1 + 1
            else:
                t_mt, n_mt = t_mt + dt, n_mt + 1
    worst = None
    if p_end and done == batches:
        worst = max(float(np.max(np.abs(pe - st.p))) for pe, st in zip(p_end, states))
    had_mt = n_mt > 0
    if not had_mt:
        t_mt, n_mt = t_1, n_1
    return {"value": round(c * n_mt / t_mt, 1) if t_mt > 0 else None, "unit": "edges/s", "cores": threads if had_mt else 1,
            "kind": "port", "ms_per_step": round(1e3 * t_mt / max(n_mt, 1), 2),
            "t1_value": round(c * n_1 / t_1, 1) if t_1 > 0 else None, "t1_ms_per_step": round(1e3 * t_1 / max(n_1, 1), 2),
            "max_abs_dp": worst, "p_cpu": [np.array(st.p) for st in states] if done == batches else None,
            "compared": (f"p of {len(states)} source(s) after batch {done} (the end of the timed region), CPU state reached with the "
                         f"{threads}-thread port and the last two batches at -t 1") if worst is not None else None,
            "sample": f"{len(states)} source(s) ({', '.join(str(v) for v in sources)}; value is per source and batch -- the GPU value sums all of a "
                      f"rank's sources), {done} batches of the same stream after the from-scratch solve: {n_mt} source-batches with "
                      f"{threads} OpenMP workers, {n_1} at -t 1; oracle restatement of cpu/PPRCPUMTCilkRev, gcc -O2"}


def reference_fifo_baseline(bin_path, directed, flags, source, eps, c, batches=4):
    """The REAL reference code that can be built in the dev container: cpu/PPRCPURev.h, the reference's (deprecated)
    single-thread FIFO reverse push, compiled from /root/reference by oracle/Makefile into oracle/_ref/ref_driver.
    Timed scope: IncExecuteImpl only. The binary exists only where it was built: a box reached through gpurun from the
    dev container has it (oracle/_ref travels with the snapshot), a fresh checkout built by __graft_entry__.build() without
    /root/reference -- the driver's round-end box -- does not, and the key is then absent from the line (None here): this
    leg is an extra of development runs on small streams (< 10 M edges), never part of `cpu_baseline.value`."""
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
