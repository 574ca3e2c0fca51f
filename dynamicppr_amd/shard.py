"""Source sharding across GPUs: independent source vertices are dealt
round-robin over ranks, every rank holds a full replica of the window graph, and there is
no data-path collective. torch.distributed is used only for the barrier around the timed
region and for reducing the bracket time (MAX) and the processed units (SUM)."""
from __future__ import annotations

import time
from typing import Callable, Sequence


def assign_sources(sources: Sequence[int], rank: int, world: int, per_rank: int | None = None) -> list[int]:
    """Round-robin deal (./pagerank -g N does the same). With ``per_rank`` every rank gets
    exactly that many (weak scaling: config 4 of BASELINE.json = one top-10 source per GPU),
    wrapping around the list if there are fewer sources than slots."""
    if per_rank is None:
        return [int(s) for i, s in enumerate(sources) if i % world == rank]
    return [int(sources[(rank + k * world) % len(sources)]) for k in range(per_rank)]


# The 8-GPU configurations of BASELINE.json are FIXED source sets: configs[3] twitter-2010 = 8 of the top-10 file,
# configs[4] com-friendster = the 10 ids of a top1000 file. bench.py deals them over its ranks with
# assign_sources(pool, rank, world): 8 / 4+4 / 2+2+2+2 / 1 each, and 10 / 5+5 / 3+3+2+2 / 2+2+1x6.
CONFIG_SOURCE_SETS = {"twitter": (8, "top10"), "friendster": (10, "top1000")}


def config_source_pool(key: str, V: int, e1, e2, W: int, directed: int) -> list[int]:
    """The source set of an 8-GPU configuration on the given stream (every rank computes the same list)."""
    from . import datagen
    n, pick = CONFIG_SOURCE_SETS[key]
    ids = (datagen.ranked_sources(V, e1, e2, W, directed, 10, 1000, 10, seed=1) if pick == "top1000"
           else datagen.top_sources(V, e1, e2, W, directed, 10))
    return [int(x) for x in ids[:n]]


def timed_region(run_steps: Callable[[], None], device_sync: Callable[[], None], dist=None, summed=None):
    """Barrier + device sync on both sides of ``run_steps``; returns (max-over-ranks seconds, world).

    ``summed`` is None: the seconds are the wall time of the whole bracket (the K steps back to back -- the pre-staged form).
    ``summed`` is a one-element list: ``run_steps`` alternates an UNTIMED graph update with a timed step per batch, the way the
    reference's loop does (gpu/PPRGPU.cuh:109-169), and adds every step's own synchronize-to-synchronize bracket to
    ``summed[0]``; the seconds are then that SUM (MAX over ranks), the barrier pair still encloses the whole run and
    ``timed_region.last_wall`` holds its wall time (graph updates included; MAX over ranks)."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1

    def fence():
        device_sync()
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    fence()
    t0 = time.perf_counter()
    run_steps()
    device_sync()
    own = time.perf_counter() - t0
    # this rank's OWN steps, its device drained, before it waits for the others (rank_census reports every rank's beside the MAX)
    timed_region.last_local = own if summed is None else float(summed[0])
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    timed_region.last_wall = max_over_ranks(wall)
    dt = timed_region.last_wall if summed is None else max_over_ranks(float(summed[0]))
    return dt, world


timed_region.last_local = 0.0
timed_region.last_wall = 0.0


def rank_census(value_local: float, dist=None):
    """Proof that N ranks ran: every rank contributes a 1 (all-reduce SUM -> ranks_seen) and its own value
    (all-gather, rank order), over whatever backend carries the barrier. Returns (ranks_seen, [values], backend)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 1, [float(value_local)], None
    import torch
    backend = dist.get_backend()
    dev = "cuda" if backend == "nccl" else "cpu"
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    mine = torch.tensor([value_local], dtype=torch.float64, device=dev)
    every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return int(one.item()), [float(t.item()) for t in every], backend


def aggregate_units(units_local: int, dist=None) -> int:
    """Whole-job units (edge updates) = SUM over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(units_local)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([units_local], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def aggregate_min(flag_local: int, dist=None) -> int:
    """A decision every rank must take the same way (e.g. "is there room for the extra block?"): MIN over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(flag_local)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(flag_local)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def line_blocks(rank: int, world: int, no_cpu_baseline: bool = False, no_merged: bool = False, schedule: str = "eager") -> dict:
    """Which blocks the ONE JSON line of bench.py carries, decided in one place so that the N > 1 line is not thinner than
    the N = 1 line: rank 0 prints the line at every N, with `parity` (every source of rank 0, CPU comparison
    included) and `cpu_baseline` (timed on rank 0's host cores while the other ranks wait at the closing barrier);
    `roofline` is rank 0's device; only the merged-loop side metric stays an N = 1 extra (a second pass over the run)."""
    head = rank == 0
    return {"line": head, "parity": head, "roofline": head, "cpu_baseline": head and not no_cpu_baseline,
            "merged_loop": head and world == 1 and not no_merged and schedule == "eager"}
