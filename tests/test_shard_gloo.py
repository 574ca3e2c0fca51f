"""The N>1 path on CPU: two gloo ranks run the same sharding / barrier / MAX-over-ranks /
SUM-of-units code bench.py uses, with the CPU oracle standing in for the device engine."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from dynamicppr_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_assign_sources_round_robin():
    src = list(range(100, 110))
    got = [shard.assign_sources(src, r, 8) for r in range(8)]
    assert sorted(sum(got, [])) == src                     # config 5: 10 sources over 8 GPUs (2+2+1x6)
    assert [len(g) for g in got] == [2, 2, 1, 1, 1, 1, 1, 1]
    one = [shard.assign_sources(src, r, 8, per_rank=1) for r in range(8)]
    assert [g[0] for g in one] == src[:8]                  # config 4: one top-10 source per GPU
    assert shard.assign_sources(src[:2], 3, 4, per_rank=2) == [src[1], src[1]]
    assert [len(shard.assign_sources(src[:8], r, 4)) for r in range(4)] == [2, 2, 2, 2]   # config 4 on 4 GPUs
    assert [len(shard.assign_sources(src, r, 4)) for r in range(4)] == [3, 3, 2, 2]
    assert shard.CONFIG_SOURCE_SETS == {"twitter": (8, "top10"), "friendster": (10, "top1000")}


WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch.distributed as dist
    from dynamicppr_amd import datagen, shard, stream as st
    from oracle import oracle as orc
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # the stream: rank 0 writes the stand-in prefix, everybody waits at the barrier and reads the same file
    # (what bench.py does for --gpus N); sources: every rank draws ITS ids from ranks [10, 1000) (a top1000 file)
    if rank == 0:
        datagen.ensure_stand_in("dblp", {out!r}, 6000)
    dist.barrier()
    path = datagen.ensure_stand_in("dblp", {out!r}, 6000)
    assert rank == 0 or datagen.PROVENANCE[path]["origin"] == "cached"
    V, e1, e2 = datagen.read_bin(path)
    wl = st.workload_config(len(e1), 0.1, 0, 0.01, 5)
    # bench.py --config friendster --gpus 2: the configuration's OWN 10 sources (a top1000 file), dealt round-robin
    pool = shard.config_source_pool("friendster", V, e1, e2, wl.window, 1)
    mine = shard.assign_sources(pool, rank, world)
    checksum = datagen.PROVENANCE[path]["checksum"]
    g = orc.Graph(V, e1, e2, 1, wl.window, wl.per_batch)
    states = [orc.State(V, v, 1e-9) for v in mine]
    for s in states:
        s.cilk_execute(g)
    steps = 3
    def run():
        for _ in range(steps):
            assert not g.stream_updates()
            g.inc_construct(1)
            for s in states:
                s.cilk_inc_execute(g)
        time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
    dt, w = shard.timed_region(run, lambda: None, dist)
    units = shard.aggregate_units(wl.per_batch * steps * len(mine), dist)
    seen, per_rank, backend = shard.rank_census(shard.timed_region.last_local, dist)   # the proof that N ranks ran (bench.py: ranks_seen)
    blocks = shard.line_blocks(rank, w)   # what bench.py's line carries at this rank / world size
    # the rolling-ring form of the timed region (bench.py default): per batch an UNTIMED graph update, then the timed step in
    # its own bracket; the seconds are the SUM of the brackets, MAX over ranks; the barrier pair encloses the whole run
    acc = [0.0]
    def run_ring():
        for _ in range(2):
            time.sleep(0.04)                       # the untimed part
            t = time.perf_counter()
            time.sleep(0.03 * (rank + 1))          # the timed step (rank 1 is the slow one)
            acc[0] += time.perf_counter() - t
    dt_ring, _ = shard.timed_region(run_ring, lambda: None, dist, summed=acc)
    ring = dict(dt=dt_ring, wall=shard.timed_region.last_wall, own=shard.timed_region.last_local, ok=shard.aggregate_min(1 if rank == 0 else 0, dist))
    json.dump(dict(rank=rank, ring=ring, world=w, sources=mine, pool=pool, dt=dt, units=units, seen=seen, per_rank=per_rank, backend=backend, psum=float(sum(s.p.sum() for s in states)), checksum=checksum, blocks=blocks),
              open(os.path.join({out!r}, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_run(tmp_path):
    import json
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:]
    a = json.load(open(tmp_path / "rank0.json"))
    b = json.load(open(tmp_path / "rank1.json"))
    assert a["world"] == b["world"] == 2
    assert a["pool"] == b["pool"] and len(set(a["pool"])) == 10      # every rank derives the same source set ...
    assert a["sources"] == a["pool"][0::2] and b["sources"] == a["pool"][1::2]   # ... and takes its round-robin share: 5 + 5
    assert a["dt"] == b["dt"] and a["dt"] >= 0.1            # MAX over ranks: the slow rank's time
    assert a["units"] == b["units"] == 10 * 6 * 3           # SUM over ranks of sources * c * steps
    # the rank census: both ranks answered, every rank's own bracket time in rank order, the MAX of them is the reported time
    assert a["seen"] == b["seen"] == 2 and a["backend"] == "gloo" and a["per_rank"] == b["per_rank"] and len(a["per_rank"]) == 2
    # (a rank's own time ends when ITS steps are done, the bracket at the closing barrier: rank 1 sleeps 0.05 s longer and sets the MAX)
    assert a["per_rank"][1] > a["per_rank"][0] + 0.03 and 0 <= a["dt"] - max(a["per_rank"]) < 0.05
    assert shard.rank_census(1.5) == (1, [1.5], None)
    # rolling ring: the reported seconds are the slow rank's SUM of brackets (2 x 0.06 s), not the wall (which holds 2 x 0.04 s of untimed work more)
    ra, rb = a["ring"], b["ring"]
    assert ra["dt"] == rb["dt"] and 0.12 <= ra["dt"] < 0.16 and abs(rb["own"] - rb["dt"]) < 1e-9 and 0.06 <= ra["own"] < 0.09
    assert ra["wall"] == rb["wall"] and ra["wall"] >= ra["dt"] + 0.08
    assert ra["ok"] == rb["ok"] == 0 and shard.aggregate_min(1) == 1        # a decision every rank takes the same way: MIN
    assert a["psum"] != b["psum"]
    assert a["checksum"] == b["checksum"]                   # one stream file, written once
    # the N = 2 line is rank 0's and is as complete as the N = 1 line: parity, roofline and the CPU baseline (only the
    # merged-loop side metric is an N = 1 extra); rank 1 prints nothing
    assert a["blocks"] == {"line": True, "parity": True, "roofline": True, "cpu_baseline": True, "merged_loop": False}
    assert not any(b["blocks"].values())
    assert shard.line_blocks(0, 1)["merged_loop"] and not shard.line_blocks(0, 1, no_cpu_baseline=True)["cpu_baseline"]
