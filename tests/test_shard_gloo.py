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
    mine = [int(x) for x in datagen.ranked_sources(V, e1, e2, wl.window, 1, 10, 200, 10, seed=1 + rank)[:1]]
    checksum = datagen.PROVENANCE[path]["checksum"]
    g = orc.Graph(V, e1, e2, 1, wl.window, wl.per_batch)
    s = orc.State(V, mine[0], 1e-9)
    s.cilk_execute(g)
    steps = 3
    def run():
        for _ in range(steps):
            assert not g.stream_updates()
            g.inc_construct(1)
            s.cilk_inc_execute(g)
        time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
    dt, w = shard.timed_region(run, lambda: None, dist)
    units = shard.aggregate_units(wl.per_batch * steps * len(mine), dist)
    json.dump(dict(rank=rank, world=w, source=mine[0], dt=dt, units=units, psum=float(s.p.sum()), checksum=checksum),
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
    assert a["source"] != b["source"]                       # independent sources, one per rank
    assert a["dt"] == b["dt"] and a["dt"] >= 0.1            # MAX over ranks: the slow rank's time
    assert a["units"] == b["units"] == 2 * 6 * 3            # SUM over ranks of c * steps
    assert a["psum"] != b["psum"]
    assert a["checksum"] == b["checksum"]                   # one stream file, written once
