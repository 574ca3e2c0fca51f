"""f3: encoder / workload tools (encoder/GraphEncoder.h, workload/) restated in dynamicppr_amd/tools.py."""
import os

import numpy as np

from dynamicppr_amd import datagen, tools


def test_encode_snap_round_trip(tmp_path):
    txt = tmp_path / "toy.ungraph.txt"
    txt.write_text("# Undirected graph\n# Nodes: 5 Edges: 6\n# FromNodeId\tToNodeId\n"
                   "10\t12\n10 15\n12 15\n15 17\n17 10\n12 17\n")
    out = tools.encode_snap(str(txt), str(tmp_path / "toy.ungraph.bin"), seed=3)
    V, e1, e2 = datagen.read_bin(out)
    assert V == 17 - 10 + 1                      # the id RANGE, isolated ids included (GraphEncoder.h:44)
    got = sorted(zip(e1.tolist(), e2.tolist()))
    assert got == sorted([(0, 2), (0, 5), (2, 5), (5, 7), (7, 0), (2, 7)])   # rebased to id - min_id
    # shuffled but reproducible
    again = tools.encode_snap(str(txt), str(tmp_path / "again.bin"), seed=3)
    assert open(out, "rb").read() == open(again, "rb").read()
    other = tools.encode_snap(str(txt), str(tmp_path / "other.bin"), seed=4)
    assert sorted(zip(*datagen.read_bin(other)[1:])) == sorted(zip(e1, e2))
    # reverse: endpoints swapped (the "_rev" datasets)
    rev = tools.encode_snap(str(txt), str(tmp_path / "rev.bin"), reverse=True, randomize=False)
    _, r1, r2 = datagen.read_bin(rev)
    assert list(zip(r1.tolist(), r2.tolist()))[0] == (2, 0)
    rb = tools.reverse_bin(rev, str(tmp_path / "revrev.bin"))
    _, q1, q2 = datagen.read_bin(rb)
    assert np.array_equal(q1, r2) and np.array_equal(q2, r1)
    # default output name: <basename without .txt>.bin / _rev.bin in the working directory
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert tools.encode_snap(str(txt)) == "toy.ungraph.bin"
        assert tools.encode_snap(str(txt), reverse=True) == "toy.ungraph_rev.bin"
    finally:
        os.chdir(cwd)


def test_workload_id_files(tmp_path):
    V, e1, e2 = datagen.rmat_stream(12, 40000, 5)
    binp = str(tmp_path / "syn.bin")
    datagen.write_bin(binp, V, e1, e2)
    out = tools.workload(binp, directed=1, is_window=0, is_out_degree=1, seed=2, out_dir=str(tmp_path))
    assert sorted(out) == [10, 1000, 1000000]
    deg, in_deg = tools.degrees(V, e1, e2, 1)
    order = np.lexsort((np.arange(V), -deg))
    top10 = np.loadtxt(out[10][0], dtype=np.int64)
    assert out[10][0].endswith("syn.bin_top10.txt") and np.array_equal(top10, order[:10])
    rank = np.empty(V, dtype=np.int64); rank[order] = np.arange(V)
    for count, (lo, hi) in ((1000, (10, 1000)), (1000000, (1000, V))):
        ids = np.loadtxt(out[count][0], dtype=np.int64)
        assert len(ids) == 10 and len(set(ids.tolist())) == 10
        assert np.all((rank[ids] >= lo) & (rank[ids] < hi))
        assert np.all(deg[ids] > 0) and np.all(in_deg[ids] > 0)      # "choose the connected ones"
    # window + in-degree variants use the reference's file-name features
    out = tools.workload(binp, directed=0, is_window=1, is_out_degree=0, seed=2, out_dir=str(tmp_path))
    assert out[10][0].endswith("syn.bin_topwindowrev10.txt")
    wdeg, _ = tools.degrees(V, e1[:4000], e2[:4000], 0)
    assert wdeg[np.loadtxt(out[10][0], dtype=np.int64)[0]] == wdeg.max()
