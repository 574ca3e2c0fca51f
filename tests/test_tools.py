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


# ---------------------------------------------------------------------------------------------
# Pinned against the REAL reference tools: tests/golden/tools_toy.npz holds what encoder/ and
# workload/ of the reference (compiled from /root/reference, tests/golden/make_tools_golden.py)
# wrote for a toy SNAP edge list.
# ---------------------------------------------------------------------------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tools_toy.npz")


def _toy(tmp_path):
    d = np.load(GOLD)
    txt = tmp_path / "toy.txt"
    with open(txt, "w") as f:
        for a, b in zip(d["snap.src"], d["snap.dst"]):
            f.write(f"{a}\t{b}\n")
    return d, str(txt)


def test_glibc_rand_sequence():
    """First outputs of glibc's rand() after srand(1) (the documented sequence every unseeded program sees)."""
    g = tools.GlibcRand(1)
    assert [g.rand() for _ in range(5)] == [1804289383, 846930886, 1681692777, 1714636915, 1957747793]


def test_encoder_output_is_byte_identical_to_the_reference_encoder(tmp_path):
    d, txt = _toy(tmp_path)
    for rev, tag in ((False, "fwd"), (True, "rev")):
        out = tools.encode_snap(txt, str(tmp_path / f"{tag}.bin"), reverse=rev, shuffle="glibc")
        V, e1, e2 = datagen.read_bin(out)
        assert V == int(d[f"bin.{tag}.V"][0]) == int(max(d["snap.src"].max(), d["snap.dst"].max())
                                                      - min(d["snap.src"].min(), d["snap.dst"].min()) + 1)
        assert np.array_equal(e1, d[f"bin.{tag}.e1"]) and np.array_equal(e2, d[f"bin.{tag}.e2"])   # order included
    # the default (numpy) shuffle gives the same rebased multiset
    out = tools.encode_snap(txt, str(tmp_path / "np.bin"), seed=9)
    _, e1, e2 = datagen.read_bin(out)
    assert sorted(zip(e1.tolist(), e2.tolist())) == sorted(zip(d["bin.fwd.e1"].tolist(), d["bin.fwd.e2"].tolist()))


def test_workload_files_against_the_reference_workload_tool(tmp_path):
    d, _ = _toy(tmp_path)
    V = int(d["bin.fwd.V"][0])
    binp = str(tmp_path / "toy.bin")
    datagen.write_bin(binp, V, d["bin.fwd.e1"], d["bin.fwd.e2"])
    for directed in (1, 0):
        for window in (0, 1):
            for outdeg in (1, 0):
                ours = tools.workload(binp, directed, window, outdeg, seed=3, out_dir=str(tmp_path))
                n = int(len(d["bin.fwd.e1"]) * 0.1) if window else len(d["bin.fwd.e1"])
                deg, in_deg = tools.degrees(V, d["bin.fwd.e1"][:n], d["bin.fwd.e2"][:n], directed)
                cmp_deg = deg if outdeg else in_deg
                ranked = np.sort(cmp_deg)[::-1]
                ref10 = d[f"wl.d{directed}.w{window}.o{outdeg}.10"]
                # top10: the reference's std::sort leaves ties in unspecified order -- the degree sequence is exact,
                # and so are the ids wherever the degree is unique
                assert np.array_equal(cmp_deg[ref10], ranked[:10])
                assert np.array_equal(cmp_deg[ours[10][1]], ranked[:10])
                for i in range(10):
                    if np.count_nonzero(cmp_deg == ranked[i]) == 1:
                        assert ours[10][1][i] == ref10[i]
                # sampled files: 10 distinct connected ids whose degree lies in the rank range's degree band
                for count, (lo, hi) in ((1000, (10, 1000)), (1000000, (1000, 1000000))):
                    hi = min(hi, V)
                    band = (ranked[hi - 1], ranked[lo])
                    for ids in (d[f"wl.d{directed}.w{window}.o{outdeg}.{count}"], ours[count][1]):
                        assert len(ids) == 10 and len(set(ids.tolist())) == 10
                        assert np.all((cmp_deg[ids] >= band[0]) & (cmp_deg[ids] <= band[1]))
                        assert np.all(deg[ids] > 0) and np.all(in_deg[ids] > 0)
