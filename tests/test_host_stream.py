"""The PRODUCT's host batch builders against the reference's own output.

tests/golden/*.npz hold, per scenario, what the real reference's SlidingGraphVec.h produced
(oracle/_ref/ref_driver): the derived workload, every batch's EdgeBatch (b{k}.batch.*) and
new_stream (b{k}.new.*). Both host sides of this repository must reproduce them bit for bit:
  * C++ host/graph_vec.hpp (SlidingGraphVec::PrepareSlidingGraph / StreamUpdates /
    SerializeEdgeStream, what ./pagerank feeds the engine with), driven by host/dump_batches;
  * Python dynamicppr_amd/stream.py (bench.py and the tests' plumbing).
Reference: SlidingGraphVec.h:46-66 (workload), :201-217 (window), :219-275 (batches). CPU only."""
import os
import struct
import subprocess

import numpy as np
import pytest

from dynamicppr_amd import datagen, stream as st
from tests.util import golden_names, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dynamicppr_amd", "host")
DUMP = os.path.join(HOST, "dump_batches")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-C", HOST, "-s", "dump"])


def parse(path):
    out, data, off = {}, open(path, "rb").read(), 0
    while off < len(data):
        (ln,) = struct.unpack_from("<I", data, off); off += 4
        name = data[off:off + ln].decode(); off += ln
        (n,) = struct.unpack_from("<Q", data, off); off += 8
        out[name] = np.frombuffer(data, dtype="<i4", count=n, offset=off).copy()
        off += 4 * n
    return out


def golden_flags(d):
    """The flag list the reference was run with, minus the input path (replaced by ours)."""
    flags = str(d["flags"]).split() if d["flags"].shape == () else " ".join(map(str, d["flags"])).split()
    i = flags.index("-d")
    return flags[:i] + flags[i + 2:]


@pytest.mark.parametrize("name", golden_names())
def test_cpp_host_graph_reproduces_reference_batches(name, tmp_path):
    d, m = load_golden(name)
    binf, out = str(tmp_path / "g.bin"), str(tmp_path / "dump.rec")
    datagen.write_bin(binf, m["V"], d["stream.e1"], d["stream.e2"])
    subprocess.check_call([DUMP, "-d", binf, *golden_flags(d), "--out", out], stdout=subprocess.DEVNULL)
    got = parse(out)
    assert got["config"].tolist() == [m["V"], m["W"], m["c"], m["batches"], m["total"], m["edge_count"]]
    assert np.array_equal(got["window.e1"], d["stream.e1"][:m["W"]])
    assert np.array_equal(got["window.e2"], d["stream.e2"][:m["W"]])
    assert int(got["batches_done"][0]) == m["done"]
    for k in range(1, m["done"] + 1):
        for key in ("batch.e1", "batch.e2", "batch.ins", "new.e1", "new.e2", "new.ins"):
            assert np.array_equal(got[f"b{k}.{key}"], d[f"b{k}.{key}"].astype(np.int32)), (k, key)


@pytest.mark.parametrize("name", golden_names())
def test_python_stream_reproduces_reference_batches(name):
    d, m = load_golden(name)
    f = golden_flags(d)
    opt = {f[i]: f[i + 1] for i in range(0, len(f) - 1) if f[i].startswith("-") and not f[i][1:2].isdigit()}
    wl = st.workload_config(len(d["stream.e1"]), float(opt.get("-w", 0.1)), int(opt.get("-n", 0)),
                            float(opt.get("-r", -1.0)), int(opt.get("-b", 0)), int(opt.get("-c", 0)), int(opt.get("-l", 0)))
    assert (wl.window, wl.per_batch, wl.batch_count, wl.total) == (m["W"], m["c"], m["batches"], m["total"])
    ss = st.SlidingStream(m["V"], d["stream.e1"], d["stream.e2"], m["directed"], wl)
    w1, w2 = ss.serialize_edge_stream()
    assert np.array_equal(w1, d["stream.e1"][:m["W"]]) and np.array_equal(w2, d["stream.e2"][:m["W"]])
    done = 0
    for k in range(1, wl.batch_count + 1):
        if ss.stream_updates():
            break
        done += 1
        b1, b2, ins = ss.batch_arrays()
        n1, n2 = ss.new_arrays()
        assert np.array_equal(b1, d[f"b{k}.batch.e1"]) and np.array_equal(b2, d[f"b{k}.batch.e2"]), k
        assert np.array_equal(ins, d[f"b{k}.batch.ins"].astype(np.uint8)), k
        assert np.array_equal(n1, d[f"b{k}.new.e1"]) and np.array_equal(n2, d[f"b{k}.new.e2"]), k
        w1, w2 = ss.serialize_edge_stream()     # the window the device builder is handed after the slide
        lo = m["W"] + (k - 1) * m["c"] + m["c"] - m["W"]
        assert np.array_equal(w1, d["stream.e1"][lo:lo + m["W"]])
    assert done == m["done"]
