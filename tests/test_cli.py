"""./pagerank (C++ host over the C ABI): flag contract on CPU, end-to-end on the GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

from dynamicppr_amd import datagen
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dynamicppr_amd", "host", "pagerank")


@pytest.fixture(scope="module")
def pagerank():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dynamicppr_amd", "host"), "-s", "all"])
    return BIN


@pytest.fixture(scope="module")
def small_bin(tmp_path_factory):
    V, e1, e2 = datagen.rmat_stream(9, 6000, 11)
    path = str(tmp_path_factory.mktemp("data") / "syn.bin")
    datagen.write_bin(path, V, e1, e2)
    return path, V, e1, e2


WATCHDOG_S = "60"   # ./pagerank ends itself with a post-mortem (dppr_debug_dump of every engine, exit code 125) after this long without progress


def run(args, env_extra=None, **kw):
    """./pagerank as a child process under its own watchdog (DPPR_WATCHDOG_S) and a harder limit here: whatever the child
    printed goes into the pytest log if either fires (round 3 lost the output of its one unexplained 300-second guard)."""
    env = dict(os.environ, DPPR_WATCHDOG_S=WATCHDOG_S, **(env_extra or {}))
    try:
        r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=150, env=env, **kw)
    except subprocess.TimeoutExpired as ex:   # (subprocess.run has killed the child -- the CHILD, never this process)
        out = ex.stdout.decode(errors="replace") if isinstance(ex.stdout, bytes) else (ex.stdout or "")
        pytest.fail(f"{' '.join(args)} did not finish within 150 s (its own watchdog should have fired at {WATCHDOG_S} s); output so far:\n{out}")
    if r.returncode in (124, 125):
        pytest.fail(f"{' '.join(args)}: the watchdog fired -- no progress for {WATCHDOG_S} s:\n{r.stdout}")
    return r


def test_invalid_arguments_print_usage_and_exit(pagerank, small_bin):
    path = small_bin[0]
    for bad in ([], ["-d", path, "-a", "0", "-i", "1", "-y", "1"],            # no workload flags
                ["-d", path, "-a", "1", "-i", "1", "-y", "1", "-r", "0.01", "-b", "5"],  # gAppType out of range
                ["-d", path, "-a", "0", "-i", "1", "-y", "1", "-n", "1", "-c", "0", "-l", "10"],
                ["-d", path, "-a", "0", "-i", "1", "-y", "0", "-r", "0.01", "-b", "5"],   # static mode
                ["-d", path, "-a", "0", "-i", "1", "-y", "1", "-r", "0.01", "-b", "5", "-o", "4"]):   # no such variant
        r = run([pagerank] + bad)
        assert r.returncode != 0
        assert "invalid arguments" in r.stdout and "[USAGE]" in r.stdout
    r = run([pagerank, "-d", path, "-a", "0", "-i", "1", "-y", "1", "-r", "0.01", "-b"])  # dangling flag
    assert r.returncode != 0 and "missing value" in r.stdout


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_no_device_is_a_loud_failure(pagerank, small_bin):
    r = run([pagerank, "-d", small_bin[0], "-a", "0", "-i", "1", "-y", "1", "-n", "0", "-r", "0.01", "-b", "3", "-s", "1"])
    assert r.returncode != 0
    assert "no CPU fallback" in r.stdout
    # the workload derivation is printed before the device is touched (SlidingGraphVec.h:67-71)
    assert "sliding window size=600,gStreamUpdateCountPerBatch=6" in r.stdout


def read_dump(path):
    out = {}
    raw = open(path, "rb").read()
    off = 0
    while off < len(raw):
        s, V = np.frombuffer(raw, dtype="<i4", count=2, offset=off)
        off += 8
        p = np.frombuffer(raw, dtype="<f8", count=V, offset=off); off += 8 * V
        r = np.frombuffer(raw, dtype="<f8", count=V, offset=off); off += 8 * V
        out[int(s)] = (p, r)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("directed", [1, 0])
@pytest.mark.parametrize("extra", [[], ["--split"], ["--sync"], ["-o", "1"], ["-o", "2"], ["-o", "3"], ["--merge-phases"], ["-o", "2", "--push-only"], ["-o", "3", "--push-only"]])
def test_cli_end_to_end_matches_oracle(pagerank, small_bin, tmp_path, directed, extra):
    path, V, e1, e2 = small_bin
    W, c = 600, 6
    src = int(datagen.top_sources(V, e1, e2, W, directed, 1)[0])
    dump = str(tmp_path / "out.dump")
    r = run([pagerank, "-d", path, "-a", "0", "-i", str(directed), "-y", "1", "-w", "0.1", "-n", "0", "-r", "0.01",
             "-b", "5", "-s", str(src), "--validate", "--dump", dump] + extra)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("validate ok") == 6          # after init and after each of 5 batches
    # stdout contract scraped by scripts/extract_gpu.py (last occurrence wins)
    last = {k: float(v) for k, v in re.findall(r"^(ppr_time|edge_num|ppr_latency|ppr_throughput) ([-+.e\d]+)$",
                                               r.stdout, flags=re.M)}
    assert last["edge_num"] == c * 5
    assert abs(last["ppr_latency"] - last["ppr_time"] / 5) < 1e-3
    assert "coming stream_batch_count=6" in r.stdout    # batches_done + 1, gpu/PPRGPU.cuh:112,170
    g = orc.Graph(V, e1, e2, directed, W, c)
    s = orc.State(V, src, 1e-9)
    s.cilk_execute(g)
    for _ in range(5):
        assert not g.stream_updates()
        g.inc_construct(1)
        s.cilk_inc_execute(g)
    p, rr = read_dump(dump)[src]
    assert np.max(np.abs(p - s.p)) < 1e-9 and np.max(np.abs(rr)) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("directed", [1, 0])
def test_cli_three_loops_agree(pagerank, small_bin, tmp_path, directed):
    """./pagerank's batch loop in its three forms -- overlapped (default: the graph of batch k + 1 built by a helper thread through
    dppr_slide_concurrent while batch k is solved), serial with the id lookahead of round 4 (DPPR_NO_OVERLAP=1), plain serial
    (+ DPPR_NO_LOOKAHEAD=1) -- on the synchronous schedule: the same stdout contract, the same p / r to rounding, the oracle's."""
    path, V, e1, e2 = small_bin
    W, c, nb = 600, 24, 12
    src = int(datagen.top_sources(V, e1, e2, W, directed, 1)[0])
    outs = {}
    for name, env in (("overlap", {}), ("lookahead", {"DPPR_NO_OVERLAP": "1"}), ("serial", {"DPPR_NO_OVERLAP": "1", "DPPR_NO_LOOKAHEAD": "1"})):
        dump = str(tmp_path / f"{name}.dump")
        r = run([pagerank, "-d", path, "-a", "0", "-i", str(directed), "-y", "1", "-w", "0.1", "-n", "1", "-c", str(c), "-l", str(c * nb),
                 "-s", str(src), "--sync", "--dump", dump], env_extra=dict(env, DPPR_HOST_TIMES="1"))
        assert r.returncode == 0, r.stdout
        assert f"overlap={1 if name == 'overlap' else 0}" in r.stdout          # (stderr is merged into stdout by run())
        last = {k: float(v) for k, v in re.findall(r"^(edge_num|ppr_latency) ([-+.e\d]+)$", r.stdout, flags=re.M)}
        assert last["edge_num"] == c * nb and f"coming stream_batch_count={nb + 1}" in r.stdout
        outs[name] = read_dump(dump)[src]
    g = orc.Graph(V, e1, e2, directed, W, c)
    s = orc.State(V, src, 1e-9)
    s.sync_execute(g)
    for _ in range(nb):
        assert not g.stream_updates()
        g.inc_construct(1)
        s.sync_inc_execute(g)
    for name, (p, r_) in outs.items():
        assert np.max(np.abs(p - s.p)) < 1e-14 and np.max(np.abs(r_ - s.r)) < 1e-14, name


@pytest.mark.gpu
@pytest.mark.parametrize("nsrc", [3, 10, 18])
@pytest.mark.parametrize("extra", [[], ["--no-groups"], ["--validate"]])
def test_cli_multiple_sources_one_gpu(pagerank, small_bin, tmp_path, extra, nsrc):
    """Several sources on one device: solved up to 16 together as a source group by default (3: one
    8-wide group, 10: one 16-wide group, 18: a 16-wide and an 8-wide one)."""
    path, V, e1, e2 = small_bin
    srcs = [int(x) for x in datagen.top_sources(V, e1, e2, 600, 1, nsrc)]
    sf = tmp_path / "sources.txt"
    sf.write_text("\n".join(map(str, srcs)) + "\n")
    dump = str(tmp_path / "out.dump")
    r = run([pagerank, "-d", path, "-a", "0", "-i", "1", "-y", "1", "-n", "1", "-c", "7", "-l", "21",
             "--sources", str(sf), "--dump", dump, "-g", "1"] + extra)
    assert r.returncode == 0, r.stdout
    assert "aggregate_edge_num %d" % (7 * 3 * nsrc) in r.stdout
    got = read_dump(dump)
    assert sorted(got) == sorted(srcs)
    for sv in srcs:
        g = orc.Graph(V, e1, e2, 1, 600, 7)
        s = orc.State(V, sv, 1e-9)
        s.cilk_execute(g)
        for _ in range(3):
            assert not g.stream_updates()
            g.inc_construct(1)
            s.cilk_inc_execute(g)
        assert np.max(np.abs(got[sv][0] - s.p)) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("directed", [1, 0])
def test_cli_lookahead_on_and_off_give_the_same_run(pagerank, small_bin, tmp_path, directed):
    """The id lookahead of the batch loop (dppr_hint_next_batch from a helper thread during the update; default without --validate)
    against DPPR_NO_LOOKAHEAD=1: the same lines on stdout apart from the timings, the same vectors; with a stream long enough
    that batches keep bringing vertices without an id (which the lookahead leaves to the call)."""
    path, V, e1, e2 = small_bin
    srcs = [int(x) for x in datagen.top_sources(V, e1, e2, 600, directed, 3)]
    sf = tmp_path / "sources.txt"
    sf.write_text("\n".join(map(str, srcs)) + "\n")
    outs, dumps = [], []
    for k, env in enumerate(({}, {"DPPR_NO_LOOKAHEAD": "1"})):
        dump = str(tmp_path / f"out{k}.dump")
        r = run([pagerank, "-d", path, "-a", "0", "-i", str(directed), "-y", "1", "-n", "1", "-c", "40", "-l", "1200",
                 "--sources", str(sf), "--dump", dump, "-o", "1"], env_extra=env)
        assert r.returncode == 0, r.stdout
        timing = ("elapsed time=", "ppr_time", "ppr_latency", "ppr_throughput", "aggregate_ppr", "wall_ms")
        outs.append([l for l in r.stdout.splitlines() if not l.startswith(timing)])
        dumps.append(read_dump(dump))
    assert outs[0] == outs[1]
    assert "coming stream_batch_count=31" in "\n".join(outs[0])
    for sv in srcs:      # (-o 1: the synchronous schedule -- only the arrival order of float atomics differs between two runs)
        assert np.max(np.abs(dumps[0][sv][0] - dumps[1][sv][0])) < 1e-14 and np.max(np.abs(dumps[0][sv][1] - dumps[1][sv][1])) < 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("ngpu,nsrc", [(2, 5), (3, 10)])
def test_cli_device_threads_share_the_device(pagerank, small_bin, tmp_path, ngpu, nsrc):
    """`./pagerank -g N --sources f --share-device`: the thread-per-device flow of pagerank_main.cpp (N host threads, each with
    its own SlidingGraphVec, engine, stream and replica of the window graph; sources dealt round-robin) on the devices that
    exist -- one on this pool. Every device thread's dump against the oracle, and the aggregate_* lines against the
    per-device figures."""
    path, V, e1, e2 = small_bin
    W, c = 600, 6
    sources = [int(x) for x in datagen.top_sources(V, e1, e2, W, 0, nsrc)]
    srcfile = tmp_path / "sources.txt"
    srcfile.write_text("\n".join(str(s) for s in sources) + "\n")
    dump = str(tmp_path / "out.bin")
    r = run([pagerank, "-d", path, "-a", "0", "-i", "0", "-y", "1", "-n", "0", "-r", "0.01", "-b", "6", "-g", str(ngpu), "--share-device",
             "--sources", str(srcfile), "--dump", dump])
    assert r.returncode == 0, r.stdout
    assert f"gpus {ngpu} sources {nsrc}" in r.stdout
    kv = {k: float(v) for k, v in re.findall(r"^(aggregate_\w+|wall_ms) ([-+0-9.e]+)$", r.stdout, re.M)}
    assert kv["aggregate_edge_num"] == c * 6 * nsrc
    assert kv["aggregate_ppr_time_slowest_gpu"] > 0
    assert abs(kv["aggregate_ppr_throughput"] - kv["aggregate_edge_num"] / kv["aggregate_ppr_time_slowest_gpu"] * 1000.0) <= 1e-4 * kv["aggregate_ppr_throughput"]   # (printed with six significant digits)
    g = orc.Graph(V, e1, e2, 0, W, c)
    states = {s: orc.State(V, s, 1e-9) for s in sources}
    for s in states.values():
        s.cilk_execute(g)
    for _ in range(6):
        assert not g.stream_updates()
        g.inc_construct(1)
        for s in states.values():
            s.cilk_inc_execute(g)
    seen = []
    for d in range(ngpu):                                  # one dump per device thread: its round-robin share of the sources
        got = read_dump(f"{dump}.{d}")
        assert sorted(got) == sorted(sources[d::ngpu])
        seen += list(got)
        for s, (p, res) in got.items():
            assert np.max(np.abs(res)) < 1e-9
            assert np.max(np.abs(p - states[s].p)) < 1e-9
    assert sorted(seen) == sorted(sources)
    # without the flag, device 1 does not exist on a one-GPU box: a loud failure, not a silent fallback
    from dynamicppr_amd import engine as eng
    if eng.lib().dppr_device_count() < 2:
        r = run([pagerank, "-d", path, "-a", "0", "-i", "0", "-y", "1", "-n", "0", "-r", "0.01", "-b", "2", "-g", "2", "--sources", str(srcfile)])
        assert r.returncode != 0 and "no usable HIP device" in r.stdout


@pytest.mark.gpu
def test_sweep_tool_scrapes_the_stdout_contract(pagerank, small_bin, tmp_path):
    """tools/sweep.py = scripts/gpu.sh + scripts/extract_gpu.py: runs the variant sweep and scrapes
    ppr_latency / ppr_throughput from the logs."""
    import json
    path, V, e1, e2 = small_bin
    src = int(datagen.top_sources(V, e1, e2, 600, 1, 1)[0])
    r = run(["python3", os.path.join(ROOT, "tools", "sweep.py"), "variant", "--data", path, "--directed", "1",
             "--source", str(src), "--log-dir", str(tmp_path / "log")])
    assert r.returncode == 0, r.stdout
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert [(x["variant"], x["push_only"]) for x in rows] == [(v, po) for v in range(4) for po in (False, True)]   # the ablation table
    assert all(x["ppr_latency_ms"] and x["ppr_throughput"] and x["ppr_throughput"] > 0 for x in rows)
    assert len(os.listdir(tmp_path / "log")) == 8


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["source_feature", "batch_ratio"])
def test_sweep_tool_source_features_and_batch_ratios(pagerank, small_bin, tmp_path, what):
    """The other two sweeps of the reference's experiment matrix (scripts/gpu.sh:112-170): sources taken from the workload
    tool's top10 / top1000 / top1000000 files (written on demand by dynamicppr_amd/tools.py workload; a 512-vertex graph has no
    vertex of rank >= 1000, so -- like the reference's tool -- no third file), -r 0.01 / 0.001 / 0.0001."""
    import json
    path, V, e1, e2 = small_bin
    r = run(["python3", os.path.join(ROOT, "tools", "sweep.py"), what, "--data", path, "--directed", "1", "--batches", "3",
             "--log-dir", str(tmp_path / "log")])
    assert r.returncode == 0, r.stdout
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    feats = ["top10", "top1000"]
    base = os.path.basename(path)
    ids = {f: [int(x) for x in open(tmp_path / "log" / f"{base}_{f}.txt").read().split()] for f in feats}
    assert all(len(v) == 10 for v in ids.values())
    if what == "source_feature":
        assert [(x["source_feature"], x["source"]) for x in rows] == [(f, ids[f][3]) for f in feats]     # entry 3 of each file, scripts/gpu.sh:17
        assert all(x["ppr_latency_ms"] and x["ppr_throughput"] > 0 for x in rows)
    else:
        assert [(x["batch_ratio"], x["source_feature"]) for x in rows] == [(r_, f) for r_ in (0.01, 0.001, 0.0001) for f in feats]
        # (6 000 stream edges: -r 0.0001 is less than one edge per batch -- the run has no complete batch and prints no latency, as the reference would)
        assert all(x["ppr_latency_ms"] for x in rows if x["batch_ratio"] == 0.01)


@pytest.mark.gpu
@pytest.mark.parametrize("nsrc", [1, 3])
def test_cli_long_churning_stream_renumbers_and_matches_oracle(pagerank, tmp_path, nsrc):
    """The reference driver's flow (slide, update, slide, ...) over a stream that churns through the id range: the
    engine renumbers its internal ids several times on the way (thresholds lowered through the tuning variables);
    results and --validate are those of the oracle."""
    from tests.test_renumbering_gpu import churn_stream
    V, n, batches = 4096, 6000, 45
    e1, e2 = churn_stream(V, n, 400, 9)
    path = str(tmp_path / "churn.bin")
    datagen.write_bin(path, V, e1, e2)
    W, c = 600, 100
    srcs = [0, 1, 2][:nsrc]
    sf = tmp_path / "sources.txt"
    sf.write_text("\n".join(map(str, srcs)) + "\n")
    dump = str(tmp_path / "out.dump")
    outs = {}
    for renumber in ("1", "0"):
        env = dict(os.environ, DPPR_RENUMBER=renumber, DPPR_RENUMBER_PCT="10", DPPR_RENUMBER_MIN="8")
        r = subprocess.run([pagerank, "-d", path, "-a", "0", "-i", "1", "-y", "1", "-w", "0.1", "-n", "1", "-c", str(c),
                            "-l", str(c * batches), "--sources", str(sf), "--validate", "--dump", dump],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        assert r.returncode == 0, r.stdout
        assert "validate failed" not in r.stdout
        m = re.search(r"id_space ids=(\d+) parked=(\d+) renumberings=(\d+) revivals=(\d+)", r.stdout)
        assert m and ((int(m.group(3)) >= 3 and int(m.group(2)) > 0) if renumber == "1" else m.group(3) == "0"), r.stdout[-400:]
        outs[renumber] = read_dump(dump)
    for sv in srcs:
        g = orc.Graph(V, e1, e2, 1, W, c)
        s = orc.State(V, sv, 1e-9)
        s.cilk_execute(g)
        for _ in range(batches):
            assert not g.stream_updates()
            g.inc_construct(1)
            s.cilk_inc_execute(g)
        for k in outs:
            assert np.max(np.abs(outs[k][sv][0] - s.p)) < 1e-9 and np.max(np.abs(outs[k][sv][1])) < 1e-9, (k, sv)


@pytest.mark.gpu
@pytest.mark.parametrize("directed", [1, 0])
def test_cli_profile_output_matches_the_reference_profile_build(pagerank, small_bin, directed):
    """--profile = the reference compiled with -DPROFILE: one `phase_id=..,iteration_id=..,frontier_count=..` line per
    frontier iteration (gpu/PPRRevPushGPU.cuh:109-111) and the phase report of GPUProfiler::ReportProfile. With --sync
    the frontier sizes are those of the oracle's synchronous schedule, iteration for iteration."""
    path, V, e1, e2 = small_bin
    W, c, batches = 600, 6, 4
    src = int(datagen.top_sources(V, e1, e2, W, directed, 1)[0])
    r = run([pagerank, "-d", path, "-a", "0", "-i", str(directed), "-y", "1", "-w", "0.1", "-n", "0", "-r", "0.01",
             "-b", str(batches), "-s", str(src), "--sync", "--profile"])
    assert r.returncode == 0, r.stdout
    got = [(int(a), int(b), int(f)) for a, b, f in re.findall(r"^phase_id=(\d+),iteration_id=(\d+),frontier_count=(\d+)$", r.stdout, flags=re.M)]
    g = orc.Graph(V, e1, e2, directed, W, c)
    s = orc.State(V, src, 1e-9)
    s.trace(True)
    want = []

    def loop(phase):
        before = len(s.traced_frontiers())
        s.sync_main_loop(g, phase)
        for it, f in enumerate(s.traced_frontiers()[before:]):
            want.append((phase, it, len(f)))

    s.sync_execute(g)  # the from-scratch solve: Init + ExecuteMainLoop(0)
    want += [(0, it, len(f)) for it, f in enumerate(s.traced_frontiers())]
    for _ in range(batches):
        assert not g.stream_updates()
        g.inc_construct(1)
        s.copy_revert_out_degree(g)
        s.stream_update(g)
        loop(0)
        loop(1)
    assert got == want
    m = re.search(r"\*+ profile time \*+\n(.*)\n(.*)\n\*+ end profile", r.stdout)
    assert m, r.stdout[-600:]
    t = {k: float(v) for k, v in re.findall(r"\[(\w+)\]=([-+.e\d]+)ms", m.group(1))}
    assert list(t) == ["inspect_time", "expand_time", "init_graph_calculation_time", "dynamic_graph_calculation_time",
                       "exclude_graph_update_time", "sort_time", "reduce_time", "repair_frontier_time", "inc_update_time",
                       "push_time", "total_time", "ppr_time", "ppr_update_time", "ppr_query_time"]
    assert t["total_time"] >= t["init_graph_calculation_time"] + t["dynamic_graph_calculation_time"] - 1e-3
    assert t["ppr_time"] >= t["inc_update_time"] + t["push_time"] - 1e-3 and t["push_time"] > 0 and t["expand_time"] > 0
    cnt = {k: int(v) for k, v in re.findall(r"\[(\w+)\]=(\d+) ", m.group(2))}
    assert cnt["expand_count"] == sum(f for _, _, f in want) and cnt["traverse_count"] == s.stats()["E"]


@pytest.mark.gpu
def test_cli_watchdog_writes_a_post_mortem(pagerank, small_bin):
    """The diagnostic path itself: a driver that stops making progress (test hook DPPR_TEST_STALL) is ended by the
    watchdog after the limit, with the engine's state (dppr_debug_dump: last error, epoch, GridBar words read through a
    side stream, per-slot counters) on stderr and exit code 125 (124 is `timeout`'s)."""
    path, V, e1, e2 = small_bin
    src = int(datagen.top_sources(V, e1, e2, 600, 0, 1)[0])
    env = dict(os.environ, DPPR_WATCHDOG_S="2", DPPR_TEST_STALL="1")
    r = subprocess.run([pagerank, "-d", path, "-a", "0", "-i", "0", "-y", "1", "-w", "0.1", "-n", "0", "-r", "0.01", "-b", "5", "-s", str(src)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=env)
    assert r.returncode == 125, r.stdout
    assert "[watchdog] no progress" in r.stdout and "GridBar: gen" in r.stdout and "slot 0: source" in r.stdout, r.stdout
    assert "engine stream: idle" in r.stdout
