"""bench.py end to end on the smallest stand-in: ONE JSON line with the keys the driver and the judge read
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
data / config.workload, plus roofline, cpu_baseline and the in-run parity block), consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra,sources", [([], 1), (["--sources", "3"], 3), (["--sources", "10"], 10)])
def test_bench_line_contract(extra, sources):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "dblp", "--steps", "4", "--warmup", "2"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and len(d["config"]["sources"]) == sources
    # value = edge updates of all sources per second of the timed region
    c = d["config"]["batch_c"]
    assert abs(d["value"] - sources * c / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    # frac is a roofline fraction of ALGORITHMIC bytes: SURVEY 8(d)'s 72 F + 24 E + 4 N for a push / binned form; a PULL form (the resident
    # launch of one source, the group sweep) performs no residual RMW per edge and is priced 72 F + (8 + 4 / S) E + 4 N
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0 < rf["frac"] <= 1
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) <= 2e-3 * rf["achieved"]
    wr = rf["work_rate_survey_unit"]      # SURVEY 8(d)'s unit beside it: never smaller
    assert rf["form"] in ("resident", "pull") and wr["over_peak"] > rf["frac"]
    assert f"(8 + 4 / {sources})" in rf["bytes_model"]
    assert "fabric" in rf["traffic_kind"] and rf["all_iteration_launches"]["launches"] >= rf["launches"] > 0
    ce = rf["ceilings_measured_in_this_run"]   # SURVEY 8(d)'s calibrated ceilings, measured by this very run
    assert 1e10 < ce["line_fills_per_s"] < 2e11 and 5e9 < ce["returning_f64_atomics_per_s"] < 1e11 and 2000 < ce["stream_copy_GBps"] < 8000
    # counter traffic is taken only from a profile of THIS build of the library; anything else is refused with a reason
    from dynamicppr_amd import engine as eng
    assert rf["build_id"] == d["build_id"] == eng.build_id()
    assert rf["traffic"] is None or eng.build_id() in rf["traffic_source"]
    assert rf["traffic"] is not None or rf["frac_traffic"] is None
    tr = d["config"]["timed_region"]    # `value` is measured under the reference's bracket; the other accounting is carried beside it
    assert tr["grouping"] == "in_region" and tr["copy_out_degree"] == "in_region" and tr["grouping_ms_per_step"] > 0
    asl = tr["at_slide_accounting"]
    assert asl["how"].startswith("measured") and 0 < asl["event_ms_per_step"] < 2.0 * d["event_ms_per_step"]   # (four sub-millisecond steps: a sanity bound)
    assert asl["max_abs_dp_vs_headline_state"] < 1e-12   # (same batches, same schedule: the two accountings end in the same state)
    # the rolling epoch ring (the reference's loop shape): three resident epochs, the graph update between the timed brackets, HBM accounted for
    assert "rolling ring of 3 resident epochs" in tr["loop"] and d["wall_ms_per_step_incl_graph_update"] > d["ms_per_step"]
    hb = d["config"]["hbm"]
    assert hb["resident_epochs"] == 3 and 0 < hb["hbm_peak_bytes"] <= hb["hbm_total_bytes"] and hb["plan_bytes"] > 0
    assert d["ranks_seen"] == 1 and d["backend"] is None and len(d["per_rank_ms_per_step"]) == 1 and d["launcher"] == "single process"
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    pr = d["parity"]
    assert pr["ok"] is True and pr["max_abs_residual"] < pr["eps"] and pr["invariant_max_err"] < 1e-12
    assert pr["max_abs_dp_vs_cpu_t1"] is not None and pr["max_abs_dp_vs_cpu_t1"] < pr["tolerance"]
    ml = d["merged_loop"]     # the same steps with the merged loop, reported BESIDE the headline (never as it)
    assert ml["parity"]["ok"] is True and ml["parity"]["max_abs_residual"] <= pr["eps"] / 4 and ml["parity"]["max_abs_dp_vs_cpu_t1"] < pr["tolerance"]
    assert ml["ms_per_step"] > 0 and abs(ml["speedup_vs_value"] - d["ms_per_step"] / ml["ms_per_step"]) < 0.02 * ml["speedup_vs_value"]
    assert "end of the timed region" in pr["cpu_compared"]
    # all of the rank's sources against the CPU when that leg is cheap (it is, on this stand-in); otherwise two, and the line says why
    assert (f"all {sources} sources" in pr["cpu_compared"] or "cpu_compared_why_not_all" in pr) if sources > 2 else f"{sources} source(s)" in pr["cpu_compared"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` typed as such (no torchrun in front): the script starts torch.distributed.run itself as a CHILD
    process before anything has touched HIP, forwards the one line of rank 0 and its exit code. On a one-GPU box the two ranks
    share the device (gloo carries the barrier: RCCL refuses two ranks on one device); the line proves both ranks ran."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "dblp", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] in ("nccl", "gloo") and d["launcher"] == "self"
    # (a rank's own time ends when its steps are done and its device has drained; the bracket ends at the closing barrier)
    assert len(d["per_rank_ms_per_step"]) == 2 and 0 <= d["ms_per_step"] - max(d["per_rank_ms_per_step"]) < 0.05 * d["ms_per_step"] + 0.05
    assert d["scaling"] == "weak" and d["parity"]["ok"] is True
    c = d["config"]["batch_c"]
    assert abs(d["value"] - 2 * c / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]      # both ranks' units over the slowest rank's time


def run_bench(args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_rolling_ring_against_prestaged_epochs():
    """the default loop keeps THREE epochs resident and alternates the untimed graph update with the timed step
    (gpu/PPRGPU.cuh:109-169); --prestage builds every epoch first (rounds 1-5). Same batches, same solver: the states agree, the
    iteration counts are equal, and the ring holds far fewer bytes."""
    common = ["--config", "youtube", "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--no-merged", "--no-extra-passes", "--no-ceilings", "--schedule", "sync"]
    r1, ring = run_bench(common)
    r2, pre = run_bench(common + ["--prestage"])
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-1500:], r2.stderr[-1500:])
    assert "rolling ring" in ring["config"]["timed_region"]["loop"] and "pre-staged" in pre["config"]["timed_region"]["loop"]
    assert ring["config"]["hbm"]["resident_epochs"] == 3 and pre["config"]["hbm"]["resident_epochs"] == 12 + 3 + 5 + 1
    assert ring["config"]["hbm"]["plan_bytes"] < pre["config"]["hbm"]["plan_bytes"]
    assert ring["parity"]["ok"] and pre["parity"]["ok"]
    # (synchronous schedule: the same frontiers up to a threshold flip where the ring's run renumbered its ids and sums in another order)
    assert abs(ring["iterations_per_step"] - pre["iterations_per_step"]) <= 0.02 * pre["iterations_per_step"]
    assert abs(ring["edges_pushed_per_step"] - pre["edges_pushed_per_step"]) <= 0.02 * pre["edges_pushed_per_step"]
    assert 0.5 < ring["ms_per_step"] / pre["ms_per_step"] < 2.0     # (sub-millisecond steps on a shared box: a sanity bound; the A/B is in DESIGN.md)


def test_a_plan_that_cannot_fit_is_refused_up_front():
    r, d = run_bench(["--config", "dblp", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--hbm-limit-gb", "0.01"])
    assert r.returncode != 0 and d is None and "HBM plan does not fit" in r.stderr and "resident epochs" in r.stderr
